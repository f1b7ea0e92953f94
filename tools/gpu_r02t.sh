#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fx; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/fx -o fx -- python $ROOT/bench.py --force-exchange --steps 40 --warmup 5 --no-cpu-baseline --no-rand64 2>/dev/null | tail -1 | cut -c1-200
python $ROOT/tools/trace_summary.py /tmp/fx
