#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== route tests"; timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["latency_us"], d["route_overflow"], d.get("value_repeats"))'
echo "== force-exchange"; timeout 300 python bench.py --force-exchange --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fx; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fx -o fx -- python $ROOT/bench.py --force-exchange --steps 40 --warmup 5 --no-cpu-baseline --no-rand64 2>/dev/null | tail -1 | cut -c1-120
python $ROOT/tools/trace_summary.py /tmp/fx | grep -i "route\|copyBuffer"
