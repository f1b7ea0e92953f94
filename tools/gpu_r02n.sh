#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== kv tests"; timeout 1200 python -m pytest tests/test_gpu_kv.py -x -q 2>&1 | tail -2
export AB_N=2
AB_ARGS="--workload smallbank --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop" bash tools/gpu_ab2.sh
AB_ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop" bash tools/gpu_ab2.sh
