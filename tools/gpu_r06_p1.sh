#!/bin/bash
# r06: rocprofv3 kernel trace + the four PMC passes for tatp and store (tools/profile_bench.py), the new segment test
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== segment test"; timeout 600 python -m pytest tests/test_gpu_route.py -x -q -k half_filled 2>&1 | tail -3
for wl in tatp store; do
  echo "== profile $wl"
  timeout 1500 python tools/profile_bench.py r06a --workload $wl 2>&1 | tail -14
done
