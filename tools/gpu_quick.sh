#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== lock tests"; timeout 1500 python -m pytest tests/test_gpu_locks.py tests/test_gpu_route.py tests/test_fasst_24m.py tests/test_gpu_shim.py tests/test_gpu_async.py -x -q -m gpu 2>&1 | tail -12
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"], d["roofline"]["frac"], d.get("replay_equals_recorded"), (d.get("cpu_baseline") or {}).get("oracle_parity"))'
echo "== r01 fasst 1M"; (cd gpurun_tmp/r01 && timeout 300 python bench.py --workload fasst --steps 50 --no-cpu-baseline --no-rand64 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"])')
echo "== fasst 1M slots"; timeout 300 python bench.py --workload fasst --steps 50 --no-rand64 2>/dev/null | python -c "$P"
echo "== fasst 36M slots"; timeout 300 python bench.py --workload fasst --slots 36000000 --steps 50 --no-rand64 2>/dev/null | python -c "$P"
