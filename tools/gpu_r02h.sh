#!/bin/bash
# dominant-key / dominant-slot thresholds: fasst and tatp bench at several values (DINT_LOCK_HOT_MIN / DINT_KV_HOT_MIN)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("value_repeats"))'
echo "== lock tests"; timeout 900 python -m pytest tests/test_gpu_locks.py -x -q 2>&1 | tail -2
for h in 256 512 1024 100000; do
echo "== fasst hot_min $h"; DINT_LOCK_HOT_MIN=$h timeout 300 python bench.py --workload fasst --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
done
for h in 256 512 1024 2048; do
echo "== tatp hot_min $h"; DINT_KV_HOT_MIN=$h timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop 2>/dev/null | python -c "$P"
done
echo "== fasst 36M slots, 1M passes"; timeout 300 python bench.py --workload fasst --slots 36000000 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
