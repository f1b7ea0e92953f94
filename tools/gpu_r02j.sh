#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_kv.py tests/test_gpu_locks.py tests/test_fasst_24m.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/exp_lock_big.py 1048576 65536 4 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("value_repeats"))'
echo "== fasst"; timeout 300 python bench.py --workload fasst --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
echo "== tatp"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop 2>/dev/null | python -c "$P"
echo "== tatp"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop 2>/dev/null | python -c "$P"
