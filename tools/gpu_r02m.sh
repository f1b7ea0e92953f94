#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== kv tests"; timeout 1200 python -m pytest tests/test_gpu_kv.py tests/test_gpu_driver.py tests/test_gpu_gdriver.py -x -q 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("value_repeats"), (d.get("closed_loop") or {}).get("value"), (d.get("cpu_baseline") or {}).get("oracle_parity"), (d.get("cpu_baseline") or {}).get("value"))'
echo "== smallbank (with the oracle on the bench stream)"; timeout 600 python bench.py --workload smallbank --steps 20 --warmup 5 --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
EXP_WL=smallbank EXP_EPOCHS=3 timeout 600 python tools/exp_big.py 524288 0.99 2>&1 | tail -8
