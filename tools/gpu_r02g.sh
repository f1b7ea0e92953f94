#!/bin/bash
# smallbank walk A/B: the kv GPU tests, then the smallbank bench (and tatp as a regression check)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== kv tests"; timeout 1200 python -m pytest tests/test_gpu_kv.py tests/test_gpu_driver.py tests/test_gpu_gdriver.py -x -q 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("value_repeats"), (d.get("closed_loop") or {}).get("value"))'
echo "== smallbank"; timeout 300 python bench.py --workload smallbank --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== smallbank reference dist"; timeout 300 python bench.py --workload smallbank --theta 0 --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== tatp"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
