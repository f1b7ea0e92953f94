#!/bin/bash
# A/B of a kernel change: full GPU suite, then the three bench workloads without the CPU legs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("value_repeats"), (d.get("closed_loop") or {}).get("value"))'
for i in 1 2; do
echo "== tatp"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
done
echo "== tatp 20"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== store"; timeout 300 python bench.py --workload store --steps 50 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
echo "== fasst"; timeout 300 python bench.py --workload fasst --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
echo "== smallbank"; timeout 300 python bench.py --workload smallbank --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
