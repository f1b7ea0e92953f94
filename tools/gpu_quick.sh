#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["requests_per_step"], d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("kernels_us"))'
for i in 1 2; do
echo "== r01"; (cd gpurun_tmp/r01 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P")
echo "== split, r01 key stream"; DINT_ZIPF_GRAY=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== merged no hot, r01 key stream"; DINT_ZIPF_GRAY=1 DINT_BENCH_FLAGS=8 DINT_KV_MERGED=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== split, exact zipf"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
done
