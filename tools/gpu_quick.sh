#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "== tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d["roofline"]["kernel"], d["roofline"]["frac"])'
for i in 1 2 3; do
echo "== r01"; (cd gpurun_tmp/r01 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P")
echo "== current"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
done
echo "== current 100"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== store"; timeout 300 python bench.py --workload store --steps 50 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
echo "== fasst"; timeout 300 python bench.py --workload fasst --steps 50 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
