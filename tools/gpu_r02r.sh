#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== route tests"; timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_driver.py tests/test_gpu_kv.py -x -q -m gpu 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["latency_us"], d["route_overflow"], d.get("value_repeats"))'
for i in 1 2; do
echo "== force-exchange"; timeout 300 python bench.py --force-exchange --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
done
echo "== fasst force-exchange"; timeout 300 python bench.py --workload fasst --force-exchange --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
echo "== --gpus 2 one gpu"; timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | python -c "$P"
