#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== route tests"; timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], d["latency_us"]["p50"], d.get("value_repeats"))'
for v in 0 1 0 1; do
echo "== force-exchange DINT_BWD_STREAM=$v"; DINT_BWD_STREAM=$v timeout 300 python bench.py --force-exchange --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P"
done
