#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== driver tests"; timeout 900 python -m pytest tests/test_gpu_gdriver.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -2
for d in 0 2; do DINT_TXN_DBG=$d timeout 120 python tools/exp_emit.py 2>&1 | tail -1; done
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["closed_loop"])'
echo "== tatp closed loop"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== smallbank closed loop"; timeout 300 python bench.py --workload smallbank --steps 30 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
