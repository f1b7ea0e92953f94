#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for d in 0 1 2 3; do DINT_TXN_DBG=$d python tools/exp_emit.py 2>&1 | tail -1; done
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("value_repeats"))'
echo "== fx q=4"; GPU_MAX_HW_QUEUES=4 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== fx q=8"; GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
echo "== fx q=16"; GPU_MAX_HW_QUEUES=16 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
