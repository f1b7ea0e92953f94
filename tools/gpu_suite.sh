#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== --gpus 2 on one GPU"; timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['transport'], d['route_overflow'])"
echo "== fasst force-exchange"; timeout 300 python bench.py --workload fasst --force-exchange --no-cpu-baseline --no-rand64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['replay_equals_recorded'])"
