#!/bin/bash
# last check of a round: the full GPU suite, smoke, and the exchange-path lines for profiles/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/last
mkdir -p "$OUT"
cd "$ROOT"
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== force-exchange"; timeout 600 python bench.py --force-exchange --no-cpu-baseline --no-rand64 > "$OUT/bench_tatp_force_exchange.json" 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/bench_tatp_force_exchange.json').read()); print(d['value'], d['ms_per_step'], d['host_issue_ms_per_step'])"
echo "== --gpus 2 on one GPU"; timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/bench_tatp_gpus2_onegpu_host.json"; python -c "import json; d=json.loads(open('$OUT/bench_tatp_gpus2_onegpu_host.json').read()); print(d['value'], d['n_gpus'], d['config']['transport'], d['route_overflow'])"
