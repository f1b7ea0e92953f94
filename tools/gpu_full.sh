#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/full
mkdir -p "$OUT"
cd "$ROOT"
echo "== full gpu suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest.log"
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench default"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_20.json" 2> "$OUT/bench_tatp_20.err"; tail -2 "$OUT/bench_tatp_20.err"; cat "$OUT/bench_tatp_20.json"
