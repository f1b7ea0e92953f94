#!/bin/bash
# r06: smallbank's hot account in pieces (kv_sb_item)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s1
mkdir -p "$OUT"
cd "$ROOT"
echo "== segment test"; timeout 600 python -m pytest tests/test_gpu_route.py -x -q -k half_filled 2>&1 | grep -E "assert|passed|failed" | head -5
echo "== smallbank tests"; timeout 1500 python -m pytest tests -m gpu -x -q -k "smallbank or sb_ or Smallbank" 2>&1 | tail -6
echo "== smallbank tests, small pieces"; DINT_KV_SB_SPLIT_MIN=200 DINT_KV_SPLIT_TARGET=64 timeout 1500 python -m pytest tests -m gpu -x -q -k "smallbank or sb_" 2>&1 | tail -6
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 900 python bench.py --workload smallbank --legs headline $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("ms_per_epoch", d["ms_per_step"]), d.get("value_repeats"), d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
ARGS=""
run sb_pieces DINT_X=0
run sb_old DINT_KV_SB_SPLIT_MIN=0
run sb_pieces_2048 DINT_KV_SB_SPLIT_MIN=2048
run sb_pieces_t256 DINT_KV_SPLIT_TARGET=256
