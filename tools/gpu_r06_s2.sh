#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s2
mkdir -p "$OUT"
cd "$ROOT"
echo "== segment test"; timeout 600 python -m pytest tests/test_gpu_route.py -x -q -k half_filled 2>&1 | grep -E "^E|passed|failed" | head -12
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 900 python bench.py --workload smallbank --legs headline $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("ms_per_epoch", d["ms_per_step"]), d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
ARGS=""
run sb_1024 DINT_KV_SB_SPLIT_MIN=1024
run sb_1536 DINT_KV_SB_SPLIT_MIN=1536
run sb_2048_t448 DINT_KV_SB_SPLIT_MIN=2048 DINT_KV_SPLIT_TARGET=448
run sb_768_t320 DINT_KV_SB_SPLIT_MIN=768 DINT_KV_SPLIT_TARGET=320
