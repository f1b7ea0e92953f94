#!/bin/bash
# r06 step b2: k_kv_pass -- how many workers, and where the partition's tiles go
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/b2
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 600 python bench.py --legs headline $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
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
run pf1_w192 DINT_KV_PART_FIRST=1 DINT_KV_WORKERS=192
run pf1_w96 DINT_KV_PART_FIRST=1 DINT_KV_WORKERS=96
run pf1_w48 DINT_KV_PART_FIRST=1 DINT_KV_WORKERS=48
run pf0_w96 DINT_KV_PART_FIRST=0 DINT_KV_WORKERS=96
run pf0_w192 DINT_KV_PART_FIRST=0 DINT_KV_WORKERS=192
run nofuse DINT_KV_NO_FUSE=1
run pf1_w320 DINT_KV_PART_FIRST=1 DINT_KV_WORKERS=320
