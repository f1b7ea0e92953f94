#!/bin/bash
# r06 step 5: k_kv_late with the closed forms first (two hot keys in one sub), thin (128 VGPRs, spills) and fat (256 VGPRs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/a5
mkdir -p "$OUT"
cd "$ROOT"
echo "== kv + ahead tests"; timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py tests/test_abi.py -x -q 2>&1 | tail -5
echo "== kv tests, fat late kernel"; DINT_KV_LATE_FAT=1 timeout 1500 python -m pytest tests/test_gpu_kv.py -x -q -k "hot_key or pieces or partition" 2>&1 | tail -3
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 600 python bench.py --legs headline $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("ms_per_epoch", d["ms_per_step"]), d.get("value_repeats"), d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("late"), d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
ARGS=""
run tatp_thin DINT_X=0
run tatp_fat DINT_KV_LATE_FAT=1
run tatp_thin2 DINT_X=0
run tatp_fat2 DINT_KV_LATE_FAT=1
run tatp_latebig DINT_KV_LATE_BIG=1
