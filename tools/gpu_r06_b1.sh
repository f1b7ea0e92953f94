#!/bin/bash
# r06 step b1: ONE launch per pass (k_kv_pass: resolve + hot-key workers + the next partition) against early r06's resolve -> hot_part
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/b1
mkdir -p "$OUT"
cd "$ROOT"
echo "== kv + ahead tests"; timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py tests/test_gpu_async.py tests/test_gpu_driver.py -x -q 2>&1 | tail -8
echo "== same, DINT_KV_NO_FUSE=1"; DINT_KV_NO_FUSE=1 timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py -x -q 2>&1 | tail -3
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
run tatp_fused DINT_X=0
run tatp_nofuse DINT_KV_NO_FUSE=1
run tatp_fused2 DINT_X=0
ARGS="--workload store"
run store_fused DINT_X=0
run store_nofuse DINT_KV_NO_FUSE=1
ARGS="--no-ahead"
run tatp_fused_noahead DINT_X=0
