#!/bin/bash
# r06 step b4: defaults (96 workers, workers before tiles) -- tests; k_kv_late without kv_do_request's scratch (A/B of two builds)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/b4
mkdir -p "$OUT"
cd "$ROOT"
echo "== kv + ahead tests"; timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py -x -q 2>&1 | tail -4
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
run base DINT_X=0
run nos DINT_LIB_PATH=$ROOT/gpurun_tmp/libdint_nos.so
run base2 DINT_X=0
run nos2 DINT_LIB_PATH=$ROOT/gpurun_tmp/libdint_nos.so
run fat DINT_KV_LATE_FAT=1
run w80 DINT_KV_WORKERS=80
run w112 DINT_KV_WORKERS=112
