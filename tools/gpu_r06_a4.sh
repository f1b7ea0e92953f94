#!/bin/bash
# r06 step 4: after the split of k_kv.hip -- kv tests again; what goes late in the bench stream; what one more (dry) launch costs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/a4
mkdir -p "$OUT"
cd "$ROOT"
echo "== kv + ahead tests"; timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py tests/test_abi.py -x -q 2>&1 | tail -5
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
run tatp_1 DINT_X=0
run tatp_2launch DINT_EXP_NO_LATE=4
run tatp_1b DINT_X=0
run tatp_latebig DINT_KV_LATE_BIG=1
ARGS="--theta 0"
run nurand_1 DINT_X=0
run nurand_nolate DINT_EXP_NO_LATE=1
