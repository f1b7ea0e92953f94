#!/bin/bash
# r06 step 2: k_kv_late (the light kernel for what k_kv_hot leaves) against r05's k_kv_big launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/a2
mkdir -p "$OUT"
cd "$ROOT"
echo "== kv + ahead tests"; timeout 1500 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py tests/test_gpu_async.py tests/test_gpu_driver.py tests/test_long_traces.py -x -q 2>&1 | tail -8
for late in 0 1; do
  echo "== tatp headline: DINT_KV_LATE_BIG=$late"
  DINT_KV_LATE_BIG=$late timeout 600 python bench.py --legs headline > "$OUT/tatp_late$late.json" 2> "$OUT/tatp_late$late.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/tatp_late$late.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_epoch"], d.get("value_repeats"), d.get("kernels_us"), d["latency_us"], d.get("parity_failures"))
except Exception as e:
    print("failed", e); print(open("$OUT/tatp_late$late.err").read()[-2000:])
PY
done
echo "== store"
timeout 600 python bench.py --workload store --legs headline > "$OUT/store.json" 2> "$OUT/store.err"
python - <<PY
import json
d = json.loads(open("$OUT/store.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"])
PY
echo "== smallbank"
timeout 600 python bench.py --workload smallbank --legs headline > "$OUT/smallbank.json" 2> "$OUT/smallbank.err"
python - <<PY
import json
d = json.loads(open("$OUT/smallbank.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_epoch"], d.get("kernels_us"), d["latency_us"], d.get("parity_failures"))
PY
