#!/bin/bash
# r06 step 1: look-ahead passes (k_kv_hot_part) + in-place replay.  kv tests, then the headline A/B in one box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/a1
mkdir -p "$OUT"
cd "$ROOT"
echo "== ahead tests"; timeout 900 python -m pytest tests/test_gpu_ahead.py -x -q 2>&1 | tail -15
echo "== kv tests"; timeout 1200 python -m pytest tests/test_gpu_kv.py tests/test_gpu_async.py tests/test_gpu_driver.py -x -q 2>&1 | tail -5
for cfg in "inplace ahead" "copy noahead" "inplace noahead" "copy ahead"; do
  set -- $cfg
  fl="--replay $1"; [ "$2" = noahead ] && fl="$fl --no-ahead"
  echo "== tatp headline: $cfg"
  timeout 600 python bench.py --legs headline $fl > "$OUT/tatp_$1_$2.json" 2> "$OUT/tatp_$1_$2.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/tatp_$1_$2.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_epoch"], d.get("value_repeats"), d.get("kernels_us"), d["latency_us"], d.get("parity_failures"))
except Exception as e:
    print("failed", e); print(open("$OUT/tatp_$1_$2.err").read()[-2000:])
PY
done
for cfg in "inplace ahead" "copy noahead"; do
  set -- $cfg
  fl="--replay $1"; [ "$2" = noahead ] && fl="$fl --no-ahead"
  echo "== store: $cfg"
  timeout 600 python bench.py --workload store --legs headline $fl > "$OUT/store_$1_$2.json" 2> "$OUT/store_$1_$2.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/store_$1_$2.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"])
except Exception as e:
    print("failed", e); print(open("$OUT/store_$1_$2.err").read()[-2000:])
PY
done
