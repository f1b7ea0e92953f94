#!/bin/bash
# r06 step b3: k_kv_pass with helping resolve workgroups: how many dedicated workers
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/b3
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
run help_w34 DINT_KV_WORKERS=34
run help_w64 DINT_KV_WORKERS=64
run help_w96 DINT_KV_WORKERS=96
run nohelp_w96 DINT_KV_HELP=0 DINT_KV_WORKERS=96
run help_w64_pf0 DINT_KV_WORKERS=64 DINT_KV_PART_FIRST=0
run help_w128 DINT_KV_WORKERS=128
ARGS="--workload store"
run store_help_w64 DINT_KV_WORKERS=64
run store_nohelp_w96 DINT_KV_HELP=0 DINT_KV_WORKERS=96
