#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/s3
mkdir -p "$OUT"
cd "$ROOT"
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_kv.py tests/test_long_traces.py -x -q 2>&1 | tail -5
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
run sb_ahead DINT_X=0
run sb_noahead DINT_KV_NO_AHEAD=1
