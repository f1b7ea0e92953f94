#!/bin/bash
# r06 step 3: what the third launch of a pass costs (DINT_EXP_NO_LATE: none at all), and the kernel trace of the new chain
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/a3
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
run tatp_late DINT_X=0
run tatp_nolate DINT_EXP_NO_LATE=1
run tatp_late2 DINT_X=0
run tatp_nolate2 DINT_EXP_NO_LATE=1
ARGS="--workload store"
run store_late DINT_X=0
run store_nolate DINT_EXP_NO_LATE=1
echo "== rocprofv3 kernel trace, tatp"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o tatp -- python "$ROOT/bench.py" --legs headline --steps 12 --warmup 1 > "$OUT/prof.log" 2>&1
cd "$ROOT"
python tools/trace_summary.py "$OUT/prof" 2>&1 | head -30
find "$OUT/prof" -name "*.db" -size +30M -delete
