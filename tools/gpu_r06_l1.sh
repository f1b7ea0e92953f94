#!/bin/bash
# r06: lock tables -- k_lock_pass (resolve of batch k + count of batch k + 1 in one launch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/l1
mkdir -p "$OUT"
cd "$ROOT"
echo "== lock tests"; timeout 1500 python -m pytest tests/test_gpu_locks.py tests/test_gpu_ahead.py tests/test_fasst_24m.py tests/test_gpu_async.py -x -q 2>&1 | tail -6
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 900 python bench.py --workload $wl --legs headline $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["ms_per_step"], d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
ARGS=""
run fasst_ahead fasst DINT_X=0
run fasst_plain fasst DINT_LOCK_NO_FUSE=1
run tpl_ahead 2pl DINT_X=0
run tpl_plain 2pl DINT_LOCK_NO_FUSE=1
ARGS="--slots 36000000"
run fasst36_ahead fasst DINT_X=0
