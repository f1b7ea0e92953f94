#!/bin/bash
# r06: same-box A/B of environment knobs on one bench workload (every experiment of NOTEBOOK.md round 6 was a call of this shape):
#   tools/gpurun.sh -- 'bash tools/gpu_r06_ab.sh tatp "base:DINT_X=0" "nofuse:DINT_KV_NO_FUSE=1" "w48:DINT_KV_WORKERS=48"'
#   AB_ARGS="--force-exchange --steps 10 --warmup 3" / AB_TESTS="tests/test_gpu_kv.py" (run first) are optional
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
cd "$ROOT"
WL=$1; shift
if [ -n "${AB_TESTS:-}" ]; then echo "== $AB_TESTS"; timeout 1800 python -m pytest $AB_TESTS -x -q 2>&1 | tail -5; fi
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env ${envs//;/ } timeout 900 python bench.py --workload $WL --legs headline ${AB_ARGS:-} > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("ms_per_epoch", d["ms_per_step"]), d.get("value_repeats"), d.get("kernels_us"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("late"), d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
done
