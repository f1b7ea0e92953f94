#!/bin/bash
# r06: the closed loop and the host path with / without the one-launch pass (launch sets of three engines, no look-ahead)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c2
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 900 python bench.py --legs gpu --no-other-workloads --no-rand64 $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    cl = d.get("closed_loop") or {}
    print("$name", d["value"], "closed", cl.get("value"), cl.get("ms_per_epoch"), "two_groups", (cl.get("two_groups") or {}).get("value"), "pcie", d.get("value_pcie"), (d.get("latency_host_us") or {}).get("p50"), "exchange", (d.get("exchange") or {}).get("value"), d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
ARGS=""
run fused DINT_X=0
run nofuse DINT_KV_NO_FUSE=1
run fused_w34 DINT_KV_WORKERS=34
run fused_w192 DINT_KV_WORKERS=192
