#!/bin/bash
# r06: the exchange on one GPU -- launch set against one stream per home engine, with the one-launch pass
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/x1
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 900 python bench.py --force-exchange --legs headline --steps 10 --warmup 3 > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("ms_per_epoch"), d.get("value_repeats"), d["latency_us"]["p50"], d["latency_us"]["p99"], d.get("route_overflow"), d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.err").read()[-1500:])
PY
}
run set DINT_X=0
run streams DINT_ROUTER_STREAMS=1
run set_nofuse DINT_KV_NO_FUSE=1
run streams_w48 DINT_ROUTER_STREAMS=1 DINT_KV_WORKERS=48
run set_q8 GPU_MAX_HW_QUEUES=8
run streams_q8 DINT_ROUTER_STREAMS=1 GPU_MAX_HW_QUEUES=8
