#!/bin/bash
# r04: the per-workload bench lines (timed region + kernels + roofline only; the CPU legs are in the default line)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
T0=$(date +%s)
timeout 120 python bench.py --force-exchange $ARGS > $OUT/bench_tatp_force_exchange.json 2>/dev/null; echo "fx rc $? $(( $(date +%s) - T0 )) s"
for w in smallbank store fasst 2pl log; do
  timeout 120 python bench.py --workload $w $ARGS > $OUT/bench_$w.json 2>/dev/null; echo "$w rc $? $(( $(date +%s) - T0 )) s"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), d.get("kernels_us"), d.get("latency_us", {}).get("p50"), d.get("latency_us", {}).get("p99"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic_over_alg"))
    except Exception as e:
        print(f, "unreadable", e)
PY
