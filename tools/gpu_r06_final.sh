#!/bin/bash
# r06, round end: the whole GPU suite + smoke, tools/profile_bench.py (rocprofv3 kernel trace + the four PMC passes) for the
# workloads named in PROFILE_WL (default: all six), the bench lines (default = the driver's command, --workload ..., --force-exchange),
# the kernel resource table.  Everything lands in gpurun_out/final/; copy what is to be judged into profiles/.
#   tools/gpurun.sh --timeout 5400 -- 'bash tools/gpu_r06_final.sh r06'
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
cd "$ROOT"
if [ -z "${SKIP_SUITE:-}" ]; then
  echo "== gpu suite"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee "$OUT/${TAG}_gpu_suite.txt"
  echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a "$OUT/${TAG}_gpu_suite.txt"
fi
for wl in ${PROFILE_WL:-tatp store smallbank fasst 2pl log}; do
  echo "== profile $wl"
  timeout 1500 python tools/profile_bench.py $TAG --workload $wl > "$OUT/profile_$wl.log" 2>&1
  tail -9 "$OUT/profile_$wl.log" | cut -c1-200
  # bench.py quotes profiles/traffic_<w>.json while the kernel sources are the profiled ones: the lines below are of this build
  cp -f gpurun_out/profiles/traffic_$wl.json profiles/ 2>/dev/null
done
line() {  # name, args...
  name=$1; shift
  timeout 1800 python bench.py "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("$name", d["value"], d["unit"], "ms/step", d["ms_per_step"], "lat", d.get("latency_us", {}).get("p50"), d.get("latency_us", {}).get("p99"),
          "roofline", r.get("kernel"), r.get("frac"), "traffic/alg", r.get("traffic_over_alg"), "rand64", r.get("rand64_frac"), "parity", d.get("parity_failures"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/${TAG}_bench_$name.err").read()[-800:])
PY
}
if [ -z "${SKIP_LINES:-}" ]; then
  line tatp
  for wl in store smallbank fasst 2pl log; do line $wl --workload $wl; done
  line tatp_force_exchange --force-exchange --no-cpu-baseline --no-rand64 --no-other-workloads
fi
