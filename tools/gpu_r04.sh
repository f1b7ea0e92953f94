#!/bin/bash
# r04 checkpoint on one box: the whole GPU suite, smoke, the default bench line (as the driver runs it)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
if [ "${1:-all}" != "bench" ]; then
echo "== gpu suite"; timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 --durations=8 2>&1 | tail -16 | tee $OUT/pytest.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
fi
echo "== default bench"; T0=$(date +%s); timeout 900 python bench.py > $OUT/bench_tatp.json 2> $OUT/bench_tatp.err; echo "rc $? in $(( $(date +%s) - T0 )) s"; tail -2 $OUT/bench_tatp.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_tatp.json").read().strip().splitlines()[-1])
print("value", d.get("value"), d.get("value_unchecked"), "ms/step", d.get("ms_per_step"), "kernels", d.get("kernels_us"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "traffic_over_alg")}, "rand64", (d.get("roofline_rand64") or {}).get("frac"))
print("lat", d.get("latency_us"), "closed", (d.get("closed_loop") or {}).get("value"), "pcie", d.get("value_pcie"))
cb = d.get("cpu_baseline") or {}
print("cpu", cb.get("kind"), cb.get("value"), (cb.get("reference_parity") or {}).get("ok"), "gpu_same", (cb.get("gpu_same_config") or {}).get("value"))
print("as_shipped_tatp", {k: v for k, v in (d.get("cpu_as_shipped_tatp") or {}).items() if k in ("value", "cores", "ops_per_s", "lost", "populate_s", "error")})
print("parity_failures", d.get("parity_failures"))
for w, r in (d.get("other_workloads") or {}).items():
    print(w, r.get("value"), r.get("kernels_us"), (r.get("roofline") or {}).get("frac"), r.get("oracle_parity"), r.get("pass_1m"), r.get("mixes"), r.get("error"))
PY
