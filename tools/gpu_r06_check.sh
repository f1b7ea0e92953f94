#!/bin/bash
# r06 checkpoint: the whole GPU suite, smoke, the default bench line as the driver runs it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c1
mkdir -p "$OUT"
cd "$ROOT"
echo "== gpu suite"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== default line"; timeout 1500 python bench.py > "$OUT/bench_tatp.json" 2> "$OUT/bench_tatp.err"; echo "rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_tatp.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["value_repeats"], d["latency_us"], d.get("parity_failures"))
print("roofline", {k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list))})
print("closed", d["closed_loop"]["value"] if d.get("closed_loop") else None, "pcie", d.get("value_pcie"), "exchange", d.get("exchange"))
for k, v in (d.get("other_workloads") or {}).items():
    print(k, v.get("value"), v.get("kernels_us"), v.get("error"), (v.get("pass_1m") or {}).get("value") if isinstance(v.get("pass_1m"), dict) else v.get("pass_1m"))
PY
