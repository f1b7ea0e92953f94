#!/bin/bash
# r05, after DINT_FLAG_INPUTS_READY (k_locks.hip's launcher changed): the whole GPU suite + smoke, the lock profiles, the default
# line (its other_workloads legs run the lock benches with the inputs_ready leg), the two lock lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05 gpurun_out/profiles
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r05/final3_suite.txt; el
echo "== smoke"; timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1 | tee -a gpurun_out/r05/final3_suite.txt
for w in fasst 2pl; do
  timeout 600 python tools/profile_bench.py r05 --workload $w > gpurun_out/profiles/r05_$w.log 2>&1; echo "profile $w rc $? $(el)"
done
cp gpurun_out/profiles/traffic_*.json profiles/ 2>/dev/null
echo "== default line"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/final3_bench_tatp.json 2> gpurun_out/r05/final3_bench_tatp.err; echo "rc $? $(el)"
for w in fasst 2pl; do
  timeout 500 python bench.py --workload $w > gpurun_out/r05/final3_bench_$w.json 2>/dev/null; echo "bench $w rc $? $(el)"
done
python - <<'P'
import json
d=json.loads(open("gpurun_out/r05/final3_bench_tatp.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","value_repeats","parity_failures")})
for w in ("fasst","2pl"):
    x=json.loads(open(f"gpurun_out/r05/final3_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, x["value"], x.get("inputs_ready"), x["roofline"].get("from_profile"))
    o=d["other_workloads"][w]; print(" leg", o["value"], o.get("inputs_ready"))
P
