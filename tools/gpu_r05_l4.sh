#!/bin/bash
# round 5: rocprofv3 + PMC profiles of the two lock workloads (k_locks.hip changed), their bench lines, a smallbank pass trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05 gpurun_out/profiles
export TMPDIR=/tmp
for w in fasst 2pl; do
  timeout 600 python tools/profile_bench.py r05 --workload $w > gpurun_out/profiles/r05_$w.log 2>&1; echo "profile $w rc $?"
done
cp gpurun_out/profiles/traffic_*.json profiles/ 2>/dev/null
for w in fasst 2pl; do
  timeout 500 python bench.py --workload $w > gpurun_out/r05/l4_bench_$w.json 2>/dev/null; echo "bench $w rc $?"
done
echo "== smallbank pass trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.99 smallbank 2>/dev/null | tail -1 > gpurun_out/r05/l4_pass_smallbank.json
python - <<'P'
import json
for w in ("fasst","2pl"):
    d=json.loads(open(f"gpurun_out/r05/l4_bench_{w}.json").read().strip().splitlines()[-1])
    print(w,{k:d.get(k) for k in ("value","ms_per_step","kernels_us")}, d["roofline"].get("frac"), d.get("cpu_baseline",{}).get("value"))
P
