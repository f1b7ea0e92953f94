#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== default bench line (as the driver runs it)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05/e6_bench_default.json 2> gpurun_out/r05/e6_bench_default.err
echo rc=$?; tail -3 gpurun_out/r05/e6_bench_default.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r05/e6_bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","value_repeats","goodput_Mtxn_s","latency_us","kernels_us","parity_failures")})
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","kernel_avg_us","achieved","frac")}, d["roofline"].get("pass"), d["roofline"].get("gpu"))
print("rand64", d.get("roofline_rand64",{}).get("frac"), "closed", (d.get("closed_loop") or {}).get("value"), "pcie", d.get("value_pcie"), d.get("latency_host_us"))
print("exchange", d.get("exchange"))
for k,v in (d.get("other_workloads") or {}).items(): print(k, v.get("value"), v.get("kernels_us"), v.get("pass_1m"), v.get("oracle_parity"), v.get("error"))
print("cpu", {k:(d.get("cpu_baseline") or {}).get(k) for k in ("value","kind","cores")}, (d.get("cpu_as_shipped_tatp") or {}).get("value"))
P
