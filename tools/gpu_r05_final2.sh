#!/bin/bash
# r05 round end, second run (after the lock-table and smallbank work): the whole GPU suite + smoke, rocprofv3 kernel trace and PMC passes (FETCH / WRITE / TCC hit+miss / EA) of all six
# workloads, then the bench lines (they quote the profiles of the same kernel sources), the exchange line, same-box A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05 gpurun_out/profiles
export TMPDIR=/tmp
TAG=${1:-r05}
T0=$(date +%s)
el() { echo "$(( $(date +%s) - T0 )) s"; }
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05/final2_suite.txt; el
echo "== smoke"; timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2 | tee -a gpurun_out/r05/final2_suite.txt
for w in ${PROFILE_WL:-tatp store smallbank}; do   # (fasst / 2pl were profiled after the last edit of k_locks.hip, log is unchanged)
  timeout 600 python tools/profile_bench.py $TAG --workload $w > gpurun_out/profiles/${TAG}_$w.log 2>&1; echo "profile $w rc $? $(el)"
done
cp gpurun_out/profiles/traffic_*.json profiles/ 2>/dev/null   # (on the box: the bench lines below quote them)
grep -h "^# pass" gpurun_out/profiles/${TAG}_*_rocprofv3_summary.txt | cut -c1-110
echo "== default line"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/final2_bench_tatp.json 2> gpurun_out/r05/final2_bench_tatp.err; echo "rc $? $(el)"
echo "== force exchange"; timeout 400 python bench.py --force-exchange --legs gpu --steps 20 --warmup 5 > gpurun_out/r05/final2_bench_tatp_force_exchange.json 2>/dev/null; echo "rc $? $(el)"
for w in store smallbank fasst 2pl log; do
  timeout 500 python bench.py --workload $w > gpurun_out/r05/final2_bench_$w.json 2>/dev/null; echo "bench $w rc $? $(el)"
done
echo "== same-box A/B + knob sweep"; EXP_SWEEP=1 timeout 400 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 > gpurun_out/r05/final2_chain.json; el
echo "== pass trace"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 > gpurun_out/r05/final2_pass_tatp.json; el
python - <<'P'
import json
d=json.loads(open("gpurun_out/r05/final2_bench_tatp.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","value_repeats","parity_failures")})
P
