#!/bin/bash
# r04 final: the two profile passes that failed (tatp, smallbank), then the checkpoint (suite, smoke, default bench)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
T0=$(date +%s)
timeout 400 python tools/profile_bench.py r04b > gpurun_out/profiles/r04b_tatp.log 2>&1; echo "tatp rc $? $(( $(date +%s) - T0 )) s"
timeout 400 python tools/profile_bench.py r04b --fast --workload smallbank > gpurun_out/profiles/r04b_smallbank.log 2>&1; echo "smallbank rc $? $(( $(date +%s) - T0 )) s"
timeout 300 python tools/profile_bench.py r04b --fast --workload store > gpurun_out/profiles/r04b_store.log 2>&1; echo "store rc $? $(( $(date +%s) - T0 )) s"
grep "^# pass" gpurun_out/profiles/r04b_tatp_rocprofv3_summary.txt gpurun_out/profiles/r04b_smallbank_rocprofv3_summary.txt | cut -c1-120
cp gpurun_out/profiles/traffic_tatp.json gpurun_out/profiles/traffic_smallbank.json gpurun_out/profiles/traffic_store.json profiles/   # the bench below quotes them
./tools/gpu_r04.sh
