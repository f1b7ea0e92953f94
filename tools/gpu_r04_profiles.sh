#!/bin/bash
# r04: rocprofv3 kernel trace + PMC passes of every workload's bench (tools/profile_bench.py) -> gpurun_out/profiles/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
TAG=${1:-r04b}
mkdir -p gpurun_out/profiles
T0=$(date +%s)
timeout 400 python tools/profile_bench.py $TAG > gpurun_out/profiles/${TAG}_tatp.log 2>&1; echo "tatp rc $? $(( $(date +%s) - T0 )) s"
for w in store smallbank log fasst 2pl; do
  timeout 300 python tools/profile_bench.py $TAG --fast --workload $w > gpurun_out/profiles/${TAG}_$w.log 2>&1; echo "$w rc $? $(( $(date +%s) - T0 )) s"
done
ls gpurun_out/profiles
