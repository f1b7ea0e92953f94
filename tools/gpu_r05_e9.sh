#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/profiles
export TMPDIR=/tmp
for w in fasst 2pl; do timeout 600 python tools/profile_bench.py r05 --workload $w > gpurun_out/profiles/r05_$w.log 2>&1; echo "profile $w rc $?"; done
grep -h -A5 "^kernel " gpurun_out/profiles/r05_fasst_rocprofv3_summary.txt gpurun_out/profiles/r05_2pl_rocprofv3_summary.txt | cut -c1-190
