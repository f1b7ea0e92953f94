#!/bin/bash
# round 5, lock tables: threshold sweep of the dominant-slot path
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for hm in 65 96 128 192 256; do
  for w in 2pl fasst; do
    DINT_LOCK_HOT_MIN=$hm timeout 300 python bench.py --workload $w --legs headline 2>/dev/null | tail -1 > /tmp/l3.json
    python -c 'import sys,json; d=json.loads(open("/tmp/l3.json").read()); print(sys.argv[1], sys.argv[2], d["value"], d["kernels_us"], d["latency_us"]["p50"])' $hm $w | tee -a gpurun_out/r05/l3_sweep.txt
  done
done
