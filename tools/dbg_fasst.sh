#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/dbg
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --workload fasst --steps 12 --warmup 1 --per-step 16 --legs gpu"
for p in "FETCH_SIZE" "WRITE_SIZE"; do
  s=$(date +%s)
  timeout -s INT 90 rocprofv3 --pmc $p --kernel-trace -d /tmp/dbg_$p -o t -- $B > $R/gpurun_out/dbg/$p.out 2> $R/gpurun_out/dbg/$p.err
  echo "pmc $p: rc $? in $(( $(date +%s) - s )) s"
  python - <<PY
import json
try:
    d=json.loads(open("$R/gpurun_out/dbg/$p.out").read().strip().splitlines()[-1]); print(d.get("parity_failures"), (d.get("inputs_ready") or {}).get("replies_equal"), d.get("value"))
except Exception as e: print("no line", e)
PY
  grep -v "simple_timer\|generateRocpd\|tool.cpp" $R/gpurun_out/dbg/$p.err | tail -3 | cut -c1-300
done
