#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
pf() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d.get('parity_failures'), (d.get('inputs_ready') or {}).get('replies_equal'), d.get('value'))
except Exception as e: print('$1 failed', e)"; }
B="--steps 12 --warmup 1 --per-step 16 --legs headline"
HIP_LAUNCH_BLOCKING=1 timeout 100 python bench.py --workload fasst $B 2>/dev/null | pf fasst_lb
HIP_LAUNCH_BLOCKING=1 DINT_LOCK_NO_DIRECT=1 timeout 100 python bench.py --workload fasst $B 2>/dev/null | pf fasst_lb_nodirect
HIP_LAUNCH_BLOCKING=1 DINT_LOCK_NO_FUSE=1 timeout 100 python bench.py --workload fasst $B 2>/dev/null | pf fasst_lb_nofuse
HIP_LAUNCH_BLOCKING=1 timeout 100 python bench.py --workload fasst $B --no-ahead 2>/dev/null | pf fasst_lb_noahead
HIP_LAUNCH_BLOCKING=1 timeout 100 python bench.py --workload 2pl $B 2>/dev/null | pf 2pl_lb
HIP_LAUNCH_BLOCKING=1 timeout 300 python -m pytest tests/test_gpu_locks.py tests/test_gpu_ahead.py -x -q 2>&1 | tail -4
