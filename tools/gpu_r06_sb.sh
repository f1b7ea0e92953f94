#!/bin/bash
# r06: smallbank's tail -- the smallbank tests, then per DINT_KV_SB_SPLIT_MIN the slow epochs (exp_sb_tail.py) and the bench's headline leg
#   tools/gpurun.sh --timeout 1500 -- 'bash tools/gpu_r06_sb.sh 2048 384 128'
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_ahead.py -k smallbank -x -q 2>&1 | tail -5
for m in "$@"; do
  echo "== SB_SPLIT_MIN $m"
  DINT_KV_SB_SPLIT_MIN=$m timeout 300 python tools/exp_sb_tail.py 200 2>&1 | grep -v "engine\|amdgpu" | cut -c1-100 | head -6
  DINT_KV_SB_SPLIT_MIN=$m timeout 300 python bench.py --workload smallbank --legs headline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_repeats'], d['kernels_us'], d['latency_us']['p50'], d['latency_us']['p99'], d['late'])"
done
