#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_us"], d.get("value_repeats"))'
for i in 1 2; do
  echo "== r01 tree #$i"; (cd gpurun_tmp/r01 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 2>/dev/null | python -c "$P")
  echo "== current q=4 one-stream #$i"; DINT_ONE_STREAM=1 GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
  echo "== current q=12 one-stream #$i"; DINT_ONE_STREAM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
  echo "== current q=4 #$i"; GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rand64 --no-host-path 2>/dev/null | python -c "$P"
done
