#!/bin/bash
# r04 dev cycle 9: the partition kernel beside the previous pass's k_kv_big (DINT_FLAG_INPUTS_READY)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== kv tests"; timeout 800 python -m pytest tests/test_gpu_kv.py tests/test_long_traces.py -m gpu -x -q --timeout 300 2>&1 | tail -3
for w in tatp smallbank store; do
  echo "== $w serial"; DINT_INPUTS_READY=0 timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
  echo "== $w overlapped"; timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
done
echo "== tatp overlapped, 8 queues"; GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
