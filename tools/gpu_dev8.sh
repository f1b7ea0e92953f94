#!/bin/bash
# r04 dev cycle 8: repeated dominant-key path, unrolled regrouped gather
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== kv tests"; timeout 800 python -m pytest tests/test_gpu_kv.py tests/test_long_traces.py -m gpu -x -q --timeout 300 2>&1 | tail -3
echo "== tatp trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.8 tatp 2>&1 | tail -1 | cut -c1-1100
echo "== smallbank trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.99 smallbank 2>&1 | tail -1 | cut -c1-800
echo "== tatp base"; (cd gpurun_tmp/base && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P")
echo "== tatp work"; timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
for w in smallbank store; do
  echo "== $w work"; timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
done
