#!/bin/bash
# r04 dev cycle: device selftest, parity, phase timeline, A/B against base
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
echo "== hot-key tests first (a hang must show early)"; timeout 200 python -m pytest tests/test_gpu_kv.py -m gpu -x -q --timeout 60 -k "million or hot or partition" 2>&1 | tail -4
echo "== kv + lock tests"; timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_locks.py tests/test_gpu_async.py tests/test_ebpf_golden.py tests/test_ebpf_surface.py tests/test_long_traces.py tests/test_gpu_gdriver.py tests/test_gpu_route.py -m gpu -x -q --timeout 300 2>&1 | tail -4
for th in 0.8; do
  echo "== plain theta $th"; timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
done
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
for i in 1 2; do
echo "== tatp base"; (cd gpurun_tmp/base && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P")
echo "== tatp work"; timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
done
for w in store smallbank fasst; do
  echo "== $w base"; (cd gpurun_tmp/base && timeout 300 python bench.py --workload $w $ARGS 2>/dev/null | python -c "$P")
  echo "== $w work"; timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
done
