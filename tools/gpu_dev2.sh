#!/bin/bash
# r04 dev cycle: parity of the variants, the phase timeline of the resolve workgroups, A/B of the variants against base
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
echo "== kv tests (default = LATE, 512)"; timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_async.py tests/test_ebpf_golden.py tests/test_ebpf_surface.py tests/test_long_traces.py -m gpu -x -q --timeout 300 2>&1 | tail -4
echo "== kv tests, WG 256"; DINT_KV_WG=256 timeout 600 python -m pytest tests/test_gpu_kv.py -m gpu -x -q --timeout 300 2>&1 | tail -3
echo "== kv tests, in place"; DINT_KV_LATE_BIG=0 timeout 600 python -m pytest tests/test_gpu_kv.py -m gpu -x -q --timeout 300 2>&1 | tail -3
for th in 0.01 0.8; do
  echo "== trace theta $th"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
  echo "== plain theta $th default"; timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
  echo "== plain theta $th no-prefetch"; DINT_KV_NO_PREFETCH=1 timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
  echo "== plain theta $th wg256"; DINT_KV_WG=256 timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
  echo "== plain theta $th in-place"; DINT_KV_LATE_BIG=0 timeout 300 python tools/exp_pass.py 524288 $th 2>&1 | tail -1
done
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim"
echo "== tatp base"; (cd gpurun_tmp/base && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P")
echo "== tatp default"; timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
echo "== tatp no-prefetch"; DINT_KV_NO_PREFETCH=1 timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
echo "== tatp wg256"; DINT_KV_WG=256 timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
echo "== tatp in-place"; DINT_KV_LATE_BIG=0 timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
for w in store smallbank; do
  echo "== $w default"; timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
  echo "== $w wg256"; DINT_KV_WG=256 timeout 300 python bench.py --workload $w $ARGS 2>gpurun_out/dev/e4 | python -c "$P" || tail -5 gpurun_out/dev/e4
done
