#!/bin/bash
# r04 dev cycle 3: smallbank bitmap ordering A/B, log small-pass tiles, exchange ratio
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"), d.get("exchange"), d.get("pass_1m"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== smallbank / hot tests"; timeout 400 python -m pytest tests/test_gpu_kv.py tests/test_long_traces.py -m gpu -x -q --timeout 200 -k "smallbank or hot or partition or Smallbank or sb" 2>&1 | tail -4
echo "== log tests"; timeout 300 python -m pytest tests -m gpu -x -q --timeout 200 -k "log" 2>&1 | tail -3
echo "== smallbank no-bitmap"; DINT_KV_NO_BM=1 timeout 300 python bench.py --workload smallbank $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
echo "== smallbank bitmap"; timeout 300 python bench.py --workload smallbank $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
echo "== log"; timeout 300 python bench.py --workload log $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
echo "== tatp no exchange"; timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e4 | python -c "$P" || tail -5 gpurun_out/dev/e4
echo "== tatp force exchange"; timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e5 | python -c "$P" || tail -5 gpurun_out/dev/e5
echo "== tatp default with exchange leg"; timeout 300 python bench.py --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-as-shipped 2>gpurun_out/dev/e6 | python -c "$P" || tail -5 gpurun_out/dev/e6
