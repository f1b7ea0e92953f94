#!/bin/bash
# r04 dev cycle 6: smallbank stretch timeline; set-mode knobs under the exchange
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== smallbank tests"; timeout 400 python -m pytest tests/test_gpu_kv.py -m gpu -x -q --timeout 200 -k "smallbank or hot or partition" 2>&1 | tail -2
echo "== smallbank trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.99 smallbank 2>&1 | tail -1
echo "== tatp trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.8 tatp 2>&1 | tail -1
echo "== fx rpt1"; DINT_KV_RPT=1 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
echo "== fx load768"; DINT_KV_COARSE_LOAD=768 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
echo "== fx load1024 rpt1"; DINT_KV_RPT=1 DINT_KV_COARSE_LOAD=1024 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
