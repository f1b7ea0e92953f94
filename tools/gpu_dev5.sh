#!/bin/bash
# r04 dev cycle 5: hardware queues vs streams under the exchange
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== B streams, default queues"; DINT_ROUTER_STREAMS=1 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e1 | python -c "$P" || tail -5 gpurun_out/dev/e1
echo "== C streams, 8 queues"; GPU_MAX_HW_QUEUES=8 DINT_ROUTER_STREAMS=1 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e2 | python -c "$P" || tail -5 gpurun_out/dev/e2
echo "== D set, 8 queues"; GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e3 | python -c "$P" || tail -5 gpurun_out/dev/e3
echo "== E no exchange, 8 queues"; GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e4 | python -c "$P" || tail -5 gpurun_out/dev/e4
echo "== F streams, 16 queues"; GPU_MAX_HW_QUEUES=16 DINT_ROUTER_STREAMS=1 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e5 | python -c "$P" || tail -5 gpurun_out/dev/e5
GPU_MAX_HW_QUEUES=8 DINT_ROUTER_STREAMS=1 python tools/fx_timeline.py > gpurun_out/dev/fx_stdout3.txt 2>&1; cp gpurun_out/dev/fx_timeline.txt gpurun_out/dev/fx_timeline_streams8.txt
