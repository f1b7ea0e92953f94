#!/bin/bash
# r04 dev cycle 4: fused route pack, engine stream priority, kernel times under the exchange
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/dev
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
echo "== route tests"; timeout 600 python -m pytest tests/test_gpu_route.py tests/test_gpu_gdriver.py tests/test_gpu_kv.py -m gpu -x -q --timeout 300 -k "route or exchange or shard or rank or router" 2>&1 | tail -4
echo "== tatp force exchange (engine priority high)"; timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e5 | python -c "$P" || tail -5 gpurun_out/dev/e5
echo "== tatp force exchange (engine priority 0)"; DINT_ENGINE_PRIORITY=0 timeout 300 python bench.py --force-exchange $ARGS 2>gpurun_out/dev/e6 | python -c "$P" || tail -5 gpurun_out/dev/e6
echo "== tatp no exchange"; timeout 300 python bench.py $ARGS 2>gpurun_out/dev/e4 | python -c "$P" || tail -5 gpurun_out/dev/e4
echo "== base force exchange"; (cd gpurun_tmp/base && timeout 300 python bench.py --force-exchange $ARGS 2>/dev/null | python -c "$P")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fx
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fx -o fx -- python $ROOT/bench.py --force-exchange --steps 10 --warmup 2 $ARGS > $ROOT/gpurun_out/dev/fx.log 2>&1
f=$(find /tmp/prof_fx -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -20 "$f" | cut -c1-200
cp "$f" $ROOT/gpurun_out/dev/fx_kernel_stats.csv 2>/dev/null
