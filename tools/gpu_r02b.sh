#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b
mkdir -p "$OUT"
cd "$ROOT"
echo "== 24M trace + route tests"; timeout 1200 python -m pytest tests/test_fasst_24m.py tests/test_gpu_route.py -x -q -m gpu 2>&1 | tail -15 | tee "$OUT/tests.log"
echo "== bench tatp (driver form)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_tatp_20.json" 2> "$OUT/bench_tatp_20.err"; tail -3 "$OUT/bench_tatp_20.err"; cat "$OUT/bench_tatp_20.json"
echo "== bench tatp --force-exchange (pipelined)"; timeout 600 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline > "$OUT/bench_tatp_fx.json" 2> "$OUT/bench_tatp_fx.err"; tail -3 "$OUT/bench_tatp_fx.err"; cat "$OUT/bench_tatp_fx.json"
echo "== bench tatp --gpus 2 on one GPU"; timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --clients 65536 --no-cpu-baseline > "$OUT/bench_tatp_g2.json" 2> "$OUT/bench_tatp_g2.err"; tail -3 "$OUT/bench_tatp_g2.err"; cat "$OUT/bench_tatp_g2.json"
echo "== bench fasst closed loop 36M"; timeout 600 python bench.py --workload fasst --slots 36000000 --steps 50 > "$OUT/bench_fasst36.json" 2> "$OUT/bench_fasst36.err"; tail -3 "$OUT/bench_fasst36.err"; cat "$OUT/bench_fasst36.json"
echo "== bench fasst --gpus 2 on one GPU"; timeout 600 python bench.py --workload fasst --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_fasst_g2.json" 2> "$OUT/bench_fasst_g2.err"; tail -3 "$OUT/bench_fasst_g2.err"; cat "$OUT/bench_fasst_g2.json"
echo "== profile tatp"; timeout 1500 python tools/profile_bench.py r02b --workload tatp 2>&1 | tail -30
echo "== profile tatp force-exchange (kernel trace only)"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/fx_trace" -o fx -- python "$ROOT/bench.py" --steps 40 --warmup 5 --force-exchange --no-cpu-baseline --no-rand64 --no-host-path > /dev/null 2> "$OUT/fx_trace.err"
python "$ROOT/tools/pmc_summary.py" --last 135 "$OUT/fx_trace" 2>&1 | tail -40 | tee "$OUT/fx_trace_summary.txt"
echo "== done"
