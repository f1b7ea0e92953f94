#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02c
mkdir -p "$OUT"
cd "$ROOT"
echo "== new tests"; timeout 1500 python -m pytest tests/test_gpu_gdriver.py tests/test_ebpf_surface.py tests/test_gpu_shim.py tests/test_gpu_async.py tests/test_gpu_route.py -x -q -m gpu 2>&1 | tail -30 | tee "$OUT/tests.log"
echo "== bench tatp (driver form)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_20.json" 2> "$OUT/bench_tatp_20.err"; tail -3 "$OUT/bench_tatp_20.err"; cat "$OUT/bench_tatp_20.json"
echo "== bench tatp 100"; timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_tatp_100.json" 2> "$OUT/bench_tatp_100.err"; tail -3 "$OUT/bench_tatp_100.err"; cat "$OUT/bench_tatp_100.json"
echo "== bench tatp --force-exchange (pipelined)"; timeout 600 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline > "$OUT/bench_tatp_fx.json" 2> "$OUT/bench_tatp_fx.err"; tail -3 "$OUT/bench_tatp_fx.err"; cat "$OUT/bench_tatp_fx.json"
echo "== bench smallbank"; timeout 900 python bench.py --workload smallbank --steps 30 --no-cpu-baseline > "$OUT/bench_sb.json" 2> "$OUT/bench_sb.err"; tail -3 "$OUT/bench_sb.err"; cat "$OUT/bench_sb.json"
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest.log"
echo "== done"
