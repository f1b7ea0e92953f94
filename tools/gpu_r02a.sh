#!/bin/bash
# r02 GPU session A: new multi-GPU / async tests first, then the whole gpu suite, then bench lines.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p "$OUT"
cd "$ROOT"
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_async.py -x -q 2>&1 | tail -25 | tee "$OUT/new_tests.log"
echo "== bench tatp (driver form)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_20.json" 2> "$OUT/bench_tatp_20.err"; tail -3 "$OUT/bench_tatp_20.err"; cat "$OUT/bench_tatp_20.json"
echo "== bench tatp --gpus 2 on one GPU (gloo / host-staged exchange)"; timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --clients 65536 --no-cpu-baseline > "$OUT/bench_tatp_g2.json" 2> "$OUT/bench_tatp_g2.err"; tail -5 "$OUT/bench_tatp_g2.err"; cat "$OUT/bench_tatp_g2.json"
echo "== bench tatp --force-exchange"; timeout 600 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline > "$OUT/bench_tatp_fx.json" 2> "$OUT/bench_tatp_fx.err"; tail -3 "$OUT/bench_tatp_fx.err"; cat "$OUT/bench_tatp_fx.json"
echo "== bench fasst 36M"; timeout 600 python bench.py --workload fasst --slots 36000000 --steps 50 > "$OUT/bench_fasst36.json" 2> "$OUT/bench_fasst36.err"; tail -3 "$OUT/bench_fasst36.err"; cat "$OUT/bench_fasst36.json"
echo "== bench store"; timeout 600 python bench.py --workload store --steps 50 > "$OUT/bench_store.json" 2> "$OUT/bench_store.err"; tail -3 "$OUT/bench_store.err"; cat "$OUT/bench_store.json"
echo "== bench smallbank"; timeout 900 python bench.py --workload smallbank --steps 50 > "$OUT/bench_sb.json" 2> "$OUT/bench_sb.err"; tail -3 "$OUT/bench_sb.err"; cat "$OUT/bench_sb.json"
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest.log"
echo "== done"
