#!/bin/bash
# bench lines re-run with the committed profiles in place (roofline.traffic quoted), and the N = 2 path on one GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02q
mkdir -p "$OUT"
cd "$ROOT"
for w in tatp fasst smallbank; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; tail -1 "$OUT/bench_$w.err"
  python -c "import json,sys; d=json.loads(open('$OUT/bench_$w.json').read()); print(d['value'], d['roofline'], d.get('rand64_roofline'))"
done
echo "== driver-style run"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_20.json" 2> "$OUT/bench_tatp_20.err"; python -c "import json; d=json.loads(open('$OUT/bench_tatp_20.json').read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"
echo "== --gpus 2 on one GPU (gloo, exchange staged through the host)"; timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 > "$OUT/bench_tatp_gpus2_onegpu.json" 2> "$OUT/bench_gpus2.err"; tail -2 "$OUT/bench_gpus2.err"; cut -c1-900 "$OUT/bench_tatp_gpus2_onegpu.json"
echo "== force-exchange"; timeout 600 python bench.py --force-exchange --no-cpu-baseline --no-rand64 > "$OUT/bench_tatp_force_exchange.json" 2> "$OUT/fx.err"; tail -1 "$OUT/fx.err"; cut -c1-700 "$OUT/bench_tatp_force_exchange.json"
