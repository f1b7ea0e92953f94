#!/bin/bash
# round-end confirmation on one box: full GPU suite, smoke, rocprofv3 profiles of the final sources (copied into
# profiles/ so that the bench lines that follow quote them), the four bench lines, the driver-style 20-step line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02v}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
for w in tatp fasst smallbank; do
  echo "== profile $w"; timeout 900 python tools/profile_bench.py $TAG --workload $w 2>&1 | grep -A7 "^kernel " | cut -c1-190
done
cp gpurun_out/profiles/traffic_*.json profiles/
for w in tatp fasst smallbank store; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; tail -1 "$OUT/bench_$w.err"
  python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('rand64_roofline'), (d.get('cpu_baseline') or {}).get('kind'), (d.get('cpu_baseline') or {}).get('value'))"
done
echo "== driver-style"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_20.json" 2> /dev/null; python -c "import json; d=json.loads(open('$OUT/bench_tatp_20.json').read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['kind'])"
echo "== force-exchange"; timeout 600 python bench.py --force-exchange --no-cpu-baseline --no-rand64 > "$OUT/bench_tatp_force_exchange.json" 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/bench_tatp_force_exchange.json').read()); print(d['value'], d['ms_per_step'])"
du -sh gpurun_out
