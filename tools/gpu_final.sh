#!/bin/bash
# round-end confirmation on one box: full GPU suite, smoke, rocprofv3 profiles of the final sources (copied into
# profiles/ on the box so that the bench lines that follow quote them), the bench lines.  Everything lands in
# gpurun_out/<tag>/ (small files only: profile_bench.py removes its raw databases).
#   usage: gpu_final.sh <tag> [suite] [profiles] [bench] [extra]     (no selector = all four parts)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
shift || true
PARTS=${*:-suite profiles bench extra}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d.get("metric"), d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"),
      "lat", (d.get("latency_us") or {}), "closed", (d.get("closed_loop") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("kind"),
      (d.get("cpu_baseline") or {}).get("value"), "parity_failures", d.get("parity_failures"))
PY
}
if has suite; then
  echo "== gpu suite"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pytest.log"
  echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee "$OUT/smoke.log"
fi
if has profiles; then
  for w in tatp fasst smallbank 2pl; do
    echo "== profile $w"; timeout 1400 python tools/profile_bench.py $TAG --workload $w 2>&1 | grep -A6 "^kernel " | cut -c1-190
  done
  cp gpurun_out/profiles/traffic_*.json profiles/ 2>/dev/null
fi
if has bench; then
  echo "== bench (default line, as the driver runs it)"; timeout 900 python bench.py > "$OUT/bench_tatp.json" 2> "$OUT/bench_tatp.err"; line "$OUT/bench_tatp.json"
  for w in fasst 2pl log store smallbank; do
    echo "== bench $w"; timeout 900 python bench.py --workload $w --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; line "$OUT/bench_$w.json"
  done
fi
if has extra; then
  echo "== driver-style"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_tatp_driver_style.json" 2> /dev/null; line "$OUT/bench_tatp_driver_style.json"
  echo "== nurand"; timeout 600 python bench.py --theta 0 --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped --no-cpu-baseline > "$OUT/bench_tatp_nurand.json" 2>/dev/null; line "$OUT/bench_tatp_nurand.json"
  echo "== force-exchange"; timeout 600 python bench.py --force-exchange --no-cpu-baseline --no-rand64 --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped > "$OUT/bench_tatp_force_exchange.json" 2>/dev/null; line "$OUT/bench_tatp_force_exchange.json"
  echo "== client sweep"; timeout 900 python bench.py --sweep-clients --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped --no-cpu-baseline > "$OUT/bench_tatp_sweep_clients.json" 2>/dev/null; line "$OUT/bench_tatp_sweep_clients.json"
fi
du -sh gpurun_out
