#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel trace + HBM counters.
# Usage (from the repo root, on the GPU box): bash tools/gpu_check.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest.log"
echo "== bench" ; timeout 900 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -3 "$OUT/bench.err"; cat "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-rand64 > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
python "$ROOT/tools/pmc_summary.py" --last ${DINT_TIMED_LAUNCHES:-330} "$OUT/trace" | tee "$OUT/trace_summary.txt"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  echo "== rocprof pmc $C"
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$N" -o pmc -- python "$ROOT/bench.py" "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-rand64 > /dev/null 2> "$OUT/pmc_$N.err"
  python "$ROOT/tools/pmc_summary.py" --last ${DINT_PMC_LAUNCHES:-135} "$OUT/pmc_$N" > "$OUT/pmc_$N.txt" 2>&1; cat "$OUT/pmc_$N.txt"
done
echo "== done"
