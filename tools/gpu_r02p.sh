#!/bin/bash
# round-2 confirmation run: full GPU suite, smoke, the default bench (with the reference CPU legs), fasst bench, profiles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02p
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench default"; ( time timeout 900 python bench.py > "$OUT/bench_tatp.json" 2> "$OUT/bench_tatp.err" ) 2>&1 | grep real; tail -2 "$OUT/bench_tatp.err"; cat "$OUT/bench_tatp.json"
echo "== bench fasst"; ( time timeout 600 python bench.py --workload fasst > "$OUT/bench_fasst.json" 2> "$OUT/bench_fasst.err" ) 2>&1 | grep real; tail -2 "$OUT/bench_fasst.err"; cat "$OUT/bench_fasst.json"
echo "== bench smallbank"; timeout 600 python bench.py --workload smallbank > "$OUT/bench_smallbank.json" 2> "$OUT/bench_smallbank.err"; tail -1 "$OUT/bench_smallbank.err"; cut -c1-400 "$OUT/bench_smallbank.json"
echo "== bench store"; timeout 600 python bench.py --workload store > "$OUT/bench_store.json" 2> "$OUT/bench_store.err"; tail -1 "$OUT/bench_store.err"; cut -c1-400 "$OUT/bench_store.json"
echo "== profile tatp"; timeout 900 python tools/profile_bench.py r02p 2>&1 | tail -5
echo "== profile fasst"; timeout 600 python tools/profile_bench.py r02p --workload fasst 2>&1 | tail -5
echo "== profile smallbank"; timeout 600 python tools/profile_bench.py r02p --workload smallbank 2>&1 | tail -5
ls gpurun_out/profiles; du -sh gpurun_out
