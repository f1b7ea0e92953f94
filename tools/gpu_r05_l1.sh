#!/bin/bash
# round 5, lock tables: the lock tests, the big-bin trace of both lock workloads, a short bench line of each
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== lock tests"; timeout 600 python -m pytest tests/test_gpu_locks.py -x -q 2>&1 | tail -6 | tee gpurun_out/r05/l1_tests.txt
echo "== trace 2pl"; EXP_WL=2pl timeout 250 python tools/exp_lock_big.py 1048576 65536 64 2>&1 | tail -4 | tee gpurun_out/r05/l1_lock_big_2pl.txt
echo "== trace fasst"; timeout 250 python tools/exp_lock_big.py 1048576 65536 64 2>&1 | tail -4 | tee gpurun_out/r05/l1_lock_big_fasst.txt
for w in 2pl fasst; do
  echo "== bench $w"; timeout 300 python bench.py --workload $w --legs headline 2>/dev/null | tail -1 > gpurun_out/r05/l1_bench_$w.json
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernels_us","latency_us","replay_matches_recording")}))' gpurun_out/r05/l1_bench_$w.json
done
