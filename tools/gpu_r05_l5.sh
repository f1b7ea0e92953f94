#!/bin/bash
# round 5, lock passes in two halves (DINT_FLAG_INPUTS_READY): lock tests, async / route tests, bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_locks.py tests/test_gpu_async.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05/l5_tests.txt
for w in fasst 2pl; do
  echo "== bench $w"; timeout 300 python bench.py --workload $w --legs headline 2>/dev/null | tail -1 > gpurun_out/r05/l5_bench_$w.json
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernels_us","latency_us","replay_equals_recorded","inputs_ready")}))' gpurun_out/r05/l5_bench_$w.json
done
