#!/bin/bash
# r05: the whole GPU suite + smoke + the lines that matter, after the hot-key rewrite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r05/e5_suite.txt
echo "== smoke"; timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3
for w in tatp store; do
  echo "== bench $w"; timeout 600 python bench.py --workload $w --legs headline --steps 12 --warmup 3 2>gpurun_out/r05/e5_bench_$w.err | tail -1 > gpurun_out/r05/e5_bench_$w.json
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernels_us","latency_us","value_repeats")})); print(json.dumps(d.get("roofline")))' gpurun_out/r05/e5_bench_$w.json
done
