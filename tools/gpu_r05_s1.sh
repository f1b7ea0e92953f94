#!/bin/bash
# round 5, smallbank resolve: kv tests, smallbank pass trace, short bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_gdriver.py -x -q 2>&1 | tail -4 | tee gpurun_out/r05/s1_tests.txt
echo "== smallbank pass trace"; DINT_KV_TRACE=1 timeout 300 python tools/exp_pass.py 524288 0.99 smallbank 2>/dev/null | tail -1 > gpurun_out/r05/s1_pass_smallbank.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/r05/s1_pass_smallbank.json").read())
print({k:d[k] for k in ('requests_per_pass','wall_us_per_pass','kernels_us')})
print(d.get('phases_us_mean_p95_max')); print(d.get('longest_wgs')[:4])
b=d.get('big_subs',{}); print(b.get('by_size_n_mean_max_us')); print(b.get('stretch_us'))
P
for w in smallbank tatp; do
  timeout 400 python bench.py --workload $w --legs headline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05/s1_bench_$w.json
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1], json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernels_us","parity_failures")}), d.get("latency_us",{}).get("p50"), d.get("latency_us",{}).get("p99"))' gpurun_out/r05/s1_bench_$w.json
done
