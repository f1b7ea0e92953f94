#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_gdriver.py -x -q 2>&1 | tail -6 | tee gpurun_out/r05/e4_tests.txt
echo "== pass trace tatp"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 > gpurun_out/r05/e4_pass_tatp.json
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e4_chain_tatp.json
echo "== chain tatp one big kernel"; DINT_KV_ONE_BIG_KERNEL=1 timeout 300 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e4_chain_tatp_onebig.json
echo "== store"; timeout 200 python bench.py --workload store --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","kernels_us","latency_us")}))' | tee gpurun_out/r05/e4_store.json
