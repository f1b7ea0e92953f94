#!/bin/bash
# r05 call 2: the hot key in pieces -- parity first, then what it does to the chain
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/e2_tests.txt
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 64 2>gpurun_out/r05/e2_chain_tatp.err | tail -1 | tee gpurun_out/r05/e2_chain_tatp.json
tail -5 gpurun_out/r05/e2_chain_tatp.err
echo "== chain tatp, no split"; DINT_KV_NO_SPLIT=1 timeout 300 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e2_chain_tatp_nosplit.json
echo "== pass trace tatp"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 | tee gpurun_out/r05/e2_pass_tatp.json
echo "== store"; timeout 200 python bench.py --workload store --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","kernels_us","latency_us")}))' | tee gpurun_out/r05/e2_store.json
echo "== store no split"; DINT_KV_NO_SPLIT=1 timeout 200 python bench.py --workload store --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","kernels_us","latency_us")}))' | tee gpurun_out/r05/e2_store_nosplit.json
