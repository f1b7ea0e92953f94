#!/bin/bash
# r05 call 2: the hot key in pieces -- parity first, then what it does to the chain
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/e2_tests.txt
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 64 2>gpurun_out/r05/e2_chain_tatp.err | tail -1 | tee gpurun_out/r05/e2_chain_tatp.json
tail -5 gpurun_out/r05/e2_chain_tatp.err
echo "== chain tatp, no split"; DINT_KV_NO_SPLIT=1 timeout 300 python tools/exp_chain.py 524288 0.8 tatp 64 2>/dev/null | tail -1 | tee gpurun_out/r05/e2_chain_tatp_nosplit.json
echo "== pass trace tatp"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 | tee gpurun_out/r05/e2_pass_tatp.json
