#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/e4_tests.txt
echo "== pass trace tatp"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 | tee gpurun_out/r05/e4_pass_tatp.json
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e4_chain_tatp.json
