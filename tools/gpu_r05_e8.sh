#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== chain tatp + sweep"; EXP_SWEEP=1 timeout 400 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e8_chain_sweep.json
