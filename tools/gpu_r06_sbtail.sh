#!/bin/bash
# r06: the smallbank tail as evidence -- exp_sb_tail.py over 400 epochs with the defaults and with the round's starting point
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/dev
{ echo "# tools/exp_sb_tail.py 400 (final tree, defaults)"
  timeout 300 python tools/exp_sb_tail.py 400 2>&1 | grep -v amdgpu | cut -c1-400
  echo "# DINT_KV_SB_WORKERS=0 DINT_KV_SB_SPLIT_MIN=2048 (how the second half of the round started: pieces from 2,048 requests, every item in k_kv_big)"
  DINT_KV_SB_WORKERS=0 DINT_KV_SB_SPLIT_MIN=2048 timeout 300 python tools/exp_sb_tail.py 400 2>&1 | grep -v amdgpu | cut -c1-400
} > gpurun_out/dev/sb_tail.txt
head -c 2500 gpurun_out/dev/sb_tail.txt
