#!/bin/bash
# round 5, lock tables: every GPU test that touches a lock engine, then the traces and short bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_locks.py tests/test_gpu_async.py tests/test_gpu_route.py tests/test_gpu_shim.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05/l2_tests.txt
bash tools/gpu_r05_l1.sh 2>&1 | grep -v "^batch 6[012]\|passed\|amdgpu.ids\|^\.\.\.\|== lock tests"
