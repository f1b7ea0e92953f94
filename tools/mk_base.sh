#!/bin/bash
# a built copy of the committed HEAD under gpurun_tmp/base (git-ignored, travels to the GPU box): the A side of same-box A/B runs
set -eu
cd "$(dirname "$0")/.."
rm -rf gpurun_tmp/base
mkdir -p gpurun_tmp/base
git archive HEAD | tar -x -C gpurun_tmp/base
(cd gpurun_tmp/base && python -m dint_amd.build >/dev/null 2>&1 && ls -la dint_amd/libdint.so)
