#!/bin/bash
# r05 call 1: sanity of the host-boundary / routing changes, what bounds the chain, the stale-bytes report taken apart
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
export TMPDIR=/tmp
echo "== tests"; timeout 400 python -m pytest tests/test_gpu_async.py tests/test_gpu_route.py -x -q 2>&1 | tail -5 | tee gpurun_out/r05/e1_tests.txt
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 64 2>gpurun_out/r05/e1_chain_tatp.err | tail -1 | tee gpurun_out/r05/e1_chain_tatp.json
echo "== pass trace tatp"; DINT_KV_TRACE=1 timeout 200 python tools/exp_pass.py 524288 0.8 tatp 2>/dev/null | tail -1 | tee gpurun_out/r05/e1_pass_tatp.json
echo "== stress, direct pageable copies (r04) under rocprofv3"
(cd /tmp && DINT_NO_BOUNCE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/sp_a -o a -- python "$ROOT/tools/stress_pageable.py" 40 2>/dev/null | grep '^{' | tail -1) | tee gpurun_out/r05/e1_stress_nobounce.json
echo "== stress, bounce buffers under rocprofv3"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/sp_b -o b -- python "$ROOT/tools/stress_pageable.py" 40 2>/dev/null | grep '^{' | tail -1) | tee gpurun_out/r05/e1_stress_bounce.json
echo "== store kernels"; timeout 200 python bench.py --workload store --no-cpu-baseline --no-mixes --no-rand64 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","kernels_us","latency_us")}))' | tee gpurun_out/r05/e1_store.json
