#!/bin/bash
# what differs when the smallbank replay diverges from the recorded run under rocprofv3?  (dumps the batch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/dev
cd /tmp && export TMPDIR=/tmp
ARGS="--workload smallbank --steps 8 --warmup 1 --no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
for i in 1 2 3 4; do
  rm -rf /tmp/pb
  DINT_DUMP_DIVERGENCE=$ROOT/gpurun_out/dev/diverge.npz timeout 100 rocprofv3 --kernel-trace -d /tmp/pb -o x -- python $ROOT/bench.py $ARGS > /tmp/pb.out 2> /tmp/pb.err
  rc=$?
  echo "run $i rc=$rc $(grep 'AssertionError' /tmp/pb.err | tail -1 | cut -c1-300) $(grep -o '"value": [0-9.]*' /tmp/pb.out | head -1)"
  [ $rc -ne 0 ] && break
done
ls -la $ROOT/gpurun_out/dev/diverge.npz
