#!/bin/bash
# same-box A/B: gpurun_tmp/base (tools/mk_base.sh: the committed HEAD) against the working tree, alternating
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
ARGS=${AB_ARGS:---no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop}
N=${AB_N:-3}
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"))'
for i in $(seq 1 $N); do
  echo "== base #$i"; (cd gpurun_tmp/base && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P")
  echo "== work #$i"; timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P"
done
