#!/bin/bash
# bin-load sweep of the kv passes (DINT_KV_BIN_LOAD = records per bin on average; the bins are any number, not a power of two)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("parity_failures"))'
for w in ${EXP_WL:-tatp}; do
  for L in ${EXP_LOADS:-26 34 40 46}; do
    echo "== $w load $L"; DINT_KV_BIN_LOAD=$L timeout 300 python bench.py --workload $w $ARGS 2>/dev/null | python -c "$P"
  done
done
