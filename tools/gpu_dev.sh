#!/bin/bash
# development cycle on one box: kv-related GPU tests first (fail fast), then the per-kernel numbers of one engine alone and a
# same-box A/B of the bench against gpurun_tmp/base (tools/mk_base.sh).   usage: gpu_dev.sh [tests] [pass] [ab] [others] [full]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
PARTS=${*:-tests pass ab}
OUT=$ROOT/gpurun_out/dev
mkdir -p "$OUT"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("kernels_us"), d.get("value_repeats"), d.get("latency_us"), "pf", d.get("parity_failures"))'
if has tests; then
  echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py tests/test_gpu_async.py tests/test_ebpf_golden.py tests/test_ebpf_surface.py tests/test_long_traces.py -m gpu -x -q --timeout 300 2>&1 | tail -15
fi
if has full; then
  echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -15
fi
if has pass; then
  for a in "524288 0.01" "524288 0.8" "1048576 0.01"; do
    echo "== exp_pass $a: base"; (cd gpurun_tmp/base && timeout 300 python tools/exp_pass.py $a 2>&1 | tail -1)
    echo "== exp_pass $a: work"; timeout 300 python tools/exp_pass.py $a 2>&1 | tail -1
  done
fi
ARGS="--no-cpu-baseline --no-rand64 --no-host-path --no-closed-loop --no-other-workloads --no-shim --no-exchange-leg --no-as-shipped"
if has ab; then
  for i in 1 2; do
    echo "== tatp base #$i"; (cd gpurun_tmp/base && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "$P")
    echo "== tatp work #$i"; timeout 300 python bench.py $ARGS 2>"$OUT/bench_tatp.err" | python -c "$P" || tail -5 "$OUT/bench_tatp.err"
  done
fi
if has others; then
  for w in store smallbank; do
    echo "== $w base"; (cd gpurun_tmp/base && timeout 300 python bench.py --workload $w $ARGS 2>/dev/null | python -c "$P")
    echo "== $w work"; timeout 300 python bench.py --workload $w $ARGS 2>"$OUT/bench_$w.err" | python -c "$P" || tail -5 "$OUT/bench_$w.err"
  done
fi
