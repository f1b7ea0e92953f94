#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02d
mkdir -p "$OUT"
cd "$ROOT"
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_route.py tests/test_ebpf_surface.py tests/test_gpu_gdriver.py tests/test_gpu_shim.py -x -q -m gpu 2>&1 | tail -30 | tee "$OUT/tests.log"
echo "== bench tatp --force-exchange (pipelined, staged route kernels)"; timeout 600 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline > "$OUT/bench_tatp_fx.json" 2> "$OUT/bench_tatp_fx.err"; tail -3 "$OUT/bench_tatp_fx.err"; python -c "import json;d=json.load(open('$OUT/bench_tatp_fx.json'));print(d['value'],d['ms_per_step'],d['value_repeats'])"
echo "== closed loop"; python tools/exp_closed.py tatp 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/cl_trace" -o cl -- python "$ROOT/tools/exp_closed.py" tatp > "$OUT/cl.log" 2>&1
python "$ROOT/tools/pmc_summary.py" "$OUT/cl_trace" 2>&1 | head -14
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/fx_trace" -o fx -- python "$ROOT/bench.py" --steps 40 --warmup 5 --force-exchange --no-cpu-baseline --no-rand64 --no-host-path > /dev/null 2> "$OUT/fx_trace.err"
python "$ROOT/tools/pmc_summary.py" --last 135 "$OUT/fx_trace" 2>&1 | tail -14
cd "$ROOT"
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest.log"
