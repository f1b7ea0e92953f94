#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/last
cd "$ROOT"
timeout 200 python bench.py --theta 0 --no-cpu-baseline --no-host-path > gpurun_out/last/bench_tatp_nurand.json 2>/dev/null
python -c "import json; d=json.loads(open('gpurun_out/last/bench_tatp_nurand.json').read()); print(d['value'], d['ms_per_step'], d['latency_us'], d['kernels_us'], d['closed_loop']['value'], d.get('ops_frac_of_rand64'))"
