#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05 gpurun_out/dev
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_epoch","host_issue_ms_per_step","latency_us","route_overflow","kernels_us","value_repeats")}))'
echo "== kv tests"; timeout 900 python -m pytest tests/test_gpu_kv.py -x -q 2>&1 | tail -3
echo "== chain tatp"; timeout 300 python tools/exp_chain.py 524288 0.8 tatp 48 2>/dev/null | tail -1 | tee gpurun_out/r05/e7_chain_tatp.json
echo "== exchange, launch set"; timeout 300 python bench.py --force-exchange --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "$P"
echo "== exchange, engine streams"; DINT_ROUTER_STREAMS=1 timeout 300 python bench.py --force-exchange --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "$P"
echo "== store"; timeout 200 python bench.py --workload store --legs headline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "$P"
