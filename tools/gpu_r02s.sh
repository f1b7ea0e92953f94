#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout 900 python -m pytest tests/test_gpu_route.py -x -q -m gpu 2>&1 | tail -30
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 2>&1 | tail -5 | cut -c1-600
