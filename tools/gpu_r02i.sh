#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout 600 python tools/exp_lock_big.py 1048576 65536 4 2>&1 | tail -4
timeout 600 python tools/exp_lock_big.py 36000000 1048576 4 2>&1 | tail -4
timeout 600 python tools/exp_lock_big.py 1048576 1048576 4 2>&1 | tail -4
