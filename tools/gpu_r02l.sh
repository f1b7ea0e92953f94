#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
EXP_WL=smallbank EXP_EPOCHS=6 timeout 600 python tools/exp_big.py 524288 0.99 2>&1 | tail -24
