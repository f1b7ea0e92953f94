#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for d in 0 1 2 3; do DINT_TXN_DBG=$d timeout 120 python tools/exp_emit.py 2>&1 | tail -1; done
