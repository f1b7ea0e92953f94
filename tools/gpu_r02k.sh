#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
timeout 900 python tools/profile_bench.py r02k --sq 2>&1 | tail -12
