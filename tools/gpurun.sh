#!/bin/bash
# gpurun with the commit id on board: the GPU box receives a snapshot without .git, so the id of HEAD (+ "-dirty" when the
# tree has uncommitted changes) travels as .commit_id (git-ignored) for tools/profile_bench.py to quote.
#   tools/gpurun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
id=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' 2>/dev/null || id="$id-dirty"
echo "$id" > .commit_id
exec /usr/local/graft/bin/gpurun "$@"
