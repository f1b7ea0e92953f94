cd $GRAFT_REPO_ROOT
for e in "DINT_X=1" "DINT_KV_NO_FUSE=1" "DINT_KV_SPLIT_TARGET=200" "DINT_KV_SPLIT_TARGET=300" "DINT_KV_WORKERS=192"; do
  echo "== $e"; env $e python -m pytest tests/test_gpu_route.py -q -k half_filled 2>&1 | grep -E "^E  +Assert|passed|failed" | head -6
done
