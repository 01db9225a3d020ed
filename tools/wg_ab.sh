cd $GRAFT_REPO_ROOT
for v in "" "MMN_WGRAD_MT=2" "MMN_WGRAD_NT=2" "MMN_WGRAD_ROWS=256" "MMN_WGRAD_ROWS=1024 MMN_WGRAD_MT=2" "MMN_WGRAD_ROWS=1024 MMN_WGRAD_MT=2 MMN_WGRAD_NT=2" "MMN_WGRAD_ROWS=1024"; do
  echo "== $v"; env $v MMN_VERBOSE=1 python tools/time_kernels.py c3 2>&1 | grep -E "step|wgrad items" | tail -2
done
