cd $GRAFT_REPO_ROOT
for wl in c1 c2; do
  echo "== $wl"; MMN_VERBOSE=1 python tools/time_kernels.py $wl 2>&1 | grep -E "step|wgrad items" | tail -2
done
for v in "MMN_WGRAD_ROWS=768" "MMN_WGRAD_ROWS=1024" "MMN_WGRAD_ROWS=400"; do
  echo "== mimic $v"; env $v MMN_VERBOSE=1 python tools/time_kernels.py mimic 2>&1 | grep -E "step|wgrad items" | tail -2
done
