cd $GRAFT_REPO_ROOT
python tools/stamps.py 2>&1 | grep k_wgrad
python tools/time_kernels.py c3 2>&1 | tail -1
python tools/time_kernels.py c3 2>&1 | tail -1
python tools/time_kernels.py c2 2>&1 | tail -1
