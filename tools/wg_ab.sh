cd $GRAFT_REPO_ROOT
python tools/stamps.py 2>&1 | tail -1
python tools/time_kernels.py c3 2>&1 | tail -1
python tools/time_kernels.py c3 2>&1 | tail -1
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
