"""Diagnostic: time k_fb8 of an OLD build of the library (tools/libmmn_old.bin, ABI 100) against the
current one on the same box, C3 shape, through the raw C ABI."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
from multimodn_amd import hip
import bench
wl = bench.WORKLOADS["c3"]
def run(libpath, old):
    hip._lib = None
    hip.LIB_PATH = libpath
    model = bench.build_model(mm, wl, torch.device("cuda"))
    model.nan_policy = "device"
    eng = model._get_engine(4096)
    xs, y = bench.synthetic_batches(wl, 4096, 4096, seed=1)[0]
    dx = [torch.from_numpy(x).cuda() for x in xs]; dy = torch.from_numpy(y).cuda()
    b = eng.make_batch(dx, dy, [(i, i) for i in range(4)], device_nan_flags=True)
    for _ in range(5):
        eng.local_step(b, 1.0, 0.003, accumulate=True)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.lib.mmn_chain_fwd_bwd(eng._plan, C.byref(b), 1.0, 0.003, st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    print(("old" if old else "new"), "k_fb8 us:", round(float(np.median(ts)), 2))
run(sys.argv[1], sys.argv[2] == "old")
