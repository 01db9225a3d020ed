"""Diagnostic: k_xpart's stamps (MMN_STAMPS=1) of the reference's real MIMIC shape with / without the xin stores (training /
forward-only launch) and with / without dropout multipliers.  Usage: python tools/stamps_xpart.py"""
import ctypes as C, os, sys
os.environ["MMN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench

for drop in (0.2, 0.0):
    wl = dict(bench.WORKLOADS["haim"]); wl["dropout"] = drop
    B = wl["B"]
    model = bench.build_model(mm, wl, torch.device("cuda"))
    model.nan_policy = "device"
    xs, y = bench.synthetic_batches(wl, B, B, seed=1)[0]
    eng = model._get_engine(B)
    dx = [torch.from_numpy(x).cuda() for x in xs]; dy = torch.from_numpy(y).cuda()
    b = eng.make_batch(dx, dy, [(i, i) for i in range(len(dx))], device_nan_flags=True)
    keep = eng.draw_dropout_masks(b) if eng.dropout_encoders else None
    for _ in range(3):
        eng.local_step(b, 1.0, 0.0, accumulate=True)
    stream = torch.cuda.current_stream().cuda_stream
    for wg in (1, 0):
        for _ in range(3):
            assert eng.lib.mmn_chain_fwd(eng._plan, C.byref(b), 1.0, 0.0, wg, stream) == 0
        torch.cuda.synchronize()
        ptr = eng.lib.mmn_debug_buffer(eng._plan, 3, 0)
        off = ptr - eng.workspace.data_ptr()
        st = eng.workspace[off:off + 8 * 250].view(torch.int64).cpu().numpy()[200:216]
        st = st[st > 0]
        print(f"dropout {drop} want_grads {wg}: ", " ".join(f"{v:.2f}" for v in np.diff(st) / 100.0), "total", (st[-1] - st[0]) / 100.0, flush=True)
