"""Diagnostic: the data-parallel step's Adam tail alone - k_adam_accumulate (Adam + scatter + gates + epoch block) against the
plain k_adam (no scatter, no gates, no epoch block) - back to back, HIP events (C3)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
model._train_steps([res[i % 8] for i in range(16)], opt)
torch.cuda.synchronize()
eng = model._engine
d = opt.fused_descriptor(eng)
st = torch.cuda.current_stream().cuda_stream
def rep(fn, n=50):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): assert fn() == 0
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return round(float(np.median(ts)), 2)
print("k_adam_accumulate", rep(lambda: eng.lib.mmn_adam_step_accumulate(eng._plan, C.byref(d), 1.0, 0.003, st)),
      "us | k_adam", rep(lambda: eng.lib.mmn_adam_step(C.byref(d), st)), "us | k_epoch_accumulate",
      rep(lambda: eng.lib.mmn_epoch_accumulate(eng._plan, 1.0, 0.003, st)), "us | empty-ish k_prepare", rep(lambda: eng.lib.mmn_pack_refresh(eng._plan, st)))
