"""Diagnostic: line-level host cost of MultiModN._replay_epoch_plan (20 steps, C3)."""
import os, sys, time
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
steps = [res[i % 8] for i in range(20)]
for _ in range(4):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
eng = model._engine
T = {}
def timed(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, w)
for n in ("epoch_reset", "begin_sequence", "adam_fusable", "group_hp_key", "replay_known", "note_parameters_current", "assign_grads"):
    timed(eng, n)
timed(opt, "fused_descriptor"); timed(opt, "fused_step_seen"); timed(model, "_get_engine"); timed(model, "_replay_epoch_plan")
for _ in range(50):
    torch.cuda.synchronize()
    model._train_steps(steps, opt)
torch.cuda.synchronize()
for k, v in T.items():
    print(f"{k:26s} calls/iter {len(v)/50:4.1f}  median {np.median(v)*1e6:6.2f} us  total/iter {np.sum(v)/50*1e6:6.2f} us")
