"""Diagnostic: the per-sample regrouping launches alone (mmn_regroup_ex: k_ps_code, k_ps_hist, k_ps_layout, k_ps_gather)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c5"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"; model.per_sample = True
for B in (4096, 16384, 65536):
    host = bench.synthetic_batches(wl, B, B, seed=1)
    xs, y = host[0]
    rng = np.random.default_rng(1)
    xs = [x.copy() for x in xs]
    for e in range(4):
        xs[e][rng.random(B) < 0.3] = np.nan
    dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
    sq = torch.from_numpy(np.stack([rng.permutation(4) for _ in range(B)]).astype(np.int64)).to(dev)
    eng = model._get_engine(B)
    for _ in range(3):
        eng.per_sample_batch(dx, dy, sq)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.per_sample_batch(dx, dy, sq); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"B = {B}: regrouping (4 launches + torch.empty of the outputs) median {np.median(ts):.1f} us, min {min(ts):.1f} us")
