"""Diagnostic: the reference pipelines' own batch size (16 rows, pipelines/mimic/mimic_multi_task_pipeline.py:77) through the
public path: MultiModN.train_epoch + test() over a DeviceResidentLoader-style list of batches, epoch after epoch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
crit = torch.nn.CrossEntropyLoss()
for name in ("c3", "mimic"):
    wl = dict(bench.WORKLOADS[name]); wl["B"] = 16
    B, NB = 16, 512
    host = bench.synthetic_batches(wl, B * NB, B, seed=1)
    loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host]
    model = bench.build_model(mm, wl, torch.device("cuda")); model.nan_policy = "device"
    opt = mm.optim.Adam(list(model.parameters()), 1e-3)
    hist = mm.MultiModNHistory(["a", "b", "c"])
    ts, te = [], []
    for ep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.train_epoch(loader, opt, crit, hist)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.test(loader, crit, hist)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0) / NB * 1e6); te.append((t2 - t1) / NB * 1e6)
    print(f"{name}: batch 16, {NB} batches per epoch: train_epoch us/step by epoch {[round(t, 1) for t in ts]}, test() us/step {[round(t, 1) for t in te]}"
          f" -> {B / ts[-1]:.2f} M samples/s training")

# the reference pipelines' loaders: host batches, a fresh order every epoch (torch DataLoader(shuffle=True)): nothing recurs,
# every step is launched eagerly from freshly staged tensors
for name in ("c3", "mimic"):
    wl = dict(bench.WORKLOADS[name]); wl["B"] = 16
    B, NB = 16, 512
    host = bench.synthetic_batches(wl, B * NB, B, seed=1)
    for optname in ("hip", "torch"):
        model = bench.build_model(mm, wl, torch.device("cuda"))
        opt = (mm.optim.Adam if optname == "hip" else torch.optim.Adam)(list(model.parameters()), 1e-3)
        hist = mm.MultiModNHistory(["a", "b", "c"])
        ts = []
        for ep in range(3):
            order = np.random.default_rng(ep).permutation(NB)
            loader = [([torch.from_numpy(host[i][0][k].copy()) for k in range(len(host[i][0]))], torch.from_numpy(host[i][1].copy())) for i in order]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.train_epoch(loader, opt, crit, hist)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ts.append((t1 - t0) / NB * 1e6)
        print(f"{name}: HOST batches of 16, fresh every epoch, optimizer {optname}: train_epoch us/step by epoch {[round(t, 1) for t in ts]}")
