"""Diagnostic: samples/s of the PUBLIC path, MultiModN.train_epoch over a list of batches (C3 shape),
for host-resident and device-resident batches and both NaN policies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench

wl = bench.WORKLOADS["c3"]
B, NB = 4096, 64
host = bench.synthetic_batches(wl, B * NB, B, seed=1)
crit = torch.nn.CrossEntropyLoss()
for where in ("cpu", "cuda"):
    loader = [([torch.from_numpy(x).to(where) for x in xs], torch.from_numpy(y).to(where)) for xs, y in host]
    for policy in ("host", "device"):
        for optname in ("torch", "hip"):
            model = bench.build_model(mm, wl, torch.device("cuda"))
            model.nan_policy = policy
            opt = (torch.optim.Adam if optname == "torch" else mm.optim.Adam)(list(model.parameters()), 1e-3)
            hist = mm.MultiModNHistory(["a", "b", "c"])
            model.train_epoch(loader[:4], opt, crit, hist)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.train_epoch(loader, opt, crit, hist)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(f"batches on {where:4s} nan_policy={policy:6s} optimizer={optname:5s}: {B * NB / el / 1e6:7.2f} M samples/s  {el / NB * 1e6:7.1f} us/step")
