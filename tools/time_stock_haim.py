"""The reference's MIMIC pipeline body at its real shape (pipelines/mimic/mimic_multi_task_pipeline.py:53-154: sources of 6 /
1024 / 768 / 99 features, state 50, batch 16) with the import swapped: torch DataLoader(PartitionDataset) + torch.optim.Adam
through MultiModN.train_epoch; beside it the resident form (DeviceResidentLoader + multimodn_amd.optim.Adam).
Usage: python tools/time_stock_haim.py [--rows 2048] [--epochs 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils.data import DataLoader
import multimodn_amd as mm
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2048)
ap.add_argument("--epochs", type=int, default=4)
a = ap.parse_args()
wl = bench.WORKLOADS["haim"]
B = wl["B"]
rng = np.random.default_rng(0)
X = rng.standard_normal((a.rows, sum(wl["F"]))).astype(np.float32)
y = rng.integers(0, 2, (a.rows, wl["D"])).astype(np.int64)
ds = mm.PartitionDataset(X, y, list(wl["F"]))
crit = torch.nn.CrossEntropyLoss()
for name, loader, Adam in (("stock DataLoader + torch.optim.Adam", DataLoader(ds, B), torch.optim.Adam),
                           ("stock DataLoader + multimodn_amd.optim.Adam", DataLoader(ds, B), mm.optim.Adam),
                           ("DeviceResidentLoader + multimodn_amd.optim.Adam", mm.DeviceResidentLoader(ds, B), mm.optim.Adam)):
    model = bench.build_model(mm, wl, torch.device("cuda"))
    opt = Adam(list(model.parameters()), wl["lr"])
    hist = mm.MultiModNHistory(["a", "b"])
    ts = []
    for ep in range(a.epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.train_epoch(loader, opt, crit, hist)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / len(loader) * 1e6)
    print(f"{name:50s} {np.median(ts[1:]):8.1f} us/step  ({[round(t, 1) for t in ts]})", flush=True)
