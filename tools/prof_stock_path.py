"""Diagnostic: cProfile of MultiModN.train_epoch over a stock torch DataLoader(PartitionDataset) with torch.optim.Adam
(bench.py's stock_path: the reference pipeline with the import swapped), C3 shape."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils.data import DataLoader
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
X = np.concatenate([np.concatenate(h[0], axis=1) for h in host], 0); y = np.concatenate([h[1] for h in host], 0)
loader = DataLoader(mm.PartitionDataset(X, y, list(wl["F"])), batch_size=B, shuffle=False)
model = bench.build_model(mm, wl, torch.device("cuda"))
opt = torch.optim.Adam(model.parameters(), wl["lr"])
hist = mm.MultiModNHistory([f"t{d}" for d in range(wl["D"])])
crit = torch.nn.CrossEntropyLoss()
for _ in range(3):
    model.train_epoch(loader, opt, crit, hist)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model.train_epoch(loader, opt, crit, hist)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
