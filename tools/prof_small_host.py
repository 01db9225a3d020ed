"""Diagnostic: host profile of train_epoch over HOST batches of 16 rows (the reference pipelines' loader), eager steps."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = dict(bench.WORKLOADS["c3"]); wl["B"] = 16
B, NB = 16, 512
host = bench.synthetic_batches(wl, B * NB, B, seed=1)
crit = torch.nn.CrossEntropyLoss()
model = bench.build_model(mm, wl, torch.device("cuda"))
opt = mm.optim.Adam(list(model.parameters()), 1e-3)
hist = mm.MultiModNHistory(["a", "b", "c"])
def mk(ep):
    order = np.random.default_rng(ep).permutation(NB)
    return [([torch.from_numpy(host[i][0][k].copy()) for k in range(4)], torch.from_numpy(host[i][1].copy())) for i in order]
model.train_epoch(mk(0), opt, crit, hist); model.train_epoch(mk(1), opt, crit, hist)
torch.cuda.synchronize()
ld = mk(2)
t0 = time.perf_counter(); model.train_epoch(ld, opt, crit, hist); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host returns after {(t1-t0)/NB*1e6:.1f} us/step, wall {(t2-t0)/NB*1e6:.1f} us/step")
ld = mk(3)
pr = cProfile.Profile(); pr.enable(); model.train_epoch(ld, opt, crit, hist); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(24)
