import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]
B, NB = 4096, 256
host = bench.synthetic_batches(wl, B * 16, B, seed=1)
crit = torch.nn.CrossEntropyLoss()
loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host] * (NB // 16)
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
opt = mm.optim.Adam(list(model.parameters()), 1e-3)
hist = mm.MultiModNHistory(["a", "b", "c"])
model.train_epoch(loader[:8], opt, crit, hist)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.train_epoch(loader, opt, crit, hist)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"public path: {B*NB/el/1e6:.2f} M samples/s {el/NB*1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
model.train_epoch(loader, opt, crit, hist)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(14)
