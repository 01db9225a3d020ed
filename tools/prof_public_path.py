import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
B, NB = wl["B"], 256
host = bench.synthetic_batches(wl, B * 16, B, seed=1)
crit = torch.nn.CrossEntropyLoss()
loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host] * (NB // 16)
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = sys.argv[2] if len(sys.argv) > 2 else "device"
model.replay_steps = not (len(sys.argv) > 4 and sys.argv[4] == "noreplay")
opt = (torch.optim.Adam if (len(sys.argv) > 3 and sys.argv[3] == "torch") else mm.optim.Adam)(list(model.parameters()), 1e-3)
hist = mm.MultiModNHistory([f"t{d}" for d in range(wl["D"])])
model.train_epoch(loader[:8], opt, crit, hist)
model.train_epoch(loader, opt, crit, hist)               # (second sighting of every batch: the steps are captured here)
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
