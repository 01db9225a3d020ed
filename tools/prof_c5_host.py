"""Diagnostic: host profile of the per-sample training loop (bench workload c5)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c5"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
model.per_sample = True
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
steps = [res[i % 8] for i in range(80)]
run = model._train_steps_per_sample
for _ in range(3):
    run(steps, opt)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps, opt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"80 steps: host returns after {(t1-t0)/80*1e6:.1f} us/step, wall {(t2-t0)/80*1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    run(steps, opt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
