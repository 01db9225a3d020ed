"""us/step of the reference's real MIMIC shape (bench.py's `haim` workload) in PER-SAMPLE mode: 30 % of the sources missing per
sample (NaN rows), default encoder order, device-resident batches through MultiModN._train_steps.
Usage: python tools/time_haim_per_sample.py [batch=16] [steps=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.WORKLOADS["haim"]
model = bench.build_model(mm, wl, torch.device("cuda"))
model.per_sample = True
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
rng = np.random.default_rng(0)
res = []
for xs, y in host:
    xs = [x.copy() for x in xs]
    for x in xs:
        x[rng.random(len(x)) < 0.3] = np.nan
    res.append(([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()))
steps = [res[i % 8] for i in range(n)]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
for _ in range(4):
    model._train_steps_per_sample(steps, opt)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); model._train_steps_per_sample(steps, opt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
print(f"haim, per-sample mode, batch {B}: {np.median(ts):.1f} us/step (min {min(ts):.1f})")
