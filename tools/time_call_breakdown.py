"""Diagnostic: where the fixed cost of one MultiModN._train_steps call goes (host side): time from the call to the first group's
launch, the launch itself, the remaining groups, the final synchronize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
steps = [res[i % 8] for i in range(20)]
for _ in range(4):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
eng = model._engine
marks = []
orig = eng.replay_known
def wrapped(*a, **k):
    marks.append(time.perf_counter())
    r = orig(*a, **k)
    marks.append(time.perf_counter())
    return r
eng.replay_known = wrapped
rows = []
for _ in range(40):
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter(); model._train_steps(steps, opt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows.append([(marks[0] - t0), (marks[1] - marks[0]), (t1 - marks[1]), (t2 - t1), (t2 - t0)])
r = np.median(np.array(rows) * 1e6, axis=0)
print(f"20 steps: to first launch {r[0]:.1f} us, first launch {r[1]:.1f}, rest of the call {r[2]:.1f}, synchronize {r[3]:.1f}, total {r[4]:.1f} ({r[4]/20:.2f} / step)")
import cProfile, pstats
pr = cProfile.Profile()
eng.replay_known = orig
pr.enable()
for _ in range(200):
    model._train_steps(steps, opt)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
