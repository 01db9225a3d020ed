"""Diagnostic: cProfile of the host side of a replayed 20-step MultiModN._train_steps call (where the ~28 us in front of the
first graph launch go)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
for _ in range(5):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 300
pr.enable()
for _ in range(N):
    model._train_steps(steps, opt)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime")
print(f"per call (us), {N} calls, sorted by own time:")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
for (f, l, name), (cc, nc, tt, ct, _) in rows:
    print(f"  {tt / N * 1e6:7.2f} own {ct / N * 1e6:7.2f} cum  x{nc / N:5.1f}  {os.path.basename(f)}:{l} {name}")
