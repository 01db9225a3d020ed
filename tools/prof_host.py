"""Diagnostic: where the HOST time of MultiModN._train_steps goes (cProfile over an epoch of device-resident batches)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev)
model.nan_policy = "device"
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
steps = [res[i % 8] for i in range(n)]
for _ in range(3):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    model._train_steps(steps, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host returned after {(t1 - t0) * 1e6:.0f} us, GPU done after {(t2 - t0) * 1e6:.0f} us ({(t2 - t0) / n * 1e6:.1f} us/step)")
pr = cProfile.Profile()
pr.enable()
model._train_steps(steps, opt)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
