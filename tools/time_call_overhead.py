"""Diagnostic: fixed cost of one MultiModN._train_steps call (intercept of wall time over the number of steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
if os.environ.get("MMN_RG"):                             # first group / later groups of a call (MultiModN.REPLAY_GROUP, _NEXT)
    type(model).REPLAY_GROUP = int(os.environ["MMN_RG"]); type(model).REPLAY_GROUP_NEXT = int(os.environ.get("MMN_RGN", "0"))
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
out = {}
for n in (8, 16, 20, 24, 40, 80, 200):
    steps = [res[i % 8] for i in range(n)]
    for _ in range(4):
        model._train_steps(steps, opt)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); model._train_steps(steps, opt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6))
    out[n] = (float(np.median([t[0] for t in ts])), float(np.median([t[1] for t in ts])))
    print(f"{n:3d} steps: wall {out[n][0]:8.1f} us ({out[n][0] / n:6.2f} / step), host returns after {out[n][1]:7.1f} us")
ns = sorted(out); a = np.polyfit(ns, [out[n][0] for n in ns], 1)
print(f"fit: {a[0]:.2f} us / step + {a[1]:.1f} us per call")
