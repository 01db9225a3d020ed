"""Diagnostic: where the host time of one 20-step MultiModN._train_steps call goes (C3): time to the first graph launch,
each graph launch, the tail."""
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
orig_refresh = eng.lib.mmn_pack_refresh
class G:
    def __init__(self, g): self.g = g
    def replay(self):
        t0 = time.perf_counter(); self.g.replay(); marks.append(("replay", t0, time.perf_counter()))
for ent in eng._step_graphs.values():
    if ent[1] is not None and not isinstance(ent[1], G):
        ent[1] = G(ent[1])
rows = []
for _ in range(60):
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    model._train_steps(steps, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r = [(a - t0) * 1e6 for _, a, b in marks] + [(b - t0) * 1e6 for _, a, b in marks[-1:]]
    rows.append([(marks[0][1] - t0) * 1e6] + [(b - a) * 1e6 for _, a, b in marks] + [(t1 - marks[-1][2]) * 1e6, (t2 - t0) * 1e6])
rows = np.median(np.array(rows), axis=0)
print("to first replay %.1f us | replays %s us | tail %.1f us | wall %.1f us" % (rows[0], " ".join("%.1f" % v for v in rows[1:-2]), rows[-2], rows[-1]))
