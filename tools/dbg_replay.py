import os, sys
sys.path.insert(0, "/root/repo")
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
orig = eng.replay_known
def dbg(ent, steps_, nxt, hp, optimizer, draw, reset_first=False):
    first = steps_[0][4]
    exp = (hp, bool(first.nan_flags and eng._prescanned is first), bool(draw and eng._predrawn is first), bool(reset_first))
    print("ent None", ent is None, "graph None", None if ent is None else ent[1] is None, "ent3 None", None if ent is None else ent[3] is None)
    if ent is not None and ent[3] is not None:
        print(" eq parts", [a == b for a, b in zip(ent[3], exp)], "hp eq", [a == b for a, b in zip(ent[3][0], exp[0])])
    return orig(ent, steps_, nxt, hp, optimizer, draw, reset_first)
eng.replay_known = dbg
model._train_steps(steps, opt)
torch.cuda.synchronize()
