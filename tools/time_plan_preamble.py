"""Diagnostic: host time of every piece of MultiModN._replay_epoch_plan in front of its first graph launch (C3, 20 steps)."""
import os, sys, time, operator
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
for _ in range(5):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
T = {}
def lap(name, t):
    now = time.perf_counter(); T.setdefault(name, []).append((now - t) * 1e6); return now
for _ in range(200):
    torch.cuda.synchronize()
    t = time.perf_counter()
    plans = model.__dict__.get("_epoch_plans"); seq = steps
    ep = plans.get((len(seq), id(seq[0]), id(seq[-1])))
    ok = not (ep is None or ep["opt"] is not opt or ep["eng"] is not model._engine or len(ep["batches"]) != len(seq) or not all(map(operator.is_, seq, ep["batches"])))
    t = lap("lookup + batch identity", t)
    flat = ep["flat"]
    for batch, (y, xs) in zip(seq, flat):
        data = batch[0]
        if batch[1] is not y or len(data) != len(xs) or not all(map(operator.is_, data, xs)) or (len(batch) > 2 and batch[2] is not None):
            raise SystemExit("mismatch")
    t = lap("tensor identity of 20 batches", t)
    eng = model._get_engine(ep["rows"]); t = lap("_get_engine", t)
    eng.begin_sequence(sig_checked=True); t = lap("begin_sequence", t)
    eng.refresh_weights(); t = lap("refresh_weights (launch)", t)
    fd = opt.fused_descriptor(eng); t = lap("fused_descriptor", t)
    eng.adam_fusable(opt, fd); t = lap("adam_fusable", t)
    g = eng.params[0].grad is not eng.grad_views[0] or eng.params[-1].grad is not eng.grad_views[-1]; t = lap("grad views check", t)
    hp = eng.group_hp_key(1.0, 0.003, opt, fd, 0); t = lap("group_hp_key", t)
    torch.cuda.synchronize()
print({k: round(float(np.median(v)), 2) for k, v in T.items()}, "sum", round(sum(float(np.median(v)) for v in T.values()), 1))
t0 = time.perf_counter()
for _ in range(200):
    steps2 = list(steps)
print("list(steps)", (time.perf_counter() - t0) / 200 * 1e6)
