"""us/step of the reference's Titanic pipeline shapes (pipelines/titanic/titanic_{mlp,partitioned,featurewise,missingness}_pipeline.py:
batch 32, Adam 0.01, penalties 0.7 / 0.3) on device-resident batches: the one-launch epoch kernel against the step-by-step path.
Usage: python tools/time_titanic_shapes.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import multimodn_amd as mm
import bench
shapes = {"mlp (C1)": dict(S=32, F=[6], H=(5, 5), D=1, B=32, lr=1e-2, pen=(0.7, 0.3)),
          "mlp, state 1": dict(S=1, F=[6], H=(5, 5), D=1, B=32, lr=1e-2, pen=(0.7, 0.3)),
          "partitioned [3,2]": dict(S=5, F=[3, 2], H=(5, 5), D=1, B=32, lr=1e-2, pen=(0.7, 0.3)),
          "featurewise x5": dict(S=5, F=[1] * 5, H=(5,), D=1, B=32, lr=1e-2, pen=(0.7, 0.3)),
          "featurewise x6": dict(S=5, F=[1] * 6, H=(5,), D=1, B=32, lr=1e-2, pen=(0.7, 0.3))}
n = 200
for name, wl in shapes.items():
    wl = dict(wl, text=name)
    B = wl["B"]
    host = bench.synthetic_batches(wl, B * 8, B, seed=1)
    res = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host]
    steps = [res[i % 8] for i in range(n)]
    out = []
    for use in (True, False):
        model = bench.build_model(mm, wl, torch.device("cuda"))
        model.nan_policy = "device"
        model.epoch_kernel = use
        opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
        for _ in range(4):
            model._train_steps(steps, opt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); model._train_steps(steps, opt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
        out.append((bool(model.__dict__.get("_small_epochs")), float(np.median(ts))))
    print(f"{name:20s} epoch kernel (taken {out[0][0]}): {out[0][1]:6.2f} us/step   step path: {out[1][1]:6.2f} us/step", flush=True)
