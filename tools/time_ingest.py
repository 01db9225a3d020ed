"""Diagnostic: where the host-batch path of train_epoch spends its time (C3 shape, host tensors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if nt:
    torch.set_num_threads(nt)
print("torch threads", torch.get_num_threads())
wl = bench.WORKLOADS["c3"]
B, NB = 4096, 32
host = bench.synthetic_batches(wl, B * NB, B, seed=1)
loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in host]
crit = torch.nn.CrossEntropyLoss()
for policy in ("host", "device"):
    for staged in (True, False):
        model = bench.build_model(mm, wl, torch.device("cuda"))
        model.nan_policy = policy
        if not staged:
            class NoStage:
                def stage(self, data, y):
                    return ([t.to("cuda", dtype=torch.float32, non_blocking=True).contiguous() for t in data],
                            y.to("cuda", non_blocking=True).contiguous())
            model._stager = NoStage()
        opt = mm.optim.Adam(list(model.parameters()), 1e-3)
        model.train_epoch(loader[:4], opt, crit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_epoch(loader, opt, crit)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"policy={policy:6s} staged={staged}: {el / NB * 1e6:8.1f} us/step")
# raw pieces
xs, y = loader[0]
for name, fn in (("isnan.any x4", lambda: [bool(torch.isnan(t).any()) for t in xs]),
                 ("pageable .to x4", lambda: [t.to("cuda", non_blocking=True) for t in xs])):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us")
