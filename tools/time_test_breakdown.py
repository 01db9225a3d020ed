import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]
B, NB = 4096, 128
host = bench.synthetic_batches(wl, B * 16, B, seed=1)
loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host] * (NB // 16)
model = bench.build_model(mm, wl, torch.device("cuda")); model.nan_policy = "device"
crit = torch.nn.CrossEntropyLoss()
model.test(loader, crit); model.test(loader, crit); torch.cuda.synchronize()
t0 = time.perf_counter(); r = model._test_steps_collected(loader); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("fast loop:", "taken" if r is not None else "NOT taken", f"host {(t1-t0)/NB*1e6:.1f} us/step, wall {(t2-t0)/NB*1e6:.1f} us/step")
t0 = time.perf_counter(); model.test(loader, crit); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"test() total {(t3-t0)*1e3:.2f} ms = {(t3-t0)/NB*1e6:.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); model.test(loader, crit); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
