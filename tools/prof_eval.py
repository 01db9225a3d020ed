"""Diagnostic: host-side profile of MultiModN.test (which Python frames cost time per batch)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
B, NB = 4096, 64
host = bench.synthetic_batches(wl, B * 16, B, seed=1)
loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host] * (NB // 16)
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
crit = torch.nn.CrossEntropyLoss()
model.test(loader[:8], crit)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
model.test(loader, crit)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
