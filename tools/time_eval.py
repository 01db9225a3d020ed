"""Diagnostic: forward-only throughput (MultiModN.test over device-resident batches, C3 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
B, NB = 4096, 128
host = bench.synthetic_batches(wl, B * 16, B, seed=1)
loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host] * (NB // 16)
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
crit = torch.nn.CrossEntropyLoss()
model.test(loader[:8], crit)
model.test(loader, crit)                                  # (the first full pass also grows torch's allocator pools)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.test(loader, crit)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"test(): {B * NB / el / 1e6:.2f} M samples/s, {el / NB * 1e6:.1f} us/step (incl. the per-epoch report)")
eng = model._get_engine(B)
xs, y = loader[0]
b = eng.make_batch(xs, y, [(k, k) for k in range(len(xs))], device_nan_flags=True)
for _ in range(20):
    eng.eval_step(b, accumulate=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    eng.eval_step(b, accumulate=True)
e1.record(); torch.cuda.synchronize()
print(f"mmn_eval_step: {e0.elapsed_time(e1) * 1e3 / 200:.1f} us/step = {B * 200 / (e0.elapsed_time(e1) * 1e-3) / 1e6:.1f} M samples/s")
