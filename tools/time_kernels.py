"""Diagnostic: per-kernel and per-step times of one workload with the library MMN_LIB_PATH names (A/B of kernel variants:
tools/ab_build.sh builds them).  Prints one line: variant, step us (steady state, through MultiModN._train_steps), kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
model = bench.build_model(mm, wl, dev)
model.nan_policy = "device"
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
steps = [res[i % 8] for i in range(256)]
import time
for _ in range(3):
    model._train_steps(steps, opt)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); model._train_steps(steps, opt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / len(steps) * 1e6)
eng = model._engine
lib, plan = eng.lib, eng._plan
stream = torch.cuda.current_stream().cuda_stream
eng.begin_sequence()
pairs = [(i, i) for i in range(len(wl["F"]))]
bs = [eng.make_batch(xs, y, pairs, device_nan_flags=True) for xs, y in res]
alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
d = opt.fused_descriptor(eng)
fused = lib.mmn_chain_kernel_name(plan, C.byref(bs[0]), 2).decode()
kern = {}
if fused:
    kern[fused] = lambda b: lib.mmn_chain_fwd_bwd(plan, C.byref(b), alpha, beta, stream)
else:
    kern["fwd"] = lambda b: lib.mmn_chain_fwd(plan, C.byref(b), alpha, beta, 1, stream)
    kern["bwd"] = lambda b: lib.mmn_chain_bwd(plan, C.byref(b), beta, stream)
kern["k_wgrad"] = lambda b: lib.mmn_wgrad(plan, C.byref(b), stream)
kern["k_reduce"] = lambda b: lib.mmn_reduce_adam(plan, C.byref(b), C.byref(d), stream)
so = mm.hip.StepOpts(); so.adam = C.pointer(d); so.accumulate_epoch = 1
kern["k_wgrad(+side work) + k_reduce(gradient blocks)"] = lambda b: lib.mmn_wgrad_reduce(plan, C.byref(b), alpha, beta, C.byref(so), stream)
so2 = mm.hip.StepOpts(); so2.accumulate_epoch = 1
kern["the same without Adam"] = lambda b: lib.mmn_wgrad_reduce(plan, C.byref(b), alpha, beta, C.byref(so2), stream)
out = {}
for name, fn in kern.items():
    t = []
    for rnd in range(5):
        b = bs[rnd % 8]
        if eng.dropout_encoders:
            keep = eng.draw_dropout_masks(b)
        eng.local_step(b, alpha, beta, accumulate=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            assert fn(b) == 0
        e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e3 / 20)
    out[name] = round(float(np.median(t)), 2)
print(os.environ.get("MMN_LIB_PATH", "default").split("/")[-1], f"step {np.median(ts):.2f} us (min {min(ts):.2f})", out, flush=True)
