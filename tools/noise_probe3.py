"""Diagnostic: after each fused Adam step, where does encoders.3.layers.0.weight leave the fp64 trajectory?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
from oracle import multimodn_oracle as O
wl = bench.WORKLOADS["c3"]; B = wl["B"]; dev = torch.device("cuda")
spec = bench.oracle_spec(O, wl)
pairs = [(i, i) for i in range(len(wl["F"]))]
batches = bench.synthetic_batches(wl, B * 3, B, seed=5)
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
eng = model._get_engine(B)
opt = mm.optim.Adam(model.parameters(), lr=1e-3)
params = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
params64 = {n: v.astype(np.float64) for n, v in params.items()}
oopt, oopt64 = O.Adam(1e-3), O.Adam(1e-3)
eng.begin_sequence()
T = "encoders.3.layers.0.weight"
for s in range(3):
    xs, y = batches[s]
    dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
    b = eng.make_batch(dx, dy, pairs, device_nan_flags=False)
    assert eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt)
    opt.step(); torch.cuda.synchronize()
    g_hip = dict(zip(eng.names, [g.detach().cpu().numpy().copy() for g in eng.grad_views]))
    r32 = O.forward_backward(params, spec, xs, y); oopt.step(params, r32.grads)
    r64 = O.forward_backward(params64, spec, xs, y, dtype=np.float64); oopt64.step(params64, r64.grads)
    w = dict(model.named_parameters())[T].detach().cpu().numpy().astype(np.float64)
    e_h = np.abs(w - params64[T]); e_c = np.abs(params[T] - params64[T])
    idx = np.unravel_index(np.argmax(e_h), e_h.shape)
    print(f"step {s+1}: {T} max err hip {e_h.max():.3e} (numpy {e_c.max():.3e}) at {idx}; there: g64 {r64.grads[T][idx]:.4e} g32 {r32.grads[T][idx]:.4e} "
          f"ghip {g_hip[T][idx]:.4e}  m64 {oopt64.state[T]['m'][idx]:.4e} v64 {oopt64.state[T]['v'][idx]:.4e}")
    print(f"         |g64| quantiles of the tensor: {np.quantile(np.abs(r64.grads[T]), [0, .01, .1, .5, .9, 1])}")
    st = opt.state_dict()["state"]
    big = np.argwhere(e_h > 20 * max(e_c.max(), 1e-12))
    print(f"         elements with hip err > 20x numpy max err: {len(big)} of {e_h.size}; rows {sorted(set(big[:,0].tolist()))[:40]}")
