"""Diagnostic: split the distance of a first-step gradient to fp64 into (i) summation noise of k_wgrad + k_reduce
(HIP grad vs the float64 product of the HIP path's own operands) and (ii) operand noise (that float64 product vs the
float64 oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
from oracle import multimodn_oracle as O
wl = bench.WORKLOADS["c3"]; B = wl["B"]; dev = torch.device("cuda")
pairs = [(i, i) for i in range(len(wl["F"]))]
spec = bench.oracle_spec(O, wl)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
xs, y = bench.synthetic_batches(wl, B, B, seed=seed)[0]
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
eng = model._get_engine(B)
eng.begin_sequence()
dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
b = eng.make_batch(dx, dy, pairs, device_nan_flags=False)
eng.local_step(b, alpha, beta, accumulate=True); torch.cuda.synchronize()
P32 = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
P = {n: v.astype(np.float64) for n, v in P32.items()}
G = dict(zip(eng.names, [g.detach().cpu().numpy().astype(np.float64) for g in eng.grad_views]))
r64 = O.forward_backward(P, spec, xs, y, dtype=np.float64)
r32 = O.forward_backward(P32, spec, xs, y)
ML = mm.hip.MAX_LAYERS
f64 = lambda t: t.cpu().numpy().astype(np.float64)
def rel(a, t): 
    d = np.abs(a - t) / np.abs(t).max(); return d.max(), np.sqrt((d ** 2).mean())
print("tensor: total (hip vs oracle64) | summation (hip vs f64 of own operands) | operands (f64 of own operands vs oracle64) | numpy32 total   [max rms]")
for e in range(4):
    h1 = f64(eng.debug_tensor(6, e * ML + 1, eng.max_batch, 32)[:B]); h0 = f64(eng.debug_tensor(6, e * ML + 0, eng.max_batch, 32)[:B])
    dS = f64(eng.debug_tensor(2, e, eng.max_batch, eng.S)[:B])
    dp1 = f64(eng.debug_tensor(7, e * ML + 1, eng.max_batch, 32)[:B]); dp0 = f64(eng.debug_tensor(7, e * ML + 0, eng.max_batch, 32)[:B])
    sprev = f64(eng.state_rows(e - 1, B)) if e > 0 else np.tile(P["init_state.state_value"], (B, 1))
    for n, own in ((f"encoders.{e}.layers.2.weight", dS.T @ np.concatenate([h1, sprev], 1)), (f"encoders.{e}.layers.1.weight", dp1.T @ h0),
                   (f"encoders.{e}.layers.0.weight", dp0.T @ xs[e].astype(np.float64))):
        t = r64.grads[n]
        print(f"  {n:30s} {rel(G[n], t)[0]:.2e} {rel(G[n], t)[1]:.2e} | {rel(G[n], own)[0]:.2e} {rel(G[n], own)[1]:.2e} | "
              f"{rel(own, t)[0]:.2e} {rel(own, t)[1]:.2e} | {rel(r32.grads[n].astype(np.float64), t)[0]:.2e} {rel(r32.grads[n].astype(np.float64), t)[1]:.2e}")
