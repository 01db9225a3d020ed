"""Diagnostic: hidden activations / pre-activation gradients of one encoder after a training step, row by row against
float64 numpy evaluated on the HIP path's own weights and its own dS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["c3"]; B = wl["B"]; dev = torch.device("cuda")
pairs = [(i, i) for i in range(len(wl["F"]))]
NFUSED = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ENC = 3
batches = bench.synthetic_batches(wl, B * (NFUSED + 1), B, seed=seed)
model = bench.build_model(mm, wl, dev); model.nan_policy = "device"
alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
eng = model._get_engine(B)
opt = mm.optim.Adam(model.parameters(), lr=1e-3)
eng.begin_sequence()
for s in range(NFUSED + 1):
    xs, y = batches[s]
    dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
    b = eng.make_batch(dx, dy, pairs, device_nan_flags=False)
    if s < NFUSED:
        assert eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt); opt.step()
    else:
        eng.local_step(b, alpha, beta, accumulate=True)
    torch.cuda.synchronize()
P = {n: p.detach().cpu().numpy().astype(np.float64) for n, p in model.named_parameters()}
x = xs[ENC].astype(np.float64)
W0, b0 = P[f"encoders.{ENC}.layers.0.weight"], P[f"encoders.{ENC}.layers.0.bias"]
W1, b1 = P[f"encoders.{ENC}.layers.1.weight"], P[f"encoders.{ENC}.layers.1.bias"]
W2 = P[f"encoders.{ENC}.layers.2.weight"]
h0 = np.maximum(x @ W0.T + b0, 0); h1 = np.maximum(h0 @ W1.T + b1, 0)
ML = mm.hip.MAX_LAYERS
hid0 = eng.debug_tensor(6, ENC * ML + 0, eng.max_batch, 32)[:B].cpu().numpy()
hid1 = eng.debug_tensor(6, ENC * ML + 1, eng.max_batch, 32)[:B].cpu().numpy()
dp0 = eng.debug_tensor(7, ENC * ML + 0, eng.max_batch, 32)[:B].cpu().numpy()
dp1 = eng.debug_tensor(7, ENC * ML + 1, eng.max_batch, 32)[:B].cpu().numpy()
dS = eng.debug_tensor(2, ENC, eng.max_batch, eng.S)[:B].cpu().numpy().astype(np.float64)
e_dp1 = (dS @ W2[:, :32]) * (hid1 > 0)
e_dp0 = (e_dp1 @ W1) * (hid0 > 0)
for name, got, exp in (("h0", hid0, h0), ("h1", hid1, h1), ("dpre1", dp1, e_dp1), ("dpre0", dp0, e_dp0)):
    err = np.abs(got - exp).max(axis=1) / max(np.abs(exp).max(), 1e-30)
    bad = np.argwhere(err > 1e-5).flatten()
    print(f"{name}: max rel err {err.max():.3e}; rows off by > 1e-5: {len(bad)} {bad[:20].tolist()}")
    for r in bad[:3]:
        print("   row", r, "got", got[r, :8], "\n        exp", exp[r, :8])
G = dict(zip(eng.names, [g.detach().cpu().numpy().astype(np.float64) for g in eng.grad_views]))
for name, exp in ((f"encoders.{ENC}.layers.0.weight", dp0.astype(np.float64).T @ x), (f"encoders.{ENC}.layers.0.bias", dp0.astype(np.float64).sum(0)),
                  (f"encoders.{ENC}.layers.1.weight", dp1.astype(np.float64).T @ hid0.astype(np.float64)), (f"encoders.{ENC}.layers.1.bias", dp1.astype(np.float64).sum(0))):
    d = np.abs(G[name] - exp) / np.abs(exp).max()
    print(f"{name}: grad vs (HIP's own dpre)^T In: max rel {d.max():.3e}")
    if d.max() > 1e-5:
        # which 512-row range of the batch explains the difference?
        diff = G[name] - exp
        A = dp0 if "layers.0" in name else dp1
        I = x if "layers.0.weight" in name else (hid0 if "layers.1.weight" in name else None)
        for k in range(0, B, 512):
            part = (A[k:k+512].astype(np.float64).T @ I[k:k+512]) if I is not None else A[k:k+512].astype(np.float64).sum(0)
            c = (diff * part).sum() / max((part * part).sum(), 1e-300)
            print(f"    rows {k}-{k+511}: projection of the difference on this range's partial = {c:+.4f}")
from oracle import multimodn_oracle as O
spec = bench.oracle_spec(O, wl)
r64 = O.forward_backward(P, spec, xs, y, dtype=np.float64)
for n in eng.names:
    d = np.abs(G[n] - r64.grads[n]).max() / np.abs(r64.grads[n]).max()
    if d > 1e-5:
        print(f"vs oracle64: {n} max rel {d:.3e}")
# the oracle's own intermediates, recomputed here
st = O.forward_backward(P, spec, xs, y, dtype=np.float64, keep_states=True).states
s3, s4 = st[ENC], st[ENC + 1]
hs4 = eng.state_rows(ENC, B).cpu().numpy()
print("state after encoder 3: hip vs oracle64", np.abs(hs4 - s4).max() / np.abs(s4).max())
print("oracle grad l0 vs my dpre0^T x:", np.abs(r64.grads[f"encoders.{ENC}.layers.0.weight"] - e_dp0.T @ x).max() / np.abs(e_dp0.T @ x).max())
