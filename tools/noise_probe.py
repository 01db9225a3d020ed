"""Diagnostic: where does the HIP step sit relative to the fp64 truth, tensor by tensor?  One training step of the c3
workload (batch 4096) without the optimizer: gradients of the HIP path and of the numpy fp32 oracle against the numpy
fp64 oracle (max and RMS distance per tensor, relative to the tensor's max |g|), then N Adam steps: trained weights the
same way.  `python tools/noise_probe.py [steps] [workload]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
from oracle import multimodn_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wname = sys.argv[2] if len(sys.argv) > 2 else "c3"
wl = bench.WORKLOADS[wname]
B = wl["B"]
dev = torch.device("cuda")
spec = bench.oracle_spec(O, wl)
model = bench.build_model(mm, wl, dev)
model.nan_policy = "device"
params = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
params64 = {n: v.astype(np.float64) for n, v in params.items()}
batches = bench.synthetic_batches(wl, B * steps, B, seed=5)
eng = model._get_engine(B)
pairs = [(i, i) for i in range(len(wl["F"]))]
alpha, beta = float(model.err_penalty), float(model.state_change_penalty)


def dist(a, t):
    s = max(np.abs(t).max(), 1e-30)
    d = np.abs(np.asarray(a, np.float64).reshape(t.shape) - t) / s
    return d.max(), np.sqrt((d ** 2).mean())


xs, y = batches[0]
dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
b = eng.make_batch(dx, dy, pairs, device_nan_flags=True)
eng.begin_sequence()
masks = None
if eng.dropout_encoders:
    keep = eng.draw_dropout_masks(b)
    masks = {e: mk.cpu().numpy() for (e, _, _), mk in zip(eng.dropout_encoders, keep)}
eng.local_step(b, alpha, beta, accumulate=True)
torch.cuda.synchronize()
g_hip = {n: gv.detach().cpu().numpy().copy() for n, gv in zip(eng.names, eng.grad_views)}
r32 = O.forward_backward(params, spec, xs, y, drop_masks=masks)
r64 = O.forward_backward(params64, spec, xs, y, drop_masks=masks, dtype=np.float64)
print(f"first-step gradients, batch {B}: distance to fp64 / max|g|   (hip max, rms | numpy-fp32 max, rms | ratio max, rms)")
tot_h = tot_c = 0.0
for n in eng.names:
    if r64.grads[n] is None:
        continue
    hm, hr = dist(g_hip[n], r64.grads[n]); cm, cr = dist(r32.grads[n], r64.grads[n])
    tot_h += hr ** 2; tot_c += cr ** 2
    print(f"  {n:32s} {hm:.2e} {hr:.2e} | {cm:.2e} {cr:.2e} | {hm / max(cm, 1e-30):5.2f} {hr / max(cr, 1e-30):5.2f}")
print(f"  all tensors, rms of rms: hip {np.sqrt(tot_h):.2e} numpy {np.sqrt(tot_c):.2e} ratio {np.sqrt(tot_h / tot_c):.2f}")
# intermediate tensors of the same step: states (forward), dS (backward)
st64 = O.forward_backward(params64, spec, xs, y, drop_masks=masks, dtype=np.float64, keep_states=True).states
st32 = O.forward_backward(params, spec, xs, y, drop_masks=masks, keep_states=True).states
for e in range(len(wl["F"])):
    hs = eng.state_rows(e, B).cpu().numpy()
    hm, hr = dist(hs, st64[e + 1]); cm, cr = dist(st32[e + 1], st64[e + 1])
    print(f"  state after encoder {e}: hip {hm:.2e} {hr:.2e} | numpy {cm:.2e} {cr:.2e}")

# trained weights, under variants of how the steps are driven
def trained(flags, warm):
    model = bench.build_model(mm, wl, dev)
    model.nan_policy = "device"
    params = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    params64 = {n: v.astype(np.float64) for n, v in params.items()}
    eng = model._get_engine(B)
    opt = mm.optim.Adam(model.parameters(), lr=1e-3)
    oopt, oopt64 = O.Adam(1e-3), O.Adam(1e-3)
    eng.begin_sequence()
    if warm:
        xs, y = batches[0]
        dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
        b = eng.make_batch(dx, dy, pairs, device_nan_flags=flags)
        eng.local_step(b, alpha, beta, accumulate=True)
        eng.begin_sequence()
    for s in range(steps):
        xs, y = batches[s]
        dx = [torch.from_numpy(x).to(dev) for x in xs]; dy = torch.from_numpy(y).to(dev)
        b = eng.make_batch(dx, dy, pairs, device_nan_flags=flags)
        masks = None
        if eng.dropout_encoders:
            keep = eng.draw_dropout_masks(b)
            masks = {e: mk.cpu().numpy() for (e, _, _), mk in zip(eng.dropout_encoders, keep)}
        stepped = eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt)
        opt.step()
        torch.cuda.synchronize()
        r = O.forward_backward(params, spec, xs, y, drop_masks=masks); oopt.step(params, r.grads)
        r = O.forward_backward(params64, spec, xs, y, drop_masks=masks, dtype=np.float64); oopt64.step(params64, r.grads)
    print(f"weights after {steps} Adam steps (flags={flags}, warm={warm}, fused={stepped}): distance to the fp64 trajectory / max|w|")
    worst = 0.0
    for n, p in model.named_parameters():
        hm, hr = dist(p.detach().cpu().numpy(), params64[n]); cm, cr = dist(params[n], params64[n])
        worst = max(worst, hm / max(cm, 1e-30))
        if hm / max(cm, 1e-30) > 3:
            print(f"  {n:32s} {hm:.2e} {hr:.2e} | {cm:.2e} {cr:.2e} | {hm / max(cm, 1e-30):5.2f} {hr / max(cr, 1e-30):5.2f}")
    print("  worst max-ratio", worst)

for flags in (False, True):
    for warm in (False, True):
        trained(flags, warm)
