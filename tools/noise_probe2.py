"""Diagnostic: gradients of step 2 (computed on the copies the fused Adam tail of step 1 scattered) against the fp64 oracle
evaluated at the SAME weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
from oracle import multimodn_oracle as O
wl = bench.WORKLOADS["c3"]; B = wl["B"]; dev = torch.device("cuda")
spec = bench.oracle_spec(O, wl)
pairs = [(i, i) for i in range(len(wl["F"]))]
def dist(a, t):
    s = max(np.abs(t).max(), 1e-30)
    d = np.abs(np.asarray(a, np.float64).reshape(t.shape) - t) / s
    return d.max(), np.sqrt((d ** 2).mean())
NFUSED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for seed in (5, 77):
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
            assert eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt)
            opt.step()
        else:
            eng.local_step(b, alpha, beta, accumulate=True)
        torch.cuda.synchronize()
    pw = {n: p.detach().cpu().numpy().astype(np.float64) for n, p in model.named_parameters()}
    pw32 = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    r64 = O.forward_backward(pw, spec, xs, y, dtype=np.float64)
    r32 = O.forward_backward(pw32, spec, xs, y)
    print(f"seed {seed}: step-2 gradients at the weights the fused step left (hip max rms | numpy fp32 max rms | ratio)")
    for n, gv in zip(eng.names, eng.grad_views):
        hm, hr = dist(gv.detach().cpu().numpy(), r64.grads[n]); cm, cr = dist(r32.grads[n], r64.grads[n])
        flag = "  <<<<" if hr > 3 * cr else ""
        print(f"  {n:32s} {hm:.2e} {hr:.2e} | {cm:.2e} {cr:.2e} | {hm / max(cm, 1e-30):7.2f} {hr / max(cr, 1e-30):7.2f}{flag}")
