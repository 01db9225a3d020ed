"""Diagnostic: per-phase timestamps of one chain-kernel workgroup (MMN_STAMPS=1).  Usage: python tools/stamps.py [c3|c2|c1]"""
import os, sys
os.environ["MMN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench                                   # the workload definition and synthetic generator of bench.py
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
B = int(wl["B"])
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
xs, y = bench.synthetic_batches(wl, B, B, seed=1)[0]
eng = model._get_engine(B)
dx = [torch.from_numpy(x).cuda() for x in xs]; dy = torch.from_numpy(y).cuda()
b = eng.make_batch(dx, dy, [(i, i) for i in range(len(dx))], device_nan_flags=True)
for _ in range(5):
    eng.local_step(b, 1.0, 0.003, accumulate=True)
torch.cuda.synchronize()
ptr = eng.lib.mmn_debug_buffer(eng._plan, 3, 0)
off = ptr - eng.workspace.data_ptr()
st = eng.workspace[off:off + 8 * 250].view(torch.int64).cpu().numpy()
xs_ = st[200:216].copy()
if xs_[0] > 0:
    xs_ = xs_[xs_ > 0]
    print("k_xpart (widest source): start->loop, passes.., barrier, store:", " ".join(f"{v:.2f}" for v in np.diff(xs_) / 100.0), "total", (xs_[-1] - xs_[0]) / 100.0)
f = st[150:166].copy()
if f[0] > 0:
    for wv, o in ((0, 0), (5, 8)):
        print(f"phase C, wave {wv}: z/sums {(f[o+1]-f[o])/100:.2f}  store_state {(f[o+2]-f[o+1])/100:.2f}  "
              f"issue {(f[o+3]-f[o+2])/100:.2f}  barrier wait {(f[o+4]-f[o+3])/100:.2f}")
w = st[100:112].copy()
bw = st[50:100].copy(); bw = bw[bw > 0]
if len(bw) > 1:
    print("backward chain kernel (slots 50..):", " ".join(f"{x:.2f}" for x in np.diff(bw) / 100.0), "total", (bw[-1] - bw[0]) / 100.0)
st = st[:50]
st = st[st > 0]
d = np.diff(st) / 100.0   # 100 MHz -> us
if w[0] > 0:
    print("k_wgrad block 200: start->item", (w[1] - w[0]) / 100.0, "item->loop end", (w[10] - w[1]) / 100.0,
          "loop end->lds", (w[11] - w[10]) / 100.0, "lds->end", (w[2] - w[11]) / 100.0, "total", (w[2] - w[0]) / 100.0)
print("n stamps", len(st), "total us", (st[-1] - st[0]) / 100.0)
print(" ".join(f"{x:.2f}" for x in d))
