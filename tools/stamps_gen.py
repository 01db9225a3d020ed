"""Diagnostic: per-phase timestamps of one workgroup of the generic tier's fast kernels (MMN_STAMPS=1),
bench.py's `mimic` workload."""
import os, sys
os.environ["MMN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS["mimic"]
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
xs, y = bench.synthetic_batches(wl, 4096, 4096, seed=1)[0]
eng = model._get_engine(4096)
dx = [torch.from_numpy(x).cuda() for x in xs]; dy = torch.from_numpy(y).cuda()
b = eng.make_batch(dx, dy, [(i, i) for i in range(4)], device_nan_flags=True)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
C = __import__("ctypes")
st_ptr = eng.lib.mmn_debug_buffer(eng._plan, 3, 0)
off = st_ptr - eng.workspace.data_ptr()
for _ in range(5):
    keep = eng.draw_dropout_masks(b)
    eng.local_step(b, 1.0, 0.003, accumulate=True)
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
eng.workspace[off:off + 8 * 250].zero_()
if which == "fwd":
    eng.lib.mmn_chain_fwd(eng._plan, C.byref(b), 1.0, 0.003, int(os.environ.get("WANT_GRADS", "1")), stream)
else:
    eng.lib.mmn_chain_bwd(eng._plan, C.byref(b), 0.003, stream)
torch.cuda.synchronize()
st = eng.workspace[off:off + 8 * 250].view(torch.int64).cpu().numpy()
st = st[:100]; st = st[st > 0]
d = np.diff(st) / 100.0
print(which, "n stamps", len(st), "total us", (st[-1] - st[0]) / 100.0)
print(" ".join(f"{x:.2f}" for x in d))
