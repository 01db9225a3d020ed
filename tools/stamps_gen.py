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
if st[160] > 0:
    print("k_dec_fb: first workgroup starts", 0.0, "| (7,1) starts", (st[100] - st[161]) / 100.0, "| last workgroup starts", (st[160] - st[161]) / 100.0,
          "ends", (st[162] - st[161]) / 100.0, "us")
st2 = st[100:160]; st2 = st2[st2 > 0]
if len(st2) > 1:
    print("k_dec_fb (tile 7, row 1): n stamps", len(st2), "total us", (st2[-1] - st2[0]) / 100.0)
    print(" ".join(f"{x:.2f}" for x in np.diff(st2) / 100.0))
st = st[:100]; st = st[st > 0]
d = np.diff(st) / 100.0
print(which, "n stamps", len(st), "total us", (st[-1] - st[0]) / 100.0)
print(" ".join(f"{x:.2f}" for x in d))
