"""Diagnostic: when and where every k_wgrad workgroup ran (MMN_STAMPS=1): start / end on the 100 MHz wall clock, CU and XCD."""
import os, sys
os.environ["MMN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodn_amd as mm
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
model = bench.build_model(mm, wl, torch.device("cuda"))
model.nan_policy = "device"
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host]
opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
if wl.get("per_sample"):                        # bench.py's C5 data: NaN rows not at random, a random encoder order per sample
    rng = np.random.default_rng(1)
    n_enc = len(wl["F"])
    ps = []
    for xs, y in res:
        p_miss = torch.where(y[:, :1] == 1, 0.45, 0.15).cpu().numpy()
        miss = torch.from_numpy(rng.random((B, n_enc)) < p_miss).cuda()
        for e in range(n_enc):
            xs[e][miss[:, e]] = float("nan")
        sq = torch.from_numpy(np.stack([rng.permutation(n_enc) for _ in range(B)]).astype(np.int64)).cuda()
        ps.append((xs, y, sq))
    model.per_sample = True
    class _L(list): pass
    steps = _L([ps[i % 8] for i in range(16)])
    for _ in range(3):
        model.train_epoch(steps, opt, torch.nn.CrossEntropyLoss())
else:
  steps = [res[i % 8] for i in range(64)]
  for _ in range(3):
    model._train_steps(steps, opt)              # the real step sequence (fused Adam, pre-scan blocks, graphs)
torch.cuda.synchronize()
eng = model._engine
ptr = eng.lib.mmn_debug_buffer(eng._plan, 3, 0)
off = ptr - eng.workspace.data_ptr()
raw = eng.workspace[off:off + 8 * (256 + 4096 + 1024)].view(torch.int64).cpu().numpy()
rd = raw[256 + 4096:].reshape(512, 2)
st = raw[256:256 + 4096].reshape(1024, 4)
n = int((st[:, 0] > 0).sum())
st = st[:n]
t0 = st[:, 0].min()
start = (st[:, 0] - t0) / 100.0; end = (st[:, 1] - t0) / 100.0
hw = st[:, 2]; xcc = st[:, 3] & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = xcc * 64 + se * 16 + sh * 8 + cu          # not dense, only an identity
print("workgroups", n, "kernel span", end.max(), "us; starts: median", np.median(start), "max", start.max())
dur = end - start
order = np.argsort(-end)
print("last 12 to finish: (block, start, end, dur, xcc, cu)")
for i in order[:12]:
    print(f"  {i:4d} {start[i]:6.2f} {end[i]:6.2f} {dur[i]:6.2f} xcc {xcc[i]} cu {cuid[i]}")
from collections import Counter
per_cu = Counter(cuid.tolist())
print("distinct CUs", len(per_cu), "workgroups per CU histogram", Counter(per_cu.values()))
# duration by co-residency
multi = np.array([per_cu[c] for c in cuid])
for k in sorted(set(multi.tolist())):
    sel = multi == k
    print(f"  CUs with {k} workgroups: {sel.sum()} wgs, dur mean {dur[sel].mean():.2f} max {dur[sel].max():.2f}, end max {end[sel].max():.2f}")
if n > 256:
    same = sum(int(cuid[j] == cuid[j - 256]) for j in range(256, n))
    print("second-round workgroups on the CU of block j - 256:", same, "of", n - 256)
    first = {int(cuid[j]): j for j in range(256)}
    print("  partner (first-round block on the same CU) of blocks 256..271:", [first.get(int(cuid[j]), -1) for j in range(256, min(n, 272))])
    print("  xcc of blocks 0..15:", xcc[:16].tolist(), " cu-in-xcc of blocks 0,8,16,..:", [int(cuid[j]) % 64 for j in range(0, 128, 8)])
qs = np.percentile(dur, [0, 25, 50, 75, 100])
print("duration quartiles", " ".join(f"{q:.2f}" for q in qs))
# duration by launch position (cost class follows the sorted order)
for lo in range(0, n, 32):
    sel = slice(lo, min(n, lo + 32))
    print(f"  blocks {lo:3d}+: start {start[sel].mean():5.2f} dur {dur[sel].mean():5.2f} (max {dur[sel].max():5.2f}) end {end[sel].max():5.2f}")


if os.environ.get("MMN_SIDE", "1") != "0":             # the side blocks stand in front of the work items
    R, D, E = len(wl["F"]) + 1, wl["D"], len(wl["F"])
    cols = os.environ.get("MMN_STATS_COLS", "1") != "0" and R * D + E <= 32
    ns = -(-(6 * R * D + E) // 32) if cols else 1
    print("side blocks: stats (%d) [%.2f - %.2f us], Adam coefficients [%.2f - %.2f us]; last work item ends at %.2f us" %
          (ns, start[:ns].min(), end[:ns].max(), start[ns], end[ns], end[ns + 1:].max()))
    print("  each stats block:", " ".join("[%.2f - %.2f]" % (start[i], end[i]) for i in range(ns)))
nr = int((rd[:, 0] > 0).sum()); rd = rd[:nr]
r0 = rd[0, 0]
print("k_reduce: workgroups", nr, "start of first after k_wgrad's last end:", (r0 - st[:, 1].max()) / 100.0, "us; span", (rd[:, 1].max() - r0) / 100.0)
gb = eng.lib.mmn_debug_buffer  # noqa
for i in [j for j in list(range(0, nr, 16)) + [92, 93, 94, 95, 96, nr - 1] if j < nr]:
    print(f"  block {i:3d}: start {(rd[i,0]-r0)/100.0:5.2f} end {(rd[i,1]-r0)/100.0:5.2f}")

if nr <= 94: sys.exit(0)
g = raw[120:123]
print("stats block: start -> partials summed", (g[0] - rd[94, 0]) / 100.0, "-> barrier", (g[1] - g[0]) / 100.0, "-> stats stored", (g[2] - g[1]) / 100.0,
      "-> end (epoch)", (rd[94, 1] - g[2]) / 100.0)

# workgroup 200 (a work item in the middle of the launch order): start -> record and flags in -> loop set up -> rows walked ->
# tile in LDS -> end (stamps 100, 101, 102, 110, 111 of the diagnostic buffer; its end = its entry of the per-workgroup table)
ph = raw[[100, 101, 102, 110, 111]]
if (ph > 0).all() and n > 200:
    s200 = st[200, 0]
    print("workgroup 200: " + " | ".join(f"{name} {(v - s200) / 100.0:.2f}" for name, v in
                                         zip(("kernel body", "record in", "loop set up", "rows walked", "tile summed"), ph)) +
          f" | end {(st[200, 1] - s200) / 100.0:.2f} us")
