"""us/step of the one-launch epoch kernel (csrc/mmn_epoch_small.inc) against the step-by-step path on the same batches:
`python tools/time_epoch_small.py [workload=c1] [n_batches=200]` (bench.py's workload table)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodn_amd as mm
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c1"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = wl["B"]
host = bench.synthetic_batches(wl, B * 8, B, seed=1)
res = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in host]
steps = [res[i % 8] for i in range(n)]
for use in (True, False):
    model = bench.build_model(mm, wl, torch.device("cuda"))
    model.nan_policy = "device"
    model.epoch_kernel = use
    opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
    for _ in range(4):
        model._train_steps(steps, opt)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        model._train_steps(steps, opt)
        e1.record()
        torch.cuda.synchronize()
        ts.append(((time.perf_counter() - t0) * 1e6 / n, e0.elapsed_time(e1) * 1e3 / n))
    ts.sort()
    took = bool(model.__dict__.get("_small_epochs"))
    print(f"epoch_kernel={use} (taken: {took}): median wall {ts[5][0]:.2f} us/step, device {sorted(t[1] for t in ts)[5]:.2f} us/step over {n} steps")
    if use and os.environ.get("MMN_STAMPS") == "1":       # thread 0's clock (100 MHz) at every barrier of the middle step
        eng = model._engine
        ptr = eng.lib.mmn_debug_buffer(eng._plan, 3, 0)
        off = ptr - eng.workspace.data_ptr()
        raw = eng.workspace[off:off + 8 * 128].view(torch.int64).cpu().numpy()
        k = int((raw[:64] > 0).sum())
        print("  shader clock during the step: %.0f MHz" % ((raw[64 + k - 1] - raw[64]) / ((raw[k - 1] - raw[0]) / 100.0)))
        print("  phase ends (us since the step's start):", " ".join(f"{(raw[i] - raw[0]) / 100.0:.2f}" for i in range(1, k)))
