#!/usr/bin/env python3
"""bench.py -- samples/sec of the MultiModN training step on MI355X (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[2], the configuration the metric is quoted on:
synthetic MIMIC-shaped tabular data, 4 modalities x 64 features, 3 binary tasks, state_dim 128,
encoder hidden (32, 32) relu, batch 4096 per GPU, Adam lr 1e-3, err_penalty 1, state_change 0.3.
A "step" = one full training step of one mini-batch whose inputs are already resident in HBM, driven
through the package's own batch loop (MultiModN._train_steps = the body of train_epoch): forward and
reverse chain (one fused kernel) + weight grads + reduction with the Adam step, the refresh of the
chain kernels' weight copies and the NaN scan of the NEXT batch fused in [N>1: reduction + pre-scan,
ONE RCCL all-reduce of grads + stats + NaN flags, then loss/epoch accumulation + Adam in one launch].

Other workloads (never what the driver reads): --workload c1|c2 (Titanic-shaped), c5 (per-sample missing modalities), c5m (the same with the MIMIC modules),
mimic (the MIMIC pipelines' own modules, MIMIC_MLPEncoder + MLPDecoder, on the chain kernels k_mfwd / k_mbwd with the batched decoders of k_dec_fb).

Launch:  python bench.py [--gpus N --steps K --warmup W]      (N>1 without a launcher: it starts its own N ranks)
         N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np

torch = None                       # imported in main() / where needed: the launcher parent (--gpus N without a launcher)
                                   # never loads it, let alone touches a GPU

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix) = vector peak
# the committed rocprofv3 --pmc passes roofline.traffic / mfma_busy_pct_pmc are read from (tools/collect_profiles.sh, collect_pmc_util.sh)
PROFILE_TAGS = {"c3": "r06_final", "mimic": "r06_mimic", "c5": "r06_c5", "c5m": "r06_c5m", "haim": "r06_haim", "c1": "r06_c1", "c2": "r06_c2"}
HBM_PEAK_GBS = 8000.0


# SURVEY.md section 8 shorthand.  c3 is the configuration the metric is quoted on (the default and
# the only one the driver reads); c1 / c2 are the Titanic-shaped cases: microseconds of arithmetic,
# i.e. launch-latency measurements (--workload c1|c2).
WORKLOADS = {
    "c3": dict(S=128, F=[64] * 4, H=(32, 32), D=3, B=4096, lr=1e-3, pen=(1.0, 0.3),
               text="MIMIC-shaped tabular: 4 modalities x 64 features, hidden (32,32) relu, 3 binary tasks, "
                    "state_dim 128, batch 4096 per GPU, Adam lr 1e-3, penalties 1.0/0.3"),
    "c5": dict(S=128, F=[64] * 4, H=(32, 32), D=3, B=4096, lr=1e-3, pen=(1.0, 0.3), per_sample=True,
               text="BASELINE configs[4]: the MIMIC-shaped model with per-sample missing modalities (30 % missing not at "
                    "random, NaN rows) and a random encoder order per sample, batch 4096 per GPU; a step includes the "
                    "on-device regrouping of the rows into tiles of one executed sequence (groups of 8 steps replayed as one "
                    "hipGraph, one multi-batch regrouping per group)"),
    "mimic": dict(S=128, F=[64] * 4, H=(32, 32), D=3, B=4096, lr=1e-3, pen=(1.0, 0.3), family="mimic", dec_hidden=(32, 32),
                  dropout=0.2,
                  text="SURVEY 8f #1, the MIMIC pipelines' own modules at the MIMIC shape: 4 x MIMIC_MLPEncoder(64 features + "
                       "state 128 -> 32 -> 32 -> 128, relu on every layer, dropout 0.2 on cat[x, state], masks drawn on the "
                       "device inside the step) and 3 x MLPDecoder(128 -> 32 -> 32 -> 2), batch 4096 per GPU, Adam lr 1e-3, "
                       "penalties 1.0/0.3; chain kernels k_mfwd / k_mbwd, decoders in k_dec_fb"),
    "c5m": dict(S=128, F=[64] * 4, H=(32, 32), D=3, B=4096, lr=1e-3, pen=(1.0, 0.3), per_sample=True, family="mimic",
                dec_hidden=(32, 32), dropout=0.2,
                text="BASELINE configs[4] with the modules the reference's MNAR pipeline builds (MIMIC_MLPEncoder + MLPDecoder, "
                     "pipelines/mimic/mimic_single_task_mnar_missingness_pipeline.py:163-165): per-sample missing modalities (30 % "
                     "not at random) and encoder order, dropout 0.2 drawn on the device, batch 4096 per GPU; k_mfwd / k_dec_fb / "
                     "k_mbwd on the regrouped 16-row tiles, groups of 8 steps replayed as one hipGraph"),
    "haim": dict(S=50, F=[6, 1024, 768, 99], H=(32, 32), D=2, B=16, lr=1e-3, pen=(1.0, 0.0), family="mimic", dec_hidden=(32, 32),
                 dropout=0.2,
                 text="the reference's real MIMIC configuration (pipelines/mimic/mimic_multi_task_pipeline.py:53-83,118-119): 4 x "
                      "MIMIC_MLPEncoder over the sources de / vd / n_ech / ts_ce (6 / 1024 / 768 / 99 features + state 50 -> 32 -> 32 -> 50, "
                      "dropout 0.2) and 2 x MLPDecoder(50 -> 32 -> 32 -> 2), batch 16, Adam lr 1e-3, penalties 1.0/0.0"),
    "c2": dict(S=64, F=[3, 2], H=(5, 5), D=2, B=512, lr=1e-2, pen=(0.7, 0.3),
               text="Titanic-shaped, 2 encoders (features split 3+2), hidden (5,5) relu, 2 binary tasks, state_dim 64, "
                    "batch 512, Adam lr 1e-2, penalties 0.7/0.3 (latency-bound)"),
    "c1": dict(S=32, F=[6], H=(5, 5), D=1, B=32, lr=1e-2, pen=(0.7, 0.3),
               text="Titanic MLP pipeline shape: 1 encoder x 6 features, hidden (5,5) relu, 1 binary task, state_dim 32, "
                    "batch 32, Adam lr 1e-2, penalties 0.7/0.3 (latency-bound)"),
}


def oracle_spec(O, w):
    """The workload as the CPU oracle describes it (cpu_baseline leg only)."""
    if w.get("family") == "mimic":
        return O.ModelSpec(w["S"], [O.EncoderSpec(f, tuple(w["H"]), O.ACT_RELU, kind="mimic", dropout=w["dropout"])
                                    for f in w["F"]], w["D"], *w["pen"],
                           decoders=[O.DecoderSpec("mlp", tuple(w["dec_hidden"])) for _ in range(w["D"])])
    return O.ModelSpec(w["S"], [O.EncoderSpec(f, tuple(w["H"]), O.ACT_RELU) for f in w["F"]], w["D"], *w["pen"])


def build_model(mm, w, device):
    """The workload's model on the product surface: MLPEncoder / LogisticDecoder / MultiModN with
    torch's default initialisation under a fixed seed (the same weights on every rank)."""
    import torch
    torch.manual_seed(0)
    if w.get("family") == "mimic":      # mimic_multi_task_pipeline.py:118-119
        enc = [mm.MIMIC_MLPEncoder(w["S"], f, tuple(w["H"]), dropout=w["dropout"]) for f in w["F"]]
        dec = [mm.MLPDecoder(w["S"], tuple(w["dec_hidden"]), 2) for _ in range(w["D"])]
    else:
        enc = [mm.MLPEncoder(w["S"], f, tuple(w["H"])) for f in w["F"]]
        dec = [mm.LogisticDecoder(w["S"]) for _ in range(w["D"])]
    return mm.MultiModN(w["S"], enc, dec, w["pen"][0], w["pen"][1], device=device)


def synthetic_batches(w, n_rows, batch_size, seed):
    """SURVEY 8d generator: standard-normal float32 features, learnable binary targets
    y_d = 1[x . w_d + 0.5 eps > 0]; returns [(list of [B, F_k] arrays, [B, D] int64)]."""
    rng = np.random.default_rng(seed)
    Fs = w["F"]
    X = rng.standard_normal((n_rows, sum(Fs))).astype(np.float32)
    wt = rng.standard_normal((sum(Fs), w["D"])).astype(np.float32)
    y = ((X @ wt + 0.5 * rng.standard_normal((n_rows, w["D"])).astype(np.float32)) > 0).astype(np.int64)
    offs = np.cumsum([0] + list(Fs))
    return [([X[s:s + batch_size, offs[k]:offs[k + 1]].copy() for k in range(len(Fs))], y[s:s + batch_size].copy())
            for s in range(0, n_rows, batch_size)]


def flops_per_sample_of(w):
    """Algorithmic FLOPs (MAC = 2) per sample of each launch (SURVEY.md section 8d formulas)."""
    S, E, D = w["S"], len(w["F"]), w["D"]
    fwd = bwd = wg = 0
    if w.get("family") == "mimic":
        # encoder: (F+S) x H1, H1 x H2, ..., Hn x S; backward through the same products except the x columns of the
        # first layer (no grad flows to x); decoder on each of the E+1 states: S x Hd1, ..., Hdn x 2 (forward, backward
        # and weight gradient alike)
        for f in w["F"]:
            dims = [f + S] + list(w["H"]) + [S]
            full = sum(a * b for a, b in zip(dims, dims[1:]))
            fwd += full
            wg += full
            bwd += full - f * dims[1]
        dd = [S] + list(w["dec_hidden"]) + [2]
        dec = (E + 1) * D * sum(a * b for a, b in zip(dd, dd[1:]))
        return {"k_chain_fwd": 2 * (fwd + dec), "k_chain_bwd": 2 * (bwd + dec), "k_wgrad": 2 * (wg + dec), "decoders_one_way": 2 * dec}
    for f in w["F"]:
        dims = [f] + list(w["H"])
        hidden = sum(a * b for a, b in zip(dims, dims[1:]))
        last = (dims[-1] + S) * S
        fwd += hidden + last
        wg += hidden + last
        bwd += last + (hidden - dims[0] * dims[1] if len(dims) > 1 else 0)   # no grad flows to x
    dec = (E + 1) * D * 2 * S
    return {"k_chain_fwd": 2 * (fwd + dec), "k_chain_bwd": 2 * (bwd + dec), "k_wgrad": 2 * (wg + dec)}


def cpu_faithful_step(O, spec, batch_size):
    """ONE step in the reference's own per-step style (SURVEY.md 8d "faithful" mode): the oracle's
    arithmetic plus the Python-level work multimodn.py does around it - `any(x.isnan().flatten())`
    iterating every element of every modality tensor (:168) and builtin `sum()` over per-sample
    comparison tensors for every grid cell (:147,183).  Reported beside the vectorised baseline; it
    is what the reference's interpreter-bound loop costs, not a target."""
    import torch
    params = O.init_params(spec, 0)
    xs, y = O.synthetic_batches(spec, batch_size, batch_size, seed=4, learnable=False)[0]
    t0 = time.perf_counter()
    for x in xs:
        any(torch.from_numpy(x).isnan().flatten())                      # multimodn.py:168
    r = O.forward_backward(params, spec, xs, y)
    yt = torch.from_numpy(y)
    for _ in range(spec.E + 1):
        for d in range(spec.D):
            sum(yt[:, d] == yt[:, d])                                   # multimodn.py:147,183 (python sum over a tensor)
    O.Adam(1e-3).step(params, r.grads)
    el = time.perf_counter() - t0
    return {"value": batch_size / el, "unit": "samples/s", "sample": f"1 step of batch {batch_size}, {el:.2f} s"}


def cpu_match(O, mm, w, device, batch_size, steps=3):
    """The metric's second half, "CPU-match delta-loss": the HIP path and the CPU oracle take the same
    `steps` training steps (same initial weights, same batches, Adam) at the workload's full size;
    reported are the largest relative difference of the (E+1) x D loss grid over the steps and of
    the trained weights (relative to each tensor's max).  Target: <= 1e-5 on the loss.

    Trained weights additionally against the yardstick that does not depend on a constant: the same steps in FLOAT64
    (exact arithmetic for this purpose).  Adam divides by sqrt(v), so on coordinates whose gradient is of rounding size
    ANY two fp32 implementations end up O(lr) apart; what can be asked of the HIP path is to sit no further from the fp64
    trajectory than the fp32 CPU oracle does.  Per tensor (tests/helpers.py::assert_within_fp32_noise): within 2e-5 of the
    fp32 oracle outright, or |w_hip - w_fp64| <= 4 |w_oracle32 - w_fp64|; `ok` = every compared tensor passes, and the
    bench exits non-zero otherwise.  Tensors whose gradient crossed a relu kink differently on the HIP path than in the
    float64 replay (the activation pattern h > 0 of a hidden layer differs on some sample: the loss is not differentiable
    there, one sample's whole contribution comes or goes) are listed in `kink_flipped` and not compared; the state-update
    layers, the decoders and the init state have no relu in front of them and are always compared.  (MIMIC family: the
    state itself is a relu output and feeds every later encoder; its activations are not read back here, the ratio is
    reported but `ok` is not derived from it.)"""
    import torch
    model = build_model(mm, w, device)
    spec = oracle_spec(O, w)
    params = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    params64 = {n: v.astype(np.float64) for n, v in params.items()}
    assert list(params) == spec.param_names()
    opt = mm.optim.Adam(list(model.parameters()), w["lr"])
    oopt, oopt64 = O.Adam(w["lr"]), O.Adam(w["lr"])
    eng = model._get_engine(batch_size)
    eng.begin_sequence()
    eng.assign_grads(None)
    alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
    worst, worst64 = 0.0, 0.0
    mimic = w.get("family") == "mimic"
    flipped = set()
    for xs, y in synthetic_batches(w, batch_size * steps, batch_size, seed=77):
        dx = [torch.from_numpy(x).to(device) for x in xs]
        dy = torch.from_numpy(y).to(device)
        b = eng.make_batch(dx, dy, [(k, k) for k in range(len(xs))], device_nan_flags=False)
        keep = eng.draw_dropout_masks(b) if eng.dropout_encoders else []     # both sides take the device's masks
        masks = {e: mk.cpu().numpy() for (e, _, _), mk in zip(eng.dropout_encoders, keep)} or None
        eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt)
        opt.step()
        got = eng.step_values()["err_loss"]
        r = O.forward_backward(params, spec, xs, y, drop_masks=masks)
        oopt.step(params, r.grads)
        r64 = O.forward_backward(params64, spec, xs, y, drop_masks=masks, dtype=np.float64, keep_states=True)
        if not mimic:                                        # relu kinks crossed differently than in the float64 replay
            for (e, l), h64 in r64.hidden.items():
                h_hip = eng.debug_tensor(6, e * mm.hip.MAX_LAYERS + l, eng.max_batch, h64.shape[1])[:batch_size].cpu().numpy()
                if ((h_hip > 0) != (h64 > 0)).any():
                    for j in range(l + 1):
                        flipped.update({f"encoders.{e}.layers.{j}.weight", f"encoders.{e}.layers.{j}.bias"})
        oopt64.step(params64, r64.grads)
        worst = max(worst, float(np.max(np.abs(got - r.err_loss)) / np.max(np.abs(r.err_loss))))
        worst64 = max(worst64, float(np.max(np.abs(got - r64.err_loss)) / np.max(np.abs(r64.err_loss))))
    dw, ratio, dw64_hip, dw64_cpu = 0.0, 0.0, 0.0, 0.0
    worst_t = None
    ss_hip, ss_cpu, n_el = 0.0, 0.0, 0
    failed, compared = [], 0
    for n, p in model.named_parameters():
        hipw = p.detach().cpu().numpy().astype(np.float64)
        scale = max(np.max(np.abs(params64[n])), 1e-30)
        e_hip = float(np.max(np.abs(hipw - params64[n])))
        e_cpu = float(np.max(np.abs(params[n].astype(np.float64) - params64[n])))
        d32 = float(np.max(np.abs(hipw - params[n])) / max(np.max(np.abs(params[n])), 1e-30))
        dw = max(dw, d32)
        if n in flipped:
            continue
        compared += 1
        dw64_hip, dw64_cpu = max(dw64_hip, e_hip / scale), max(dw64_cpu, e_cpu / scale)
        ss_hip += float(np.sum(((hipw - params64[n]) / scale) ** 2))
        ss_cpu += float(np.sum(((params[n].astype(np.float64) - params64[n]) / scale) ** 2))
        n_el += hipw.size
        r_t = e_hip / e_cpu if e_cpu > 0 else (0.0 if e_hip == 0 else float("inf"))
        if d32 > 2e-5 and r_t > 4.0:
            failed.append({"tensor": n, "ratio": r_t, "vs_fp32_oracle": d32})
        if r_t > ratio:
            ratio, worst_t = r_t, {"tensor": n, "ratio": r_t, "hip": e_hip / scale, "cpu_fp32_oracle": e_cpu / scale}
    return {"delta_loss": worst, "delta_loss_vs_fp64": worst64, "delta_weights": dw,
            "delta_weights_vs_fp64": {"hip": dw64_hip, "cpu_fp32_oracle": dw64_cpu,
                                      "fp64_ratio": ratio, "worst_tensor": worst_t,
                                      "tensors_compared": compared, "kink_flipped": sorted(flipped), "failed": failed,
                                      # the same distances as root mean squares over ALL compared weights (a max is one outlier)
                                      "rms": {"hip": (ss_hip / max(n_el, 1)) ** 0.5, "cpu_fp32_oracle": (ss_cpu / max(n_el, 1)) ** 0.5,
                                              "ratio": (ss_hip / ss_cpu) ** 0.5 if ss_cpu > 0 else 0.0}},
            # the gate (exit code 4): loss cells within 1e-5; trained weights - MLPEncoder family: no tensor further from the
            # float64 trajectory than 4x the fp32 oracle is, over at least 12 tensors without a relu-kink flip (the tests' floor);
            # MIMIC modules (no hidden activations to look for kink flips in): the RMS distance over all weights within 4x
            "ok": bool(worst < 1e-5 and ((not failed and compared >= min(12, len(params))) if not mimic
                                         else (ss_cpu > 0 and (ss_hip / ss_cpu) ** 0.5 <= 4.0))),
            "ok_rule": "loss 1e-5; weights: per-tensor 4x rule over >= 12 tensors (MIMIC modules: RMS ratio <= 4)",
            "steps": steps, "batch": batch_size,
            "against": "numpy fp32 oracle (oracle/multimodn_oracle.py), itself pinned to the reference by tests/golden; "
                       "fp64 = the same oracle in float64"}


def _curve_replay_job(w, batch_size, epochs, steps, params, dtype_name):
    """The numpy oracle's replay of cpu_match_curve's steps in one precision (a worker process: CPU only)."""
    sys.path.insert(0, REPO)
    from oracle import multimodn_oracle as O                 # the checker: imported for the cpu_match leg only
    dtype = np.dtype(dtype_name).type
    spec = oracle_spec(O, w)
    batches = synthetic_batches(w, batch_size * steps, batch_size, seed=55)
    p = {n: np.asarray(v, dtype).copy() for n, v in params.items()}
    o = O.Adam(w["lr"])
    return p, [O.train_epoch(p, spec, batches, o, dtype=dtype) for _ in range(epochs)]


def cpu_match_curve(O, mm, w, device, batch_size, epochs=3, steps=64):
    """The loss CURVE half of "CPU-match" (VERDICT r4 #4): `epochs` x `steps` training steps through the public
    MultiModN.train_epoch (device-resident batches, the fused Adam, hipGraph replay) against the numpy oracle's replay of the
    same steps in float64 and, as the yardstick of what fp32 can do at all, in float32.  Reported per epoch: the largest
    relative difference of the History loss grid and of the state-change vector to the float64 replay (HIP path / fp32 oracle),
    and how many trained weight tensors sit further from the float64 trajectory than 4x the fp32 oracle does."""
    import torch
    model = build_model(mm, w, device)
    spec = oracle_spec(O, w)
    params = {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    opt = mm.optim.Adam(list(model.parameters()), w["lr"])
    hist = mm.MultiModNHistory([f"t{d}" for d in range(w["D"])])
    batches = synthetic_batches(w, batch_size * steps, batch_size, seed=55)
    loader = [([torch.from_numpy(x).to(device) for x in xs], torch.from_numpy(y).to(device)) for xs, y in batches]
    crit = torch.nn.CrossEntropyLoss()
    # the two replays (float64, float32: ~30 s of single-threaded numpy each) run side by side in worker processes (spawned:
    # this process holds a GPU context; a few BLAS threads each - the oracle's products are small) while the GPU trains
    import concurrent.futures as cf
    import multiprocessing as mp
    t0 = time.perf_counter()
    keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ[k] = "4"
    try:
        pool = cf.ProcessPoolExecutor(2, mp_context=mp.get_context("spawn"))
        f64 = pool.submit(_curve_replay_job, w, batch_size, epochs, steps, params, "float64")
        f32 = pool.submit(_curve_replay_job, w, batch_size, epochs, steps, params, "float32")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for _ in range(epochs):
        model.train_epoch(loader, opt, crit, hist)
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))

    try:
        p64, e64 = f64.result()
        p32, e32 = f32.result()
        pool.shutdown(wait=False)
    except Exception:                                         # (no worker processes in this environment: replay here, one after the other)
        p64, e64 = _curve_replay_job(w, batch_size, epochs, steps, params, "float64")
        p32, e32 = _curve_replay_job(w, batch_size, epochs, steps, params, "float32")
    per_epoch, ok = [], True
    for ep in range(epochs):
        dl, dl32 = rel(hist.loss["train"][ep], e64[ep].loss), rel(e32[ep].loss, e64[ep].loss)
        ds, ds32 = rel(hist.state_change_loss[ep], e64[ep].state_change), rel(e32[ep].state_change, e64[ep].state_change)
        da = float(np.abs(np.asarray(hist.accuracy["train"][ep]) - e64[ep].accuracy).max())
        per_epoch.append({"epoch": ep, "delta_loss_vs_fp64": dl, "fp32_oracle_delta_loss_vs_fp64": dl32,
                          "delta_state_change_vs_fp64": ds, "fp32_oracle_delta_state_change_vs_fp64": ds32,
                          "delta_accuracy_vs_fp64": da})
        ok = ok and dl < max(1e-5, 4.0 * dl32) and ds < max(1e-5, 4.0 * ds32)
    far, n_t = [], 0
    for n, p in model.named_parameters():
        n_t += 1
        hw = p.detach().cpu().numpy().astype(np.float64)
        if rel(hw, p32[n]) <= 2e-5:
            continue
        e_hip, e_cpu = float(np.abs(hw - p64[n]).max()), float(np.abs(p32[n].astype(np.float64) - p64[n]).max())
        if e_hip > 4.0 * e_cpu:
            far.append({"tensor": n, "ratio": e_hip / max(e_cpu, 1e-300)})
    return {"epochs": epochs, "steps_per_epoch": steps, "batch": batch_size, "per_epoch": per_epoch,
            "weight_tensors": n_t, "tensors_beyond_4x_fp32_noise": far,
            "ok": bool(ok and len(far) <= n_t // 4),
            "ok_rule": "per epoch: History loss and state change within 1e-5 of the float64 replay or within 4x the fp32 oracle's own distance; "
                       "at most a quarter of the weight tensors further from the float64 trajectory than 4x the fp32 oracle",
            "oracle_wall_s": round(time.perf_counter() - t0, 1)}


def cpu_baseline(O, spec, batch_size, budget_s=15.0):
    """The numpy oracle (a port of the reference step, oracle/multimodn_oracle.py) timed on the
    host cores on a bounded sample of the same workload."""
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    params = O.init_params(spec, 0)
    batches = O.synthetic_batches(spec, batch_size * 2, batch_size, seed=3, learnable=False)
    opt = O.Adam(1e-3)
    r = O.forward_backward(params, spec, *batches[0])          # warm-up
    opt.step(params, r.grads)
    rng = np.random.default_rng(9)

    def draw():                                                # nn.Dropout of the MIMIC encoders, drawn per step like the reference
        return {e: ((rng.random((batch_size, enc.n_features + spec.state_size), dtype=np.float32) >= enc.dropout)
                    / np.float32(1 - enc.dropout)).astype(np.float32)
                for e, enc in enumerate(spec.encoders) if enc.kind == "mimic" and enc.dropout > 0} or None
    n, t0 = 0, time.perf_counter()
    while True:
        xs, y = batches[n % 2]
        r = O.forward_backward(params, spec, xs, y, drop_masks=draw())
        opt.step(params, r.grads)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": n * batch_size / el, "unit": "samples/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} training steps of batch {batch_size} (numpy fp32 oracle: fwd+bwd+Adam), {el:.1f} s"}


def cpu_torch_vectorised(mm, w, batch_size, budget_s=10.0):
    """BASELINE.md section 4, mode (ii) "vectorised" (SURVEY 8d): the package's own nn.Module forwards - the reference's
    module arithmetic, multimodn_amd/encoders.py / decoders.py / state.py - driven on the HOST cores by a batch loop with
    tensor-level NaN test and reductions, in-memory batches, torch.autograd, torch.optim.Adam, nn.CrossEntropyLoss: what
    the reference's own modules reach on this host once its Python-level any() / sum() are out of the way (survey probe:
    118 k samples/s on 8 cores).  MLPEncoder family only (the MIMIC modules draw their dropout inside forward: same code)."""
    import torch
    torch.manual_seed(0)
    cpu = torch.device("cpu")
    S, D, E = w["S"], w["D"], len(w["F"])
    if w.get("family") == "mimic":
        enc = [mm.MIMIC_MLPEncoder(S, f, tuple(w["H"]), dropout=w["dropout"], device=cpu) for f in w["F"]]
        dec = [mm.MLPDecoder(S, tuple(w["dec_hidden"]), 2, device=cpu) for _ in range(D)]
    else:
        enc = [mm.MLPEncoder(S, f, tuple(w["H"]), device=cpu) for f in w["F"]]
        dec = [mm.LogisticDecoder(S, device=cpu) for _ in range(D)]
    init = mm.TrainableInitState(S, cpu)
    mods = torch.nn.ModuleList(enc + dec + [init]).train()
    opt = torch.optim.Adam(mods.parameters(), w["lr"])
    crit = torch.nn.CrossEntropyLoss()
    alpha, beta = float(w["pen"][0]), 0.01 * float(w["pen"][1])
    batches = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in synthetic_batches(w, batch_size * 2, batch_size, seed=3)]

    def step(xs, y):
        opt.zero_grad()
        state = init(y.shape[0])
        err, sc = [], []
        counts = torch.zeros(E + 1, D, 5)

        def decode(row, st):
            for d in range(D):
                o = dec[d](st)
                err.append(crit(o, y[:, d]))
                pred = o.argmax(dim=1)
                t = y[:, d]
                counts[row, d] = torch.stack([(pred == t).sum(), ((pred == 1) & (t == 1)).sum(), ((pred == 0) & (t == 0)).sum(),
                                              ((pred == 1) & (t == 0)).sum(), ((pred == 0) & (t == 1)).sum()]).float()
        decode(0, state)
        for e in range(E):
            if bool(torch.isnan(xs[e]).any()):               # the whole-batch skip of multimodn.py:168, as ONE tensor op
                continue
            old = state
            state = enc[e](state, xs[e])
            sc.append(((state - old) ** 2).mean())
            decode(e + 1, state)
        loss = alpha * sum(err) / (D * (E + 1)) + beta * sum(sc) / E
        loss.backward()
        opt.step()
    # intra-op threads: on a many-core host ATen's pool at its default size (one thread per core) loses to a small pool on
    # tensors of this size (measured on the 128-core box: 18 k samples/s with 128 threads); a baseline should be the
    # best the host does, so a few pool sizes get a second each and the best one the rest of the budget
    n_default = torch.get_num_threads()
    step(*batches[0])                                         # warm-up
    best_thr, best_rate = n_default, 0.0
    try:
        for thr in sorted({t for t in (4, 8, 16, 32, 64, n_default) if t <= n_default}):
            torch.set_num_threads(thr)
            step(*batches[0])
            k, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 1.0:
                step(*batches[k % 2])
                k += 1
            rate = k / (time.perf_counter() - t1)
            if rate > best_rate:
                best_thr, best_rate = thr, rate
        torch.set_num_threads(best_thr)
        n, t0 = 0, time.perf_counter()
        while True:
            step(*batches[n % 2])
            n += 1
            el = time.perf_counter() - t0
            if el > max(budget_s - 6.0, 3.0) or n >= 400:
                break
    finally:
        torch.set_num_threads(n_default)
    model_name = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model_name = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n * batch_size / el, "unit": "samples/s", "cores": int(best_thr), "kind": "port",
            "os_cpu_count": os.cpu_count(), "torch_default_threads": int(n_default), "cpu_model": model_name, "torch": torch.__version__,
            "sample": f"{n} training steps of batch {batch_size} ({el:.1f} s): multimodn_amd's nn.Module forwards on the host, "
                      f"torch.autograd, torch.optim.Adam, CrossEntropyLoss, tensor-level NaN test and counters"}


def spawn_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher: this process never touches the GPU (no torch import, no
    HIP call); it starts N fresh worker processes - one rank per GPU, rendezvous on 127.0.0.1 - and passes rank 0's JSON
    line through.  ALL children are polled: the first one that exits non-zero takes the others down with it (a rank that
    died would otherwise leave the rest waiting in the collective until the caller's timeout), and the launcher returns
    non-zero within seconds.  Every rank's stderr goes to bench_rank<r>.err next to this file's working directory.
    (Under `python -m torch.distributed.run` the ranks already exist and this is never reached.)"""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", MMN_BENCH_SPAWNED="1")
        try:
            ef = open(f"bench_rank{r}.err", "wb")
        except OSError:
            ef = tempfile.TemporaryFile()
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=ef))
    rc, failed = 0, None
    while True:
        codes = [p_.poll() for p_ in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed is not None:
        for p_ in procs:                                     # fresh processes of ours, by PID: never a pattern kill
            if p_.poll() is None:
                p_.terminate()
        t_end = time.time() + 10
        for p_ in procs:
            try:
                p_.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p_.kill()
        sys.stderr.write(f"bench.py: rank {failed} exited with code {rc}; the other ranks were terminated "
                         f"(stderr of every rank: bench_rank<r>.err)\n")
        try:
            errs[failed].flush()
            with open(f"bench_rank{failed}.err", "rb") as fh:
                sys.stderr.write(fh.read().decode(errors="replace")[-4000:])
        except OSError:
            pass
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    for ef in errs:
        ef.close()
    return rc


def rows_sweep(batches=(8192, 16384, 65536), steps=40, warmup=10):
    """the headline workload at more rows per GPU (every CU then runs 2 / 4 / 16 tiles of the chain kernel one
    after the other): step time, samples/s, the chain kernel's launch time and fraction of the fp32 MFMA roof.  Child
    processes, like the secondary workloads."""
    import subprocess
    res = {}
    for bsz in batches:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c3", "--batch", str(bsz), "--steps", str(steps), "--warmup", str(warmup),
               "--resident-batches", "4", "--no-cpu-baseline", "--no-public-path", "--no-secondary", "--preroll", "0.3"]
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                res[str(bsz)] = {"error": f"exit code {pr.returncode}", "stderr_tail": pr.stderr[-400:]}
                continue
            d = json.loads(line[-1])
            res[str(bsz)] = {"rows_per_gpu": bsz, "us_per_step": d["ms_per_step"] * 1e3, "value": d["value"], "unit": d["unit"],
                             "dominant_kernel": d["roofline"]["kernel"], "roofline_frac": d["roofline"]["frac"],
                             "step_frac_of_fp32_roof": d["roofline"]["step_frac_of_fp32_roof"],
                             "avg_launch_us": d["roofline"]["avg_launch_us"]}
        except Exception as ex:
            res[str(bsz)] = {"error": repr(ex)[:300]}
    return res


def secondary_workloads(names=("mimic", "c5", "c5m", "c1", "c2", "haim"), steps=40, warmup=10):
    """The other workloads through the same entry point, each as a CHILD process of this one (a fresh process: the parent
    has initialised the GPU and must not exec): `python bench.py --workload <w> --steps 40 --warmup 10` without CPU legs.
    Reported per workload: samples/s, ms per step, launch mode, per-kernel HIP-event times."""
    import subprocess
    res = {}
    for nm in names:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", nm, "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-public-path", "--no-secondary", "--preroll", "0.3"]
        try:
            t0 = time.perf_counter()
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                res[nm] = {"error": f"exit code {pr.returncode}", "stderr_tail": pr.stderr[-400:]}
                continue
            d = json.loads(line[-1])
            res[nm] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": steps, "warmup": warmup,
                       "workload": d["config"]["workload"], "launch": d["config"]["launch"],
                       "dominant_kernel": d["roofline"]["kernel"], "roofline_frac": d["roofline"]["frac"],
                       "avg_launch_us": d["roofline"]["avg_launch_us"], "child_wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as ex:                              # a secondary leg never takes the headline line down
            res[nm] = {"error": repr(ex)[:300]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="c3",
                    help="c3 = BASELINE.json configs[2] (the metric's configuration); c1/c2 = Titanic-shaped, latency-bound")
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU per step (default: the workload's batch)")
    ap.add_argument("--resident-batches", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--graph-steps", type=int, default=8, help="consecutive steps captured into one hipGraph")
    ap.add_argument("--graph-steps-next", type=int, default=0, help="... behind the first group of a call (0: the package's default)")
    ap.add_argument("--optimizer", choices=("hip", "torch"), default="hip",
                    help="hip: multimodn_amd.optim.Adam (fused into the step's last launch); torch: torch.optim.Adam(fused, capturable)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, what the driver's scaling run uses): every GPU gets the workload's batch; "
                         "strong: the workload's batch is split over the GPUs (SURVEY 8e: the mode whose loss curve "
                         "must equal the single-device one)")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing aid: take the data-parallel code path (process group, all-reduce, separate Adam) "
                         "even with one rank")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI (the product); gloo: testing aid, the reduce buffer travels through the host")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing aid: every rank uses GPU 0 (RCCL refuses two ranks per device: combine with --dist-backend gloo)")
    ap.add_argument("--preroll", type=float, default=0.5, help="seconds of untimed load before the warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-public-path", action="store_true", help="skip the train_epoch-over-DeviceResidentLoader leg")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-curve", action="store_true", help="skip cpu_match's 3 x 64-step loss curve (about a minute of oracle time)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (mimic, c5, c1, c2: child processes of 40 steps each); profiling runs use this")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:          # plain `python bench.py --gpus N`: be our own launcher
        raise SystemExit(spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC only on this stack: RCCL's peer mappings need it)
    global torch
    import torch
    import multimodn_amd as mm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    dp = world > 1 or args.force_dist
    if dp:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                  # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
        if os.environ.get("MMN_BENCH_FAIL_RANK") == str(rank):   # testing aid: this rank dies behind the rendezvous
            os._exit(7)

    wl = WORKLOADS[args.workload]
    n_enc = len(wl["F"])
    B = args.batch or wl["B"]
    if args.scaling == "strong":
        B = max(1, B // max(world, 1))                       # rows per GPU; the global batch stays the workload's
    model = build_model(mm, wl, dev)
    model.nan_policy = "device"
    model.replay_steps = not args.no_graph
    model.REPLAY_GROUP = max(1, args.graph_steps)
    if args.graph_steps_next:
        model.REPLAY_GROUP_NEXT = max(1, args.graph_steps_next)
    per_sample = bool(wl.get("per_sample"))
    model.per_sample = per_sample
    if dp:
        model.enable_data_parallel()
    eng = model._get_engine(B)
    if args.optimizer == "hip":
        # optimizer.step() inside the step's last launch (single GPU) / in the launch behind the all-reduce
        opt = mm.optim.Adam(list(model.parameters()), wl["lr"])
    else:
        # torch's multi-tensor fused Adam (one block per 64K-element chunk per tensor: 15 us for the
        # 31 tensors, plus a foreach add for the step counters)
        opt = torch.optim.Adam(list(model.parameters()), wl["lr"], fused=True, capturable=True)
    crit = torch.nn.CrossEntropyLoss()

    # synthetic data resident in HBM (weak scaling: every rank its own B rows per step)
    host = synthetic_batches(wl, B * args.resident_batches, B, seed=100 + rank)
    resident = []
    for xs, y in host:
        resident.append(([torch.from_numpy(x).to(dev) for x in xs], torch.from_numpy(y).to(dev)))
    if per_sample:
        # modality e is missing (NaN row) with probability 0.45 if y_0 = 1 else 0.15
        # (mean 0.3); every sample carries its own random permutation of the encoders
        rng = np.random.default_rng(1 + rank)
        with_seq = []
        for xs, y in resident:
            p_miss = torch.where(y[:, :1] == 1, 0.45, 0.15).cpu().numpy()
            if os.environ.get("MMN_BENCH_PS_UNIFORM"):      # diagnostic: per-sample mode on data where nothing varies
                p_miss = p_miss * 0.0
            miss = torch.from_numpy(rng.random((B, n_enc)) < p_miss).to(dev)
            for e in range(n_enc):
                xs[e][miss[:, e]] = float("nan")
            sq = torch.from_numpy(np.stack([rng.permutation(n_enc) if not os.environ.get("MMN_BENCH_PS_UNIFORM") else np.arange(n_enc)
                                            for _ in range(B)]).astype(np.int64)).to(dev)
            with_seq.append((xs, y, sq))
        resident = with_seq
    alpha, beta = float(model.err_penalty), float(model.state_change_penalty)

    def steps_list(n, first=0):
        """n consecutive mini-batches in the batch format train_epoch unpacks, cycling through the resident ones."""
        return [resident[(first + i) % len(resident)] for i in range(n)]

    class _Sized(list):
        pass

    def run_steps(n, first=0):
        """n training steps through the package's own batch loop: MultiModN.train_epoch's body (look-ahead ingest,
        pre-scan, hipGraph groups, data-parallel protocol), without the History bookkeeping of an epoch's end."""
        if n <= 0:
            return
        if per_sample:
            model.train_epoch(_Sized(steps_list(n, first)), opt, crit)
        else:
            model._train_steps(_Sized(steps_list(n, first)), opt)

    group = 1 if (dp or args.no_graph or per_sample or args.optimizer != "hip") else max(1, args.graph_steps)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed pre-roll, before the W warm-up steps the contract asks for: the GPU leaves its idle power
    # state only after tens of milliseconds of load; without this the first timed steps of a short run
    # execute at a fraction of the clock (measured: 76 us/step steady, up to 400 us/step averaged over a
    # cold 200-step region).  It runs the very step lists that are timed below, so their hipGraph groups are
    # captured here (second sighting) and the timed region only replays.
    run_steps(args.warmup)
    run_steps(args.steps, args.warmup)
    torch.cuda.synchronize()
    t_ramp = time.perf_counter()
    while True:
        go = time.perf_counter() - t_ramp < args.preroll
        if dp:                                               # every rank must run the same number of collectives:
            flag = torch.tensor([1 if go else 0], device=dev if args.dist_backend == "nccl" else "cpu")    # rank 0's clock decides
            dist.broadcast(flag, src=0)
            go = bool(flag.item())
        if not go:
            break
        run_steps(args.steps, args.warmup)
        torch.cuda.synchronize()
    run_steps(args.warmup)
    n_coll = {"all_reduce": 0, "other": 0}                   # COUNTED: every collective torch.distributed issues in the timed region
    if dp:
        import torch.distributed.distributed_c10d as c10d
        _orig = {}

        def _count(name, key):
            fn = getattr(dist, name)
            _orig[name] = fn

            def wrapped(*a_, **k_):
                n_coll[key] += 1
                return fn(*a_, **k_)
            setattr(dist, name, wrapped)
        _count("all_reduce", "all_reduce")
        for nm in ("broadcast", "all_gather", "all_gather_into_tensor", "reduce_scatter", "reduce_scatter_tensor", "reduce",
                   "all_to_all", "all_to_all_single", "gather", "scatter"):
            if hasattr(dist, nm):
                _count(nm, "other")
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    if dp:
        for nm, fn in _orig.items():
            setattr(dist, nm, fn)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    replayed = bool(getattr(eng, "_graph_hits", 0))

    # distribution of the step time: HIP events around groups of `group` steps
    dist_us = None
    if world == 1 and not dp:
        samples = []
        n_ev = max(group, 8)
        run_steps(n_ev)
        run_steps(n_ev)
        for rep_i in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_steps(n_ev)
            e1.record()
            torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3 / n_ev)
        dist_us = {"median": float(np.median(samples)), "p10": float(np.percentile(samples, 10)),
                   "p90": float(np.percentile(samples, 90)), "groups": len(samples), "steps_per_group": n_ev,
                   "note": "every sample is ONE _train_steps call of steps_per_group steps, i.e. it carries the whole per-sequence "
                           "cost (epoch reset, repack + scan launch, idle until the first graph is out: ~50-70 us) over few "
                           "steps; the spread is the information, the level is ms_per_step's"}

    # ---- the public entry point itself: MultiModN.train_epoch over a DeviceResidentLoader, History
    #      appended every epoch.  Same step code as `value`; what it adds is the epoch's end (one readback of the epoch
    #      accumulators, the History arrays).
    public = None
    if world == 1 and not dp and not per_sample and not args.no_public_path:
        nb_pub = 64
        rows = []
        for i in range(nb_pub):
            rows.append(resident[i % len(resident)])
        ds = ([torch.cat([r[0][k] for r in rows], 0) for k in range(n_enc)], torch.cat([r[1] for r in rows], 0))
        loader = mm.DeviceResidentLoader(ds, batch_size=B, device=dev)
        hist = mm.MultiModNHistory([f"t{d}" for d in range(wl["D"])])
        for _ in range(3):                                   # eager sighting, capture, first replay
            model.train_epoch(loader, opt, crit, hist)
        torch.cuda.synchronize()
        n_ep = 4
        t_p = time.perf_counter()
        for _ in range(n_ep):
            model.train_epoch(loader, opt, crit, hist)
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t_p
        public = {"entry_point": "MultiModN.train_epoch(DeviceResidentLoader, multimodn_amd.optim.Adam, CrossEntropyLoss, History)",
                  "value": B * nb_pub * n_ep / el_p, "unit": "samples/s", "us_per_step": el_p / (nb_pub * n_ep) * 1e6,
                  "epochs": n_ep, "steps_per_epoch": nb_pub, "ratio_to_value": (B * nb_pub * n_ep / el_p) / value}
        del loader, ds

    # ---- the other two ways a pipeline can drive the same step, measured, never `value`:
    #  h2d_path   host batches (pinned memory) through the package's staging ring: one packed H2D copy per step, the
    #             reference's `data.to(device)` (multimodn/multimodn.py:132-135) - the PCIe-inclusive rate;
    #  stock_path the import-swap-only pipeline: torch.utils.data.DataLoader over a PartitionDataset (per-sample
    #             __getitem__ + default collate, as pipelines/titanic/titanic_mlp_pipeline.py:57-60 builds it) and a stock
    #             torch.optim.Adam (host NaN policy, .grad views, optimizer.step() as its own launches)
    other_paths = None
    if world == 1 and not dp and not per_sample and not args.no_public_path and args.optimizer == "hip":
        other_paths = {}
        try:
            nb = 32
            host_batches = [([x.cpu().pin_memory() for x in resident[i % len(resident)][0]], resident[i % len(resident)][1].cpu().pin_memory())
                            for i in range(min(nb, len(resident)))]
            hb = [host_batches[i % len(host_batches)] for i in range(nb)]
            for _ in range(2):
                model._train_steps(_Sized(hb), opt)
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            n_rep = 3
            rep_us = []
            for _ in range(n_rep):
                t_r = time.perf_counter()
                model._train_steps(_Sized(hb), opt)
                torch.cuda.synchronize()
                rep_us.append((time.perf_counter() - t_r) / nb * 1e6)
            el_h = time.perf_counter() - t_h
            bytes_step = sum(int(x.numel()) * 4 for x in hb[0][0]) + int(hb[0][1].numel()) * 8
            # (the MEDIAN of the three 32-step calls: on this host one call in three or four takes 13 - 16 ms longer - 500 us per
            #  step for that call - whatever the layout of the copies; every call's figure rides along)
            med = float(np.median(rep_us))
            other_paths["h2d_path"] = {
                "entry_point": "MultiModN._train_steps over pinned HOST batches (staging ring: one packed copy per step, on its own copy stream), multimodn_amd.optim.Adam",
                "us_per_step": med, "value": B / (med * 1e-6), "unit": "samples/s",
                "h2d_bytes_per_step": bytes_step, "h2d_gbs": bytes_step / (med * 1e-6) / 1e9, "steps": nb * n_rep,
                "us_per_step_by_call": rep_us, "us_per_step_mean_of_calls": el_h / (nb * n_rep) * 1e6, "host_threads": torch.get_num_threads()}
        except Exception as ex:
            other_paths["h2d_path"] = {"error": repr(ex)[:300]}
        try:
            from torch.utils.data import DataLoader
            n_rows_s = B * 8
            Xs = np.concatenate([np.concatenate(h[0], axis=1) for h in host[:8]], 0)[:n_rows_s]
            ys = np.concatenate([h[1] for h in host[:8]], 0)[:n_rows_s]
            dset = mm.PartitionDataset(Xs, ys, list(wl["F"]))
            stock_loader = DataLoader(dset, batch_size=B, shuffle=False)
            model2 = build_model(mm, wl, dev)
            opt2 = torch.optim.Adam(model2.parameters(), wl["lr"])
            hist2 = mm.MultiModNHistory([f"t{d}" for d in range(wl["D"])])
            model2.train_epoch(stock_loader, opt2, crit, hist2)
            torch.cuda.synchronize()
            nb_s = len(stock_loader)
            ep_us = []                                        # (three epochs, the median: same host hiccups as the h2d path)
            for _ in range(3):
                t_s = time.perf_counter()
                model2.train_epoch(stock_loader, opt2, crit, hist2)
                torch.cuda.synchronize()
                ep_us.append((time.perf_counter() - t_s) / nb_s * 1e6)
            el_s = float(np.median(ep_us)) * nb_s * 1e-6
            # where the time goes: the loader alone (PartitionDataset.__getitems__: one gather per partition and batch), and -
            # for reference - the same loader over a dataset that only has the reference's per-sample __getitem__
            t_l = time.perf_counter()
            for _b in stock_loader:
                pass
            el_l = time.perf_counter() - t_l

            class _PerSample(torch.utils.data.Dataset):
                def __init__(self, d): self.d = d
                def __len__(self): return len(self.d)
                def __getitem__(self, i): return self.d[i]
            t_p = time.perf_counter()
            for _b in DataLoader(_PerSample(dset), batch_size=B, shuffle=False):
                pass
            el_p = time.perf_counter() - t_p
            other_paths["stock_path"] = {
                "entry_point": "MultiModN.train_epoch(torch DataLoader(PartitionDataset), torch.optim.Adam, CrossEntropyLoss, History): the reference pipeline with the import swapped",
                "us_per_step": el_s / nb_s * 1e6, "value": B * nb_s / el_s, "unit": "samples/s", "steps": nb_s,
                "us_per_step_by_epoch": ep_us,
                "loader_alone_us_per_step": el_l / nb_s * 1e6,
                "per_sample_loader_alone_us_per_step": el_p / nb_s * 1e6,
                "note": "host-bound; round 6: torch's intra-op pool is capped once per call instead of around every batch (two pool resizes per step cost ~0.3 ms of it); "
                        "round 5: PartitionDataset answers the loader's batched fetch (__getitems__), the "
                        "reference's per-sample __getitem__ + collate of 4096 rows is per_sample_loader_alone_us_per_step; "
                        "DeviceResidentLoader is the resident form"}
            del model2, opt2, stock_loader, dset
        except Exception as ex:
            other_paths["stock_path"] = {"error": repr(ex)[:300]}

    # ---- per-kernel durations with HIP events on the launch stream.  Each kernel of the step is
    # launched REP times back to back between one event pair (same stream the step uses), so the
    # host-side launch cost (~3 us per eager launch) does not pollute a ~30 us kernel; the
    # rocprofv3 --kernel-trace averages of the same command are committed under profiles/.
    lib, plan, C = eng.lib, eng._plan, __import__("ctypes")
    stream = torch.cuda.current_stream().cuda_stream
    pairs = [(i, i) for i in range(n_enc)]
    eng.begin_sequence()
    if per_sample:
        batches, _keep_ps = [], []
        for xs, y, sq in resident:
            bb, kk = eng.per_sample_batch(xs, y, sq)
            bb.batch_global = B * world
            batches.append(bb); _keep_ps.append(kk)
    else:
        batches = [eng.make_batch(xs, y, pairs, batch_global=B * world, device_nan_flags=True) for xs, y in resident]
    plan = eng._plan                                          # (per-sample batches may have re-planned)
    b0 = batches[0]
    fwd_name = lib.mmn_chain_kernel_name(plan, C.byref(b0), 0).decode()
    bwd_name = lib.mmn_chain_kernel_name(plan, C.byref(b0), 1).decode()
    fused_name = lib.mmn_chain_kernel_name(plan, C.byref(b0), 2).decode()
    dec_name = lib.mmn_chain_kernel_name(plan, C.byref(b0), 3).decode()      # generic tier, split form: the decoders' own launch
    # k_prepare (NaN scan + repack of the weights) is NOT part of a steady-state step any more: the scan of batch t+1
    # rides in step t's k_reduce, the repack is replaced by the Adam tail's scatter.  Timed here as what the first step
    # of an epoch still pays.
    kern = {"k_prepare(first step of an epoch only)": lambda b: lib.mmn_prepare(plan, C.byref(b), 1, stream)}
    if fused_name:        # forward + backward chain in one launch (what mmn_train_step uses)
        kern[fused_name] = lambda b: lib.mmn_chain_fwd_bwd(plan, C.byref(b), alpha, beta, stream)
    else:
        kern[fwd_name] = lambda b: lib.mmn_chain_fwd(plan, C.byref(b), alpha, beta, 1 | (2 if dec_name else 0), stream)
        if dec_name:
            kern[dec_name] = lambda b: lib.mmn_chain_fwd(plan, C.byref(b), alpha, beta, 1 | 4, stream)
        kern[bwd_name] = lambda b: lib.mmn_chain_bwd(plan, C.byref(b), beta, stream)
    kern["k_wgrad"] = lambda b: lib.mmn_wgrad(plan, C.byref(b), stream)
    fuse_opt = opt if (not dp and args.optimizer == "hip") else None
    if args.optimizer == "hip":
        adam_desc = opt.fused_descriptor(eng)
    # the step's second half exactly as mmn_train_step_ex launches it since round 5 (ABI 111): k_wgrad with the stats block
    # (+ Adam's coefficient block) in its launch, then k_reduce as gradient blocks (+ Adam + scatter); "k_reduce" below is the
    # difference to the k_wgrad launch alone
    so = mm.hip.StepOpts()
    so.accumulate_epoch = 0 if dp else 1
    if fuse_opt is not None and adam_desc is not None:
        so.adam = C.pointer(adam_desc)
    elif args.optimizer == "hip" and adam_desc is not None:
        kern["k_adam_accumulate"] = lambda b: lib.mmn_adam_step_accumulate(plan, C.byref(adam_desc), alpha, beta, stream)
    kern["k_wgrad+k_reduce"] = lambda b: lib.mmn_wgrad_reduce(plan, C.byref(b), alpha, beta, C.byref(so), stream)
    REP, ROUNDS = 20, 5
    avg_us = {}
    for name, fn in kern.items():
        times = []
        for rnd in range(ROUNDS):
            b = batches[rnd % len(batches)]
            if eng.dropout_encoders:
                _m = eng.draw_dropout_masks(b)
            eng.local_step(b, alpha, beta, accumulate=False)        # valid inputs for every kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                assert fn(b) == 0
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3 / REP)
        avg_us[name] = float(np.median(times))
    avg_us["k_reduce"] = max(avg_us["k_wgrad+k_reduce"] - avg_us["k_wgrad"], 0.0)     # (includes the boundary between the two launches)
    flp = flops_per_sample_of(wl)
    if fused_name:
        fl = {fused_name: flp["k_chain_fwd"] + flp["k_chain_bwd"], "k_wgrad": flp["k_wgrad"]}
    elif dec_name:     # the decoders' forward AND backward products are k_dec_fb's
        dd = flp["decoders_one_way"]
        fl = {fwd_name: flp["k_chain_fwd"] - dd, dec_name: 2 * dd, bwd_name: flp["k_chain_bwd"] - dd, "k_wgrad": flp["k_wgrad"]}
    else:
        fl = {fwd_name: flp["k_chain_fwd"], bwd_name: flp["k_chain_bwd"], "k_wgrad": flp["k_wgrad"]}
    dominant = max(fl, key=lambda k: avg_us[k])
    achieved = fl[dominant] * B / (avg_us[dominant] * 1e-6) / 1e12
    traffic, mfma_busy, traffic_error = None, None, None
    tag = None
    try:                                                      # HBM bytes per launch / matrix-pipe busy % from the committed PMC passes
        tag = PROFILE_TAGS.get(args.workload)
        if tag and B == wl["B"]:                             # the passes were made on this workload at this batch
            pmc = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_traffic.json")))
            traffic = pmc["kernels"].get(dominant, {}).get("hbm_bytes_per_launch")
            util = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_util.json")))
            mfma_busy = util["kernels"].get(dominant, {}).get("mfma_busy_pct")
            if traffic is None:
                traffic_error = f"profiles/{tag}_pmc_traffic.json has no entry for {dominant}"
    except Exception as ex:                                   # (a missing / renamed profile is REPORTED, not swallowed)
        traffic_error = f"{type(ex).__name__}: {ex}"[:300]
    roofline = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "mfma_busy_pct_pmc": mfma_busy, "traffic_error": traffic_error,
                "traffic_source": (f"profiles/{tag}_pmc_traffic.json, profiles/{tag}_pmc_util.json (builder-run: separate rocprofv3 "
                                   "--pmc passes of this command on another box, committed; NOT measured in this run)") if traffic else None,
                "traffic_note": "HBM bytes per launch of the dominant kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 and its "
                                "SQ_VALU_MFMA_BUSY_CYCLES share; achieved / frac / avg_launch_us ARE measured in this run (HIP events)",
                "avg_launch_us": avg_us, "flops_per_sample": fl,
                "algorithmic_flops_per_launch": fl[dominant] * B,
                "step_frac_of_fp32_roof": value / world * sum(fl.values()) / (FP32_MFMA_PEAK_TFLOPS * 1e12),
                "hbm_frac_of_peak": (traffic / (avg_us[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None}

    if bool(model.__dict__.get("_small_epochs")):
        # the timed call ran as ONE launch of k_epoch_small (Titanic-sized models): the roofline block describes THAT kernel -
        # one launch = every step of the call; per step it does the whole step's algorithmic work in ms_per_step (VERDICT r5:
        # the block used to name k_fb9 and the step path's launch times while `launch` said the call took k_epoch_small)
        step_us = ms_per_step * 1e3
        tot = float(sum(fl.values()))
        ach = tot * B / (step_us * 1e-6) / 1e12
        ek_traffic, ek_busy = None, None
        try:
            if tag and B == wl["B"]:
                ek_traffic = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_traffic.json")))["kernels"].get("k_epoch_small", {}).get("hbm_bytes_per_launch")
                ek_busy = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_util.json")))["kernels"].get("k_epoch_small", {}).get("mfma_busy_pct")
        except Exception as ex:
            roofline["traffic_error"] = f"{type(ex).__name__}: {ex}"[:300]
        roofline.update({"kernel": "k_epoch_small", "achieved": ach, "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": ek_traffic,
                         "mfma_busy_pct_pmc": ek_busy, "algorithmic_flops_per_launch": tot * B * args.steps,
                         "latency_bound": True,
                         "step_path_launch_us": avg_us,
                         "avg_launch_us": {"k_epoch_small": step_us * args.steps, "k_epoch_small_per_step": step_us},
                         "flops_per_sample": {"k_epoch_small": tot}})
    if args.optimizer != "hip":
        opt_text = "torch.optim.Adam(fused, capturable)"
    elif dp:
        opt_text = ("multimodn_amd.optim.Adam inside the one-shot exchange launch (k_adam_accumulate_oneshot)" if getattr(model, "_dp_oneshot", False)
                    else "multimodn_amd.optim.Adam in the launch behind the all-reduce (k_adam_accumulate)")
    else:
        opt_text = "multimodn_amd.optim.Adam fused into k_reduce (coefficients and step counters: the k_wgrad launch's side block)"
    epoch_kernel = bool(model.__dict__.get("_small_epochs"))   # the call ran as ONE launch (k_epoch_small: Titanic-sized models)
    if epoch_kernel:
        opt_text = "multimodn_amd.optim.Adam inside k_epoch_small (parameters and moments resident in LDS for the whole call)"
    out = {
        "metric": "samples/sec/GPU (MIMIC 4-enc/3-dec, state_dim=128) + CPU-match Δloss",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["text"] if B == wl["B"] else wl["text"] + f" [rows per GPU: {B}]",
                   "optimizer": opt_text,
                   "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                   "step_path": "MultiModN._train_steps (the batch loop of MultiModN.train_epoch)",
                   "launch": (f"hipGraph replay ({max(group, int(getattr(model, 'REPLAY_GROUP', 8))) if (dp or per_sample) else group} steps per graph"
                              + (f", {int(model.REPLAY_GROUP_NEXT)} behind the first group of a call" if (not per_sample and int(getattr(model, 'REPLAY_GROUP_NEXT', 0)) not in (0, group)) else "")
                              + ")") if replayed else ("one launch per call: k_epoch_small, all steps in one workgroup (roofline describes that launch; "
                                                       "roofline.step_path_launch_us: the step-by-step path's kernels, timed through the C ABI)" if epoch_kernel else "eager"),
                   # counted around every torch.distributed collective of the timed region (n steps + 1 for the sequence's
                   # first batch, whose NaN flags have no predecessor to ride with)
                   "collectives_per_step": (n_coll["all_reduce"] + n_coll["other"]) / args.steps if dp else 0,
                   "collectives_counted": dict(n_coll, steps=args.steps) if dp else None,
                   "dist_world_size": (dist.get_world_size() if dp else 1),
                   "dist_backend": (dist.get_backend() if dp else None),
                   "samples_per_sec_per_gpu": value / world},
        "roofline": roofline,
        "step_us_hip_events": dist_us,
        "public_path": public,
        "other_paths": other_paths,
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        from oracle import multimodn_oracle as O      # the CPU oracle: imported for this leg ONLY, never measured as `value`
        spec = oracle_spec(O, wl)
        # `value` / `kind` of record: the stronger CPU yardstick, torch's own kernels under a vectorised driver (BASELINE.md
        # section 4 mode ii); the numpy restatement (the parity oracle itself) and the reference-style step ride along
        if not per_sample:
            out["cpu_baseline"] = cpu_torch_vectorised(mm, wl, B, min(args.cpu_budget, 10.0))
            out["cpu_baseline"]["torch_vectorised"] = True
            out["cpu_baseline"]["numpy_oracle"] = cpu_baseline(O, spec, B, min(args.cpu_budget, 8.0))
        else:
            out["cpu_baseline"] = cpu_baseline(O, spec, B, args.cpu_budget)
        out["cpu_baseline"]["reference_style_step"] = cpu_faithful_step(O, spec, B)
        if not per_sample:
            out["cpu_match"] = cpu_match(O, mm, wl, dev, B)
            if args.workload == "c3" and not args.no_curve:
                out["cpu_match"]["curve"] = cpu_match_curve(O, mm, wl, dev, B)
                out["cpu_match"]["ok"] = bool(out["cpu_match"]["ok"] and out["cpu_match"]["curve"]["ok"])
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not dp and args.workload == "c3" and not args.no_secondary:
        out["secondary"] = secondary_workloads()
        out["rows_sweep"] = rows_sweep()
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL prints a version banner through C stdio (block-buffered when stdout
        # is a pipe, i.e. it would otherwise surface at process exit, behind this line)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
        if out.get("cpu_match") is not None and not out["cpu_match"].get("ok", True):
            sys.stderr.write("bench.py: the HIP path's trained weights / loss grid left the parity bar (cpu_match.ok = false)\n")
            raise SystemExit(4)


if __name__ == "__main__":
    main()
