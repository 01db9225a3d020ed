#!/usr/bin/env python3
"""Generate golden vectors for the train_epoch hot path by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  Nothing here travels as reference
code: the script imports the reference modules, feeds them seeded inputs, and stores only
inputs + observed outputs as .npz data files next to this script.

The reference does not import as-is; three shims are installed in THIS
process only, none of which touches the arithmetic of the path:
  * `torchsummary`  (not installed)            -> stub module with a no-op `summary`
  * `torchmetrics`  (not installed)            -> stub whose binary ConfusionMatrix returns
        bincount(target*2+preds).reshape(2,2) = C[true][pred], the documented torchmetrics layout
        that multimodn.py:53-58 relies on
  * `torch._utils._accumulate` (removed)       -> itertools.accumulate
Per-step values are observed through the caller-injected criterion / optimizer objects and by
wrapping Tensor.backward (instrumentation of the harness, not of the reference).

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz and verifies
                                                       the oracle against every vector)
"""
import itertools
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def install_shims():
    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules["torchsummary"] = ts

    tm = types.ModuleType("torchmetrics")

    class ConfusionMatrix:
        def __init__(self, task="binary", num_classes=2, **kw):
            pass

        def to(self, device):
            return self

        def __call__(self, preds, target):
            idx = target.long() * 2 + preds.long()
            return torch.bincount(idx, minlength=4).reshape(2, 2)

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return torch.tensor(float("nan"))

    class _Inert3(_Inert):                         # the curve metrics return (x, y, thresholds)
        def __call__(self, *a, **k):
            nan = torch.tensor(float("nan"))
            return nan, nan, nan

    tm.ConfusionMatrix = ConfusionMatrix
    for n in ("F1Score", "Accuracy", "AUROC"):
        setattr(tm, n, type(n, (_Inert,), {}))
    for n in ("ROC", "PrecisionRecallCurve"):
        setattr(tm, n, type(n, (_Inert3,), {}))
    sys.modules["torchmetrics"] = tm

    import torch._utils
    torch._utils._accumulate = itertools.accumulate
    sys.path.insert(0, REF)


install_shims()
from multimodn.multimodn import MultiModN                      # noqa: E402  (reference)
from multimodn.encoders import MLPEncoder, MLPFeatureEncoder, MIMIC_MLPEncoder    # noqa: E402
from multimodn.decoders import LogisticDecoder, MLPDecoder     # noqa: E402
from multimodn.history import MultiModNHistory                 # noqa: E402
import datasets as _ref_datasets                               # noqa: E402
assert _ref_datasets.__file__.startswith(REF), "HuggingFace `datasets` shadowed the reference's"

sys.path.insert(0, REPO)
from oracle import multimodn_oracle as O                       # noqa: E402

ACTS = {"relu": torch.nn.functional.relu, "sigmoid": torch.sigmoid, "identity": (lambda x: x)}
ACT_ID = {"relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "identity": O.ACT_IDENTITY}


class RecordingCriterion:
    def __init__(self):
        self.inner = torch.nn.CrossEntropyLoss()
        self.calls = []

    def __call__(self, out, tgt):
        loss = self.inner(out, tgt)
        self.calls.append(float(loss.detach()))
        return loss


class OptimizerProxy:
    """Delegates to a real torch.optim.Adam; snapshots grads before and params after step()."""

    def __init__(self, model, lr):
        self.model = model
        self.opt = torch.optim.Adam(list(model.parameters()), lr)
        self.grads, self.params = [], []

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        self.grads.append({n: (None if p.grad is None else p.grad.detach().numpy().copy())
                           for n, p in self.model.named_parameters()})
        self.opt.step()
        self.params.append({n: p.detach().numpy().copy() for n, p in self.model.named_parameters()})


CONFIGS = {
    # name: dict(F, H, S, D, B, N, lr, pen, epochs, act, + optional nan / seq)
    "c1_titanic": dict(F=[6], H=(5, 5), S=32, D=1, B=32, N=100, lr=0.01, pen=(0.7, 0.3), epochs=3,
                       act="relu", store="all"),
    "c1_curve20": dict(F=[6], H=(5, 5), S=32, D=1, B=32, N=96, lr=0.01, pen=(0.7, 0.3), epochs=20,
                       act="relu", store="epochs"),
    "c2_split": dict(F=[3, 2], H=(5, 5), S=64, D=2, B=64, N=150, lr=0.01, pen=(0.7, 0.3), epochs=2,
                     act="relu", store="all"),
    "c3_small": dict(F=[64] * 4, H=(32, 32), S=128, D=3, B=64, N=128, lr=1e-3, pen=(1.0, 0.3),
                     epochs=1, act="relu", store="first_last"),
    "nan_skip": dict(F=[4, 3, 5], H=(8,), S=16, D=2, B=16, N=48, lr=0.01, pen=(1.0, 0.5), epochs=2,
                     act="relu", store="all", nan=[(1, 1, 3, 2)]),   # (batch, slot, row, col)
    "seq_perm": dict(F=[4, 3, 5, 2], H=(6, 7), S=24, D=2, B=16, N=32, lr=0.01, pen=(1.0, 0.5),
                     epochs=2, act="relu", store="all", seq=[2, 0, 1, 3]),
    "slp_sigmoid": dict(F=[5, 7], H=(), S=16, D=1, B=16, N=40, lr=0.01, pen=(0.7, 0.3), epochs=2,
                        act="sigmoid", store="all"),
    "mlp_sigmoid": dict(F=[5, 7], H=(9,), S=16, D=2, B=16, N=40, lr=0.01, pen=(0.7, 0.3), epochs=2,
                        act="sigmoid", store="all"),
    "mlp_identity": dict(F=[6], H=(4, 4, 4), S=8, D=3, B=8, N=24, lr=0.01, pen=(1.0, 1.0), epochs=2,
                         act="identity", store="all"),
    # ---- the feature-wise Titanic pipelines: one MLPFeatureEncoder(state 5, hidden 5) per feature over a FeatureWiseDataset,
    # one LogisticDecoder, lr 0.01, penalties (0.7, 0.3).  titanic_featurewise: pipelines/titanic/titanic_featurewise_pipeline.py
    # :26-73 (five features, batch 32); titanic_missingness: titanic_missingness_pipeline.py:26-74 (six features, batch size 1,
    # missing values kept as NaN: a sample's missing feature skips that encoder, multimodn.py:167-171)
    "titanic_featurewise": dict(F=[1] * 5, H=(5,), S=5, D=1, B=32, N=100, lr=0.01, pen=(0.7, 0.3), epochs=3, act="relu",
                                store="all", feature_encoders=True),
    "titanic_missingness": dict(F=[1] * 6, H=(5,), S=5, D=1, B=1, N=20, lr=0.01, pen=(0.7, 0.3), epochs=2, act="relu",
                                store="all", feature_encoders=True,
                                nan=[(1, 2, 0, 0), (4, 5, 0, 0), (4, 0, 0, 0), (7, 5, 0, 0), (11, 2, 0, 0), (11, 5, 0, 0), (12, 3, 0, 0),
                                     (16, 0, 0, 0), (16, 1, 0, 0), (16, 2, 0, 0), (16, 3, 0, 0), (16, 4, 0, 0), (16, 5, 0, 0)]),
    # ---- MIMIC family: MIMIC_MLPEncoder (mlp_encoder.py:9-47) + MLPDecoder
    # (decoders.py:22-46).  enc_kinds: per-encoder class ("mimic" | "mlp"); dec: per-decoder
    # ("mlp", hidden) | ("class", ()); dropout: MIMIC encoders' p (masks are recorded per step).
    "mimic_p0": dict(F=[6, 5], H=(8, 8), S=16, D=2, B=16, N=40, lr=0.01, pen=(1.0, 0.5), epochs=2,
                     act="relu", store="all", enc_kinds=["mimic", "mimic"], dropout=0.0,
                     dec=[("mlp", (8, 8)), ("mlp", (8, 8))]),
    "mimic_drop": dict(F=[7, 4, 5], H=(8,), S=24, D=2, B=16, N=40, lr=0.01, pen=(1.0, 0.5), epochs=2,
                       act="relu", store="all", enc_kinds=["mimic"] * 3, dropout=0.2,
                       dec=[("mlp", (6,)), ("mlp", (6,))]),
    "mimic_mixed": dict(F=[4, 3, 5], H=(6,), S=16, D=3, B=16, N=48, lr=0.01, pen=(0.7, 0.3), epochs=2,
                        act="sigmoid", store="all", enc_kinds=["mimic", "mlp", "mimic"], dropout=0.25,
                        dec=[("mlp", (5,)), ("class", ()), ("mlp", ())], nan=[(1, 0, 3, 1)]),
    "mimic_c3_small": dict(F=[64] * 4, H=(32, 32), S=128, D=3, B=64, N=128, lr=1e-3, pen=(1.0, 0.3),
                           epochs=1, act="relu", store="first_last", enc_kinds=["mimic"] * 4, dropout=0.2,
                           dec=[("mlp", (32, 32))] * 3),
    # ---- the reference's REAL MIMIC configuration: pipelines/mimic/mimic_multi_task_pipeline.py:53-83,118-119 -
    # state_size 50, encoder / decoder hidden (32, 32), dropout 0.2, batch 16, lr 1e-3, err_penalty 1, state_change_penalty 0,
    # two targets; source widths from datasets/mimic/mimic_dataset.py:21.  haim_pipeline: the four sources the pipeline
    # selects ('de', 'vd', 'n_ech', 'ts_ce'); haim_all9: all nine sources of the dataset (E = 9).
    "haim_pipeline": dict(F=[6, 1024, 768, 99], H=(32, 32), S=50, D=2, B=16, N=48, lr=1e-3, pen=(1.0, 0.0), epochs=2,
                          act="relu", store="first_last", enc_kinds=["mimic"] * 4, dropout=0.2, dec=[("mlp", (32, 32))] * 2),
    "haim_all9": dict(F=[6, 1024, 1024, 99, 242, 110, 768, 768, 768], H=(32, 32), S=50, D=2, B=16, N=16, lr=1e-3, pen=(1.0, 0.0),
                      epochs=2, act="relu", store="first_last", enc_kinds=["mimic"] * 9, dropout=0.2, dec=[("mlp", (32, 32))] * 2),
}


# ---- per-sample extension (BASELINE configs[4]): the reference defines per-sample sequences / NaN rows only at
# batch size 1 (multimodn.py:518-523, :167-171; pipelines/titanic/titanic_missingness_pipeline.py:35 runs exactly that).
# These runs feed the reference ONE sample per batch - its own sequence row, its own NaN rows - with an optimizer proxy that
# snapshots the gradients and does NOT step, so that every sample sees the same weights: the per-sample values and their
# means over the N samples are what oracle.per_sample_step and the HIP per-sample step must reproduce for the N-row batch.
PER_SAMPLE_B1 = {
    "per_sample_b1_mlp": dict(F=[6, 6, 6], H=(8, 8), S=16, D=2, N=32, pen=(1.0, 0.5), act="relu", p_missing=0.3),
    # an MLPEncoder shape OUTSIDE the fused kernel's tiled form (n_features 72 > 64, hidden 48 > 32): the generic tier's tiled
    # form runs it (round 6, VERDICT r5 #8: per-sample mode for every MLPEncoder shape)
    "per_sample_b1_wide": dict(F=[72, 72, 72], H=(48,), S=24, D=2, N=32, pen=(1.0, 0.5), act="relu", p_missing=0.3),
    # the reference's feature-wise missingness pipeline (pipelines/titanic/titanic_missingness_pipeline.py:26-74: six
    # MLPFeatureEncoder(state 5, hidden 5), one feature each, batch size 1, NaN = the passenger's feature is missing, default
    # encoder order) - what a per-sample BATCH of such a model must reproduce (round 6: per-sample mode for 5 .. 8 encoders)
    "per_sample_b1_feature6": dict(F=[1] * 6, H=(5,), S=5, D=1, N=48, pen=(0.7, 0.3), act="relu", p_missing=0.3,
                                   feature_encoders=True, default_order=True),
    "per_sample_b1_mimic": dict(F=[6, 6, 6], H=(8,), S=16, D=2, N=32, pen=(1.0, 0.5), act="relu", p_missing=0.3,
                                enc_kinds=["mimic"] * 3, dropout=0.2, dec=[("mlp", (6,)), ("mlp", (6,))]),
}


class SnapshotOnlyProxy:
    """zero_grad() is real, step() only snapshots the gradients: the weights never move."""

    def __init__(self, model):
        self.model = model
        self.grads = []

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None

    def step(self):
        self.grads.append({n: (None if p.grad is None else p.grad.detach().numpy().copy())
                           for n, p in self.model.named_parameters()})


def _reference_encoder(cfg, S, f, kind):
    if cfg.get("feature_encoders"):                           # mlp_encoder.py:81-94: MLPEncoder(S, 1, (hidden,)) by another constructor
        assert kind == "mlp" and f == 1 and len(cfg["H"]) == 1
        return MLPFeatureEncoder(S, cfg["H"][0], ACTS[cfg["act"]])
    if kind == "mlp":
        return MLPEncoder(S, f, tuple(cfg["H"]), ACTS[cfg["act"]])
    return MIMIC_MLPEncoder(S, f, tuple(cfg["H"]), dropout=cfg.get("dropout", 0.0), activation=ACTS[cfg["act"]])


def build_reference_model(cfg):
    S, D = cfg["S"], cfg["D"]
    kinds = cfg.get("enc_kinds", ["mlp"] * len(cfg["F"]))
    encoders = [_reference_encoder(cfg, S, f, k) for f, k in zip(cfg["F"], kinds)]
    decoders = [LogisticDecoder(S) if k == "class" else MLPDecoder(S, tuple(h), 2)
                for k, h in cfg.get("dec", [("class", ())] * D)]
    return MultiModN(S, encoders, decoders, cfg["pen"][0], cfg["pen"][1], device=torch.device("cpu")), encoders


def run_reference_b1(name, cfg, seed=0):
    """The reference's train_epoch over N batches of ONE sample each (per-row sequence, NaN rows), weights frozen."""
    torch.manual_seed(seed)
    torch.set_num_threads(1)
    E, D, N = len(cfg["F"]), cfg["D"], cfg["N"]
    model, encoders = build_reference_model(cfg)
    rng = np.random.default_rng(seed + 11)
    X = [rng.standard_normal((N, f)).astype(np.float32) for f in cfg["F"]]
    y = rng.integers(0, 2, size=(N, D)).astype(np.int64)
    # missing not at random: a modality is missing more often when y_0 = 1; a missing modality = a NaN row
    pm = np.where(y[:, :1] == 1, 1.5 * cfg["p_missing"], 0.5 * cfg["p_missing"])
    miss = rng.random((N, E)) < pm
    miss[0, :] = False                       # sample 0 has everything, sample 1 nothing at all
    miss[1, :] = True
    for k in range(E):
        X[k][miss[:, k]] = np.nan
    seq = np.stack([rng.permutation(E) for _ in range(N)]).astype(np.int64)
    if cfg.get("default_order"):                                 # (missing modalities only: every sample in the default order)
        seq = np.tile(np.arange(E, dtype=np.int64), (N, 1))
    masks = {}
    opt = SnapshotOnlyProxy(model)

    def pre_hook(mod, inp):
        mod._rng_before = torch.get_rng_state()

    def make_post(e):
        def post_hook(mod, inp, out):
            if not mod.training or mod.p == 0:
                return
            after = torch.get_rng_state()
            torch.set_rng_state(mod._rng_before)
            mask = torch.nn.functional.dropout(torch.ones_like(inp[0]), mod.p, True)
            assert torch.equal(torch.get_rng_state(), after)
            assert torch.equal(out, inp[0] * mask)
            masks[(len(opt.grads), e)] = mask.numpy().copy()
        return post_hook

    for e, enc in enumerate(encoders):
        if isinstance(enc, MIMIC_MLPEncoder):
            enc.layers[0].register_forward_pre_hook(pre_hook)
            enc.layers[0].register_forward_hook(make_post(e))
    init = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}
    loader = [([torch.from_numpy(X[k][b:b + 1]) for k in range(E)], torch.from_numpy(y[b:b + 1]),
               torch.from_numpy(seq[b:b + 1])) for b in range(N)]
    crit = RecordingCriterion()
    hist = MultiModNHistory([f"t{d}" for d in range(D)])
    losses = []
    orig_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        losses.append(float(self.detach()))
        return orig_backward(self, *a, **k)

    torch.Tensor.backward = rec_backward
    try:
        model.train_epoch(loader, opt, crit, hist)
    finally:
        torch.Tensor.backward = orig_backward
    for n, p in model.named_parameters():
        assert np.array_equal(p.detach().numpy(), init[n]), "the weights must not move"
    # per-sample loss cells from the criterion's call order: row 0's decoders, then every EXECUTED encoder in the sample's
    # own order (multimodn.py:141-157,159-191); a skipped encoder leaves its row at 0
    cells = np.zeros((N, E + 1, D))
    executed = np.zeros((N, E), bool)
    ci = 0
    for b in range(N):
        rows = [0]
        for k, e in enumerate(seq[b]):                      # data slot k feeds encoder seq[b][k] (multimodn.py:162-163)
            if not miss[b, k]:
                rows.append(int(e) + 1)
                executed[b, e] = True
        for r in rows:
            for d in range(D):
                cells[b, r, d] = crit.calls[ci]
                ci += 1
    assert ci == len(crit.calls) and len(opt.grads) == N and len(losses) == N
    return dict(init=init, X=X, y=y, seq=seq, miss=miss, masks=masks, cells=cells, executed=executed, losses=losses,
                grads=opt.grads, hist=hist, model=model)


def write_per_sample_b1(name, cfg):
    ref = run_reference_b1(name, cfg)
    spec = spec_of(cfg)
    E, D, N = spec.E, spec.D, cfg["N"]
    h = ref["hist"]
    out = {"config_json": np.array(json.dumps(cfg)), "torch_version": np.array(torch.__version__),
           "param_names": np.array(spec.param_names()), "y": ref["y"], "seq": ref["seq"],
           "sample_loss": np.array(ref["losses"], np.float64), "sample_cells": ref["cells"], "executed": ref["executed"],
           "hist/loss": np.asarray(h.loss["train"][0]), "hist/state_change": np.asarray(h.state_change_loss[0]),
           "hist/accuracy": np.asarray(h.accuracy["train"][0]), "hist/sensitivity": np.asarray(h.sensitivity["train"][0]),
           "hist/specificity": np.asarray(h.specificity["train"][0]),
           "hist/balanced_accuracy": np.asarray(h.balanced_accuracy["train"][0])}
    for n, v in ref["init"].items():
        out[f"init/{n}"] = v
    for k, x in enumerate(ref["X"]):
        out[f"x{k}"] = x
    # the batch's dropout multipliers, row b = what the reference drew for sample b (ones where the encoder did not run)
    drop = None
    if ref["masks"]:
        drop = {e: np.ones((N, spec.encoders[e].n_features + spec.state_size), np.float32) for e in range(E)
                if spec.encoders[e].kind == "mimic"}
        for (b, e), mk in ref["masks"].items():
            drop[e][b] = mk[0]
        for e, m in drop.items():
            out[f"mask{e}"] = m
    # the mean over the samples of the reference's gradients (None = the encoder did not run for that sample = 0)
    mean_g = {}
    for n in spec.param_names():
        gs = [g[n] for g in ref["grads"] if g[n] is not None]
        mean_g[n] = (np.sum(np.stack(gs).astype(np.float64), 0) / N) if gs else None
        if mean_g[n] is not None:
            out[f"mean_grad/{n}"] = mean_g[n]
    out["grad_none"] = np.array([n for n, g in mean_g.items() if g is None])
    for b in (0, 1, 2):                                      # three samples' own gradients, for the record
        for n, g in ref["grads"][b].items():
            if g is not None:
                out[f"sample{b}/grad/{n}"] = g
    # ---- the oracle's per-sample step on the N-row batch against the reference's N batch-size-1 runs
    r = O.per_sample_step(ref["init"], spec, ref["X"], ref["y"], ref["seq"], drop_masks=drop)
    w = dict(cells=rel(r.err_loss, ref["cells"].sum(0) / N), hist_loss=rel(r.err_loss, out["hist/loss"]),
             sc=rel(r.state_change, out["hist/state_change"]), loss=abs(r.loss - np.mean(ref["losses"])) / abs(np.mean(ref["losses"])),
             grad=max(rel(r.grads[n].reshape(g.shape), g) for n, g in mean_g.items() if g is not None))
    assert np.array_equal(r.row_counts[1:], ref["executed"].sum(0)), name
    for n, g in mean_g.items():
        assert (g is None) == (r.grads[n] is None), (name, n)
    er = O.aggregate_epoch(E, D, [r], [N])                  # the N-row batch as one step = the reference's N one-sample steps
    assert np.array_equal(er.accuracy, out["hist/accuracy"]), name
    assert np.array_equal(er.sensitivity, out["hist/sensitivity"]) and np.array_equal(er.specificity, out["hist/specificity"]), name
    assert np.array_equal(er.balanced_accuracy, out["hist/balanced_accuracy"]), name
    print(f"{name:20s} samples={N} oracle.per_sample_step vs reference at batch size 1: " + " ".join(f"{k}={v:.2e}" for k, v in w.items()))
    assert w["cells"] < 2e-6 and w["hist_loss"] < 2e-6 and w["sc"] < 2e-6 and w["loss"] < 2e-6 and w["grad"] < 1e-5, w
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


def make_data(cfg, seed):
    rng = np.random.default_rng(seed)
    D, N = cfg["D"], cfg["N"]
    # data slot k feeds encoders[seq[k]] (multimodn.py:162-163), so slot k has THAT encoder's width
    Fs = [cfg["F"][e] for e in cfg["seq"]] if "seq" in cfg else cfg["F"]
    X = rng.standard_normal((N, sum(Fs))).astype(np.float32)
    w = rng.standard_normal((sum(Fs), D)).astype(np.float32)
    y = ((X @ w + 0.5 * rng.standard_normal((N, D)).astype(np.float32)) > 0).astype(np.int64)
    offs = np.cumsum([0] + Fs)
    batches = []
    for bi, s in enumerate(range(0, N, cfg["B"])):
        xs = [X[s:s + cfg["B"], offs[k]:offs[k + 1]].copy() for k in range(len(Fs))]
        for (b, slot, r, c) in cfg.get("nan", []):
            if b == bi:
                xs[slot][r, c] = np.nan
        yb = y[s:s + cfg["B"]].copy()
        if "seq" in cfg:
            batches.append((xs, yb, np.tile(np.array(cfg["seq"], np.int64), (len(yb), 1))))
        else:
            batches.append((xs, yb))
    return batches


def run_reference(name, cfg, seed=0):
    torch.manual_seed(seed)
    torch.set_num_threads(1)
    S, D = cfg["S"], cfg["D"]
    kinds = cfg.get("enc_kinds", ["mlp"] * len(cfg["F"]))
    encoders = [_reference_encoder(cfg, S, f, k) for f, k in zip(cfg["F"], kinds)]
    decoders = [LogisticDecoder(S) if k == "class" else MLPDecoder(S, tuple(h), 2)
                for k, h in cfg.get("dec", [("class", ())] * D)]
    model = MultiModN(S, encoders, decoders, cfg["pen"][0], cfg["pen"][1], device=torch.device("cpu"))
    # Dropout masks of the MIMIC encoders, observed without touching the reference: a forward hook
    # rewinds the CPU generator to where the Dropout module found it and draws the same noise again on
    # a tensor of ones (ATen: noise.bernoulli_(1-p).div_(1-p); out = input * noise), then checks that
    # the generator ends where the module left it and that out == input * mask bit for bit.
    masks = {}

    def pre_hook(mod, inp):
        mod._rng_before = torch.get_rng_state()

    def make_post(e):
        def post_hook(mod, inp, out):
            if not mod.training or mod.p == 0:
                return
            after = torch.get_rng_state()
            torch.set_rng_state(mod._rng_before)
            mask = torch.nn.functional.dropout(torch.ones_like(inp[0]), mod.p, True)
            assert torch.equal(torch.get_rng_state(), after)
            assert torch.equal(out, inp[0] * mask)
            masks[(len(opt.grads), e)] = mask.numpy().copy()
        return post_hook

    for e, enc in enumerate(encoders):
        if isinstance(enc, MIMIC_MLPEncoder):
            enc.layers[0].register_forward_pre_hook(pre_hook)
            enc.layers[0].register_forward_hook(make_post(e))
    init = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}
    batches_np = make_data(cfg, seed + 1)
    loader = [tuple([[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])] +
                    ([torch.from_numpy(b[2])] if len(b) > 2 else [])) for b in batches_np]
    crit = RecordingCriterion()
    opt = OptimizerProxy(model, cfg["lr"])
    hist = MultiModNHistory([f"t{d}" for d in range(D)])
    losses = []
    orig_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        losses.append(float(self.detach()))
        return orig_backward(self, *a, **k)

    torch.Tensor.backward = rec_backward
    try:
        for _ in range(cfg["epochs"]):
            model.train_epoch(loader, opt, crit, hist)
    finally:
        torch.Tensor.backward = orig_backward
    # forward-only entry points on the TRAINED model (multimodn.py:255-492): History arrays of
    # test(), predict() on the first batch, get_states() over the loader.  (test()'s per-decoder
    # report goes through torchmetrics, which is stubbed here: not recorded.)
    ev = {}
    test_hist = MultiModNHistory([f"t{d}" for d in range(D)])
    last_skipped = any(bool(torch.isnan(b[0][-1]).any()) for b in loader) and "seq" not in cfg
    try:
        model.test(loader, torch.nn.CrossEntropyLoss(), test_hist, tag="test")
        assert not last_skipped
    except RuntimeError:
        # multimodn.py:354-357 collects the last encoder's outputs only from batches where it RAN, :410-418 sets them against
        # the targets of ALL batches: once a batch's last feature is missing the report's confusion matrix gets two lengths
        # and raises - after the History arrays were appended (:389-409), which is what this fixture keeps
        assert last_skipped and len(test_hist.loss["test"]) == 1, name
        ev["test_report_raised"] = np.array(1)
    for k in ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        ev[f"test_{k}"] = np.asarray(getattr(test_hist, k)["test"][0])
    b0 = loader[0]
    if not any(bool(torch.isnan(t).any()) for t in b0[0]):
        seq0 = b0[2] if len(b0) > 2 else None      # (a Tensor: get_encoder_iterable calls .numpy(), multimodn.py:518)
        ev["predict"] = model.predict(b0[0], seq0)
    ev["states"] = torch.stack(model.get_states(loader)).numpy()
    return dict(init=init, batches=batches_np, crit_calls=crit.calls, grads=opt.grads,
                params=opt.params, losses=losses, hist=hist, eval=ev, masks=masks, model=model)


def spec_of(cfg):
    kinds = cfg.get("enc_kinds", ["mlp"] * len(cfg["F"]))
    encs = [O.EncoderSpec(f, tuple(cfg["H"]), ACT_ID[cfg["act"]], kind=k,
                          dropout=cfg.get("dropout", 0.0) if k == "mimic" else 0.0) for f, k in zip(cfg["F"], kinds)]
    decs = [O.DecoderSpec(k, tuple(h)) for k, h in cfg["dec"]] if "dec" in cfg else None
    return O.ModelSpec(cfg["S"], encs, cfg["D"], cfg["pen"][0], cfg["pen"][1], decoders=decs)


def checkpoint_round_trip(ref, cfg, spec):
    """SURVEY 8f #4.  (a) The checkpoint the MIMIC pipelines write (mimic_multi_task_pipeline.py:150-154:
    torch.save({'epoch', 'model_state_dict', 'auc_bac_val_cum'})) from the reference's TRAINED model becomes a
    fixture (tests/golden/ref_checkpoint_mimic_drop.pt: tensors only, no code); tests load it into the build.
    (b) The other direction, checked here because only this container has the reference: a multimodn_amd model
    (CPU, no engine needed for state_dict) saves the same kind of checkpoint, the reference's model loads it with
    load_state_dict(strict=True) and its eval-mode forward reproduces the oracle on those weights."""
    import io
    import multimodn_amd as mm
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import build_torch_model
    model = ref["model"]
    torch.save({"epoch": cfg["epochs"], "model_state_dict": model.state_dict(), "auc_bac_val_cum": 1.25},
               os.path.join(HERE, "ref_checkpoint_mimic_drop.pt"))
    params = O.init_params(spec, 123)
    ours = build_torch_model(spec, params, "cpu", mm)
    buf = io.BytesIO()
    torch.save({"epoch": 1, "model_state_dict": ours.state_dict()}, buf)
    buf.seek(0)
    model.load_state_dict(torch.load(buf)["model_state_dict"], strict=True)
    model.eval()
    b0 = ref["batches"][0]
    with torch.no_grad():
        state = model.init_state(len(b0[1]))
        for e, enc in enumerate(model.encoders):
            state = enc(state, torch.from_numpy(b0[0][e]))
        out = torch.stack([dec(state) for dec in model.decoders], 1).numpy()
    r = O.forward_backward(params, spec, b0[0], b0[1], want_grads=False, keep_states=True)
    want = O.decoder_outputs(params, spec, r.states[spec.E])
    assert rel(out, want) < 2e-6, rel(out, want)
    print("checkpoint round trip: build -> reference ok (rel err %.1e); reference -> fixture written" % rel(out, want))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-12)) if a.size else 0.0


def main():
    worst = {}
    only = set(sys.argv[1:])                                  # optional: regenerate just these configurations
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        ref = run_reference(name, cfg)
        spec = spec_of(cfg)
        n_steps = len(ref["losses"])
        nb = len(ref["batches"])
        out = {"config_json": np.array(json.dumps({k: v for k, v in cfg.items()})),
               "torch_version": np.array(torch.__version__), "numpy_version": np.array(np.__version__),
               "param_names": np.array(spec.param_names()),
               "step_loss": np.array(ref["losses"], np.float64)}
        for n, v in ref["init"].items():
            out[f"init/{n}"] = v
        for bi, b in enumerate(ref["batches"]):
            for k, x in enumerate(b[0]):
                out[f"batch{bi}/x{k}"] = x
            out[f"batch{bi}/y"] = b[1]
            if len(b) > 2:
                out[f"batch{bi}/seq"] = b[2]
        h = ref["hist"]
        out["hist/state_change"] = np.stack(h.state_change_loss)
        out["hist/loss"] = np.stack(h.loss["train"])
        out["hist/accuracy"] = np.stack(h.accuracy["train"])
        out["hist/sensitivity"] = np.stack(h.sensitivity["train"])
        out["hist/specificity"] = np.stack(h.specificity["train"])
        out["hist/balanced_accuracy"] = np.stack(h.balanced_accuracy["train"])
        assert h.loss["train"][0].dtype == np.float64 and h.sensitivity["train"][0].dtype == np.float32

        store = cfg["store"]
        steps_to_store = {"all": range(n_steps), "first_last": [0], "epochs": []}[store]
        for s in steps_to_store:
            for n, g in ref["grads"][s].items():
                if g is not None:
                    out[f"step{s}/grad/{n}"] = g
            out[f"step{s}/grad_none"] = np.array([n for n, g in ref["grads"][s].items() if g is None])
            if store == "all":
                for n, p in ref["params"][s].items():
                    out[f"step{s}/param/{n}"] = p
        for (ms, e), mk in ref["masks"].items():          # every step's masks: the run cannot be replayed without them
            out[f"step{ms}/mask{e}"] = mk
        for n, p in ref["params"][-1].items():
            out[f"final/{n}"] = p
        for k, v in ref["eval"].items():
            out[f"eval/{k}"] = v

        # ---- verify the oracle (fp32 and fp64) against what the reference just produced
        params = {n: v.copy() for n, v in ref["init"].items()}
        opt = O.Adam(cfg["lr"])
        w = dict(loss=0.0, grad=0.0, param=0.0, hist=0.0, crit=0.0, eval=0.0)
        ci = 0
        ep_results = []
        for ep in range(cfg["epochs"]):
            results, sizes = [], []
            for bi, b in enumerate(ref["batches"]):
                s = ep * nb + bi
                xs, y, seq = (list(b) + [None])[:3]
                r = O.forward_backward(params, spec, xs, y, seq,
                                       drop_masks={e: mk for (ms, e), mk in ref["masks"].items() if ms == s})
                w["loss"] = max(w["loss"], abs(r.loss - ref["losses"][s]) / abs(ref["losses"][s]))
                # criterion call order: row 0 decoders, then each executed encoder in sequence order
                rows = [0] + [e + 1 for _, e in O.encoder_iterable(spec.E, seq) if r.executed[e]]
                for row in rows:
                    for d in range(spec.D):
                        w["crit"] = max(w["crit"], abs(float(r.err_loss[row, d]) - ref["crit_calls"][ci])
                                        / abs(ref["crit_calls"][ci]))
                        ci += 1
                for n, g in ref["grads"][s].items():
                    assert (g is None) == (r.grads[n] is None), (name, s, n)
                    if g is not None:
                        w["grad"] = max(w["grad"], rel(r.grads[n].reshape(g.shape), g))
                opt.step(params, r.grads)
                for n, p in ref["params"][s].items():
                    w["param"] = max(w["param"], rel(params[n], p))
                results.append(r); sizes.append(len(y))
            er = O.aggregate_epoch(spec.E, spec.D, results, sizes)
            ep_results.append(er)
            w["hist"] = max(w["hist"], rel(er.loss, h.loss["train"][ep]), rel(er.state_change, h.state_change_loss[ep]))
            assert np.array_equal(er.accuracy, h.accuracy["train"][ep]), name
            assert np.array_equal(er.sensitivity, h.sensitivity["train"][ep]), name
            assert np.array_equal(er.specificity, h.specificity["train"][ep]), name
            assert np.array_equal(er.balanced_accuracy, h.balanced_accuracy["train"][ep]), name
        assert ci == len(ref["crit_calls"])
        # ---- forward-only entry points: oracle on the reference's TRAINED weights vs the reference
        fin = {n: p.copy() for n, p in ref["params"][-1].items()}
        ev = ref["eval"]
        tep, _ = O.test_epoch(fin, spec, ref["batches"])
        w["eval"] = max(rel(tep.loss, ev["test_loss"]), rel(O.get_states(fin, spec, ref["batches"]), ev["states"]))
        assert np.array_equal(tep.accuracy, ev["test_accuracy"]), name
        assert np.array_equal(tep.sensitivity, ev["test_sensitivity"]) and np.array_equal(tep.specificity, ev["test_specificity"]), name
        if "predict" in ev:
            b0 = ref["batches"][0]
            assert np.array_equal(O.predict(fin, spec, b0[0], b0[2] if len(b0) > 2 else None), ev["predict"]), name
        if name == "mimic_drop":
            checkpoint_round_trip(ref, cfg, spec)
        worst[name] = w
        print(f"{name:14s} steps={n_steps:3d} oracle-vs-reference rel err: " +
              " ".join(f"{k}={v:.2e}" for k, v in w.items()))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    for name, cfg in PER_SAMPLE_B1.items():
        if only and name not in only:
            continue
        write_per_sample_b1(name, cfg)
    bad = {n: w for n, w in worst.items() if w["loss"] > 2e-6 or w["crit"] > 2e-6 or w["eval"] > 2e-6}
    assert not bad, bad
    # gradients and trained weights (the printed `grad` / `param` columns): gated by the very rule the tests apply to the
    # fixtures just written - tests/test_oracle_golden.py, i.e. 1e-5 / 2e-5 outright or within fp32 noise of the float64
    # replay (helpers.assert_within_fp32_noise) - so that the generator cannot leave behind a file the suite would reject
    sys.path.insert(0, os.path.dirname(HERE))
    import test_oracle_golden as T
    for name in worst:
        T.test_oracle_reproduces_reference_run(name)
    for name in PER_SAMPLE_B1:
        if not only or name in only:
            T.test_per_sample_oracle_reproduces_the_reference_at_batch_size_one(name)
    print(f"oracle gate (tests/test_oracle_golden.py rules) passed for {len(worst)} + {len(PER_SAMPLE_B1)} fixtures")


if __name__ == "__main__":
    main()
