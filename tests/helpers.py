"""Shared test helpers: golden-vector loading, oracle <-> torch model conversion."""
import json
import os

import numpy as np

from oracle import multimodn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_ID = {"relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "identity": O.ACT_IDENTITY}
GOLDEN_NAMES = ["c1_titanic", "c1_curve20", "c2_split", "c3_small", "nan_skip", "seq_perm",
                "slp_sigmoid", "mlp_sigmoid", "mlp_identity",
                # the feature-wise Titanic pipelines (MLPFeatureEncoder per feature; batch 32 / batch size 1 with missing values)
                "titanic_featurewise", "titanic_missingness"]
# MIMIC family: MIMIC_MLPEncoder + MLPDecoder runs of the reference, dropout masks recorded
MIMIC_GOLDEN_NAMES = ["mimic_p0", "mimic_drop", "mimic_mixed", "mimic_c3_small"]
# the reference's real MIMIC configuration (pipelines/mimic/mimic_multi_task_pipeline.py:53-83,118-119; datasets/mimic/
# mimic_dataset.py:21): state 50, hidden (32, 32), dropout 0.2, batch 16; the pipeline's four sources / all nine
HAIM_GOLDEN_NAMES = ["haim_pipeline", "haim_all9"]


def spec_from_cfg(c):
    kinds = c.get("enc_kinds", ["mlp"] * len(c["F"]))
    encs = [O.EncoderSpec(f, tuple(c["H"]), ACT_ID[c["act"]], kind=k,
                          dropout=c.get("dropout", 0.0) if k == "mimic" else 0.0) for f, k in zip(c["F"], kinds)]
    decs = [O.DecoderSpec(k, tuple(h)) for k, h in c["dec"]] if "dec" in c else None
    return O.ModelSpec(c["S"], encs, c["D"], c["pen"][0], c["pen"][1], decoders=decs)


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = json.loads(str(self.z["config_json"]))
        c = self.cfg
        self.spec = spec_from_cfg(c)
        self.n_batches = len([k for k in self.z.files if k.endswith("/y")])
        self.epochs = c["epochs"]

    def init_params(self):
        return {n: self.z["init/" + n].copy() for n in self.spec.param_names()}

    def final_params(self):
        return {n: self.z["final/" + n] for n in self.spec.param_names()}

    def batch(self, bi):
        n_slots = len([k for k in self.z.files if k.startswith(f"batch{bi}/x")])
        xs = [self.z[f"batch{bi}/x{k}"] for k in range(n_slots)]
        y = self.z[f"batch{bi}/y"]
        key = f"batch{bi}/seq"
        return (xs, y, self.z[key]) if key in self.z.files else (xs, y)

    def batches(self):
        return [self.batch(i) for i in range(self.n_batches)]

    def step_grads(self, s):
        pre = f"step{s}/grad/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}

    def step_masks(self, s):
        """Dropout multipliers the reference drew in step s: {encoder id: [B, F_e + S]} (MIMIC family)."""
        pre = f"step{s}/mask"
        return {int(k[len(pre):]): self.z[k] for k in self.z.files if k.startswith(pre)}

    def has_step(self, s):
        return f"step{s}/grad_none" in self.z.files


PER_SAMPLE_B1_NAMES = ["per_sample_b1_mlp", "per_sample_b1_mimic", "per_sample_b1_wide", "per_sample_b1_feature6"]


class PerSampleGolden:
    """tests/golden/per_sample_b1_*.npz: the REFERENCE run with ONE sample per batch (its own encoder order, its own NaN rows,
    weights frozen by an optimizer proxy that does not step; tests/golden/make_golden.py::run_reference_b1).  The N samples
    as ONE N-row batch in per-sample mode must give the means of the reference's N results."""

    def __init__(self, name):
        self.z = z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = json.loads(str(z["config_json"]))
        self.spec = spec_from_cfg(self.cfg)
        self.N = self.cfg["N"]
        self.xs = [z[f"x{k}"] for k in range(self.spec.E)]
        self.y, self.seq = z["y"], z["seq"]
        self.masks = {e: z[f"mask{e}"] for e in range(self.spec.E) if f"mask{e}" in z.files} or None
        self.grad_none = set(str(n) for n in z["grad_none"])

    def init_params(self):
        return {n: self.z["init/" + n].copy() for n in self.spec.param_names()}

    def check(self, err_loss, state_change, loss, row_counts, grads, counters=None, tol=1e-5, tolg=2e-5):
        """A per-sample step's results against the reference's means (loss cells / state change / loss 1e-5, executed-row
        counts exact, gradients 2e-5 of the tensor's largest element; an encoder nobody executed has no gradient)."""
        z = self.z
        assert rel_err(err_loss, z["sample_cells"].sum(0) / self.N) < tol
        assert rel_err(err_loss, z["hist/loss"]) < tol               # the reference's own epoch mean over its N one-sample steps
        assert rel_err(state_change, z["hist/state_change"]) < tol
        assert abs(float(loss) - z["sample_loss"].mean()) / abs(z["sample_loss"].mean()) < tol
        assert np.array_equal(np.asarray(row_counts, np.int64)[1:], z["executed"].sum(0))
        for n in self.spec.param_names():
            g = grads[n]
            if n in self.grad_none:
                assert g is None or float(np.max(np.abs(g))) == 0.0, n
            else:
                want = z["mean_grad/" + n]
                assert rel_err(np.asarray(g).reshape(want.shape), want) < tolg, (n, rel_err(np.asarray(g).reshape(want.shape), want))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def build_torch_model(spec, params, device, lib):
    """Instantiate multimodn_amd modules for an oracle ModelSpec and load `params`."""
    import torch
    import torch.nn.functional as F
    acts = {O.ACT_RELU: F.relu, O.ACT_SIGMOID: torch.sigmoid}
    acts[O.ACT_IDENTITY] = lib.encoders._identity
    encoders = []
    for e in spec.encoders:
        if e.kind == "mimic":
            enc = lib.MIMIC_MLPEncoder(spec.state_size, e.n_features, tuple(e.hidden), dropout=e.dropout,
                                       activation=acts[e.activation])
        else:
            enc = lib.MLPEncoder(spec.state_size, e.n_features, tuple(e.hidden), acts[e.activation])
        encoders.append(enc)
    decoders = [lib.LogisticDecoder(spec.state_size) if spec.dec(d).kind == "class" else
                lib.MLPDecoder(spec.state_size, tuple(spec.dec(d).hidden), 2, hidden_activation=acts[spec.dec(d).hidden_activation])
                for d in range(spec.D)]
    model = lib.MultiModN(spec.state_size, encoders, decoders, spec.err_penalty, spec.state_change_penalty,
                          device=torch.device(device))
    sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in params.items()}
    model.load_state_dict(sd)
    return model


# ------------------------------------------------------------------------------------------------
# fp64 yardstick.  After tens of Adam steps two correct fp32 implementations of the same step differ by far more than
# 1e-5 on individual weights: Adam divides by sqrt(v), so on coordinates whose gradient is of rounding size an fp32
# rounding difference moves the update by O(lr).  The principled bar is therefore not a widened constant but the
# distance to the EXACT trajectory: run the pinned oracle in fp64 from the golden's initial weights over the golden's
# batches (and recorded dropout masks) and require the implementation under test to sit no further from it than a small
# multiple of what the reference's own fp32 run (the golden's final weights / History) sits from it.
# ------------------------------------------------------------------------------------------------
_FP64_CACHE = {}
TIE_MARGIN = 1e-6


def fp64_trajectory(g):
    """Golden `g` replayed in float64: (final params, History loss [epochs, E+1, D], History state change [epochs, E],
    ties [epochs, E+1, D], n_samples [epochs, E+1]).  ties = how many (sample, batch) pairs of that epoch had the two
    sigmoid outputs of that grid cell within TIE_MARGIN of each other: only those predictions can legitimately flip
    between two correct implementations (argmax, multimodn.py:144); n_samples = the accuracy denominators
    (1 + rows seen, multimodn.py:105,121,171)."""
    if g.name not in _FP64_CACHE:
        spec = g.spec
        params = {n: np.asarray(v, np.float64) for n, v in g.init_params().items()}
        opt = O.Adam(g.cfg["lr"])
        losses, scs, ties, nsamp, s = [], [], [], [], 0
        for _ in range(g.epochs):
            results, sizes = [], []
            tie = np.zeros((spec.E + 1, spec.D), np.int64)
            ns = np.ones(spec.E + 1)
            for bi in range(g.n_batches):
                b = g.batch(bi)
                r = O.forward_backward(params, spec, b[0], b[1], b[2] if len(b) > 2 else None, dtype=np.float64,
                                       drop_masks=g.step_masks(s), keep_states=True)
                for row, st in r.states.items():
                    o = O.decoder_outputs(params, spec, st, np.float64)
                    tie[row] += (np.abs(o[:, :, 1] - o[:, :, 0]) < TIE_MARGIN).sum(axis=0)
                    ns[row] += len(b[1])
                opt.step(params, r.grads)
                results.append(r)
                sizes.append(len(b[1]))
                s += 1
            er = O.aggregate_epoch(spec.E, spec.D, results, sizes)
            losses.append(er.loss)
            scs.append(er.state_change)
            ties.append(tie)
            nsamp.append(ns)
        _FP64_CACHE[g.name] = (params, np.stack(losses), np.stack(scs), np.stack(ties), np.stack(nsamp))
    return _FP64_CACHE[g.name]


def assert_counts_match(hist, z, g, tag="train"):
    """accuracy / sensitivity / specificity / balanced accuracy are ratios of integer counts (multimodn.py:144-157,
    222-242): against the reference's History they must be EQUAL, except in grid cells where the fp64 replay finds
    predictions whose two outputs tie within TIE_MARGIN - there the count may move by at most that many flips."""
    _, _, _, ties, nsamp = fp64_trajectory(g)
    acc = np.stack(hist.accuracy[tag])
    clean = ties == 0
    d_counts = np.abs(acc - z["hist/accuracy"]) * nsamp[:, :, None]
    assert (d_counts <= ties + 1e-6).all(), ("prediction flips beyond the near-ties", float(d_counts.max()), int(ties.max()))
    for k in ("sensitivity", "specificity", "balanced_accuracy"):
        got, ref = np.stack(getattr(hist, k)[tag]), z["hist/" + k]
        assert np.array_equal(got[clean], ref[clean]), k
        # a flip moves tp/(tp+fn) by at most flips / (tp+fn) >= ... bounded by the cell's accuracy movement x n / positives;
        # with near-ties present only the count bound above is asserted


def fp32_noise_ratio(got, ref, truth):
    """max|got - truth| / max|ref - truth| (inf if the reference sits exactly on the truth and `got` does not)."""
    got, ref, truth = (np.asarray(a, np.float64) for a in (got, ref, truth))
    e_got, e_ref = np.abs(got.reshape(truth.shape) - truth).max(), np.abs(ref.reshape(truth.shape) - truth).max()
    return 0.0 if e_got == 0 else (np.inf if e_ref == 0 else e_got / e_ref)


def adam_well_conditioned(opt64, name=None, floor=1e-6, rel=1e-3, into=None):
    """Coordinates on which Adam's update is a well-conditioned function of the gradients, judged on the float64 replay
    `opt64` (an oracle.Adam) as it stands NOW: sqrt(v_hat) >= max(floor, rel x the tensor's largest).  Call it after every
    step with `into` = a dict {name: mask}: the masks are ANDed over the steps (one ill-conditioned step is enough to
    leave two correct implementations apart for good); with `name` it returns that tensor's mask for the current step.
    Adam normalises every coordinate by ITS OWN gradient scale: the update is m_hat / (sqrt(v_hat) + eps).  fp32 leaves a
    gradient element with an ABSOLUTE error of ~1e-7 of the tensor's largest gradient (the products that feed it carry
    that much), so a coordinate whose gradient is 1000x smaller than the tensor's largest carries a RELATIVE error of
    1e-4, which Adam passes on as 1e-4 of a full lr-sized step; below `floor` = 100 x Adam's eps the denominator is eps
    itself and the amplification is lr / eps (first steps: v_hat = g^2 of ONE batch).  Any two fp32 implementations end
    up a visible fraction of a step apart on such coordinates (with 30 % of the modalities missing whole rows of weights
    are in that regime): they carry no information about correctness, the well-conditioned ones do."""
    def one(n):
        st = opt64.state[n]
        s_hat = np.sqrt(np.asarray(st["v"], np.float64) / (1.0 - opt64.betas[1] ** st["step"]))
        return s_hat >= max(floor, rel * float(s_hat.max()))
    if into is not None:
        for n in opt64.state:
            m = one(n)
            into[n] = m if n not in into else (into[n] & m)
        return into
    return one(name)


def assert_within_fp32_noise(got, ref, truth, what="", factor=4.0, tight=2e-5, mask=None):
    """`got` agrees with the reference's value `ref` to `tight` (relative to the tensor's max) outright, or it is no
    further from the fp64 truth than `factor` x the reference's own fp32 run is.  `mask`: compare these coordinates only
    (adam_well_conditioned)."""
    got = np.asarray(got, np.float64).reshape(np.asarray(truth).shape)
    if mask is not None:
        mask = np.asarray(mask).reshape(got.shape)
        if not mask.any():
            return
        scale = max(float(np.max(np.abs(np.asarray(ref, np.float64)))), 1e-30)
        got, ref, truth = got[mask], np.asarray(ref, np.float64).reshape(mask.shape)[mask], np.asarray(truth, np.float64)[mask]
        if float(np.max(np.abs(got - ref))) / scale <= tight:
            return
    if rel_err(got, ref) <= tight:
        return
    ratio = fp32_noise_ratio(got, ref, truth)
    assert ratio <= factor, (what, f"|got-fp64| / |ref-fp64| = {ratio:.2f}", f"got vs ref {rel_err(got, ref):.2e}",
                            f"ref vs fp64 {rel_err(ref, truth):.2e}")


# ------------------------------------------------------------------------------------------------
# Trained weights at FULL size (no golden: the reference never ran these).  The yardstick is the same - distance to the
# float64 trajectory, against the fp32 numpy oracle's own distance - with one addition the small goldens never needed:
# relu kinks.  With 4096 rows x 32 units x 8 hidden layers x several steps, some pre-activation lands within fp32
# rounding of zero in most runs; act'(.) of that sample then is 0 in one correct implementation and 1 in another, and
# the sample's whole contribution to the gradients upstream of the unit comes or goes (~1/sqrt(batch) of an element:
# 1e4 x rounding noise).  That is a property of the loss (it is not differentiable there), not of an implementation, so
# a tensor is taken out of the comparison when - and only when - the HIP path's activation pattern (h > 0) of a layer
# that feeds its gradient actually DIFFERS from the float64 replay's on some sample of some step.
# ------------------------------------------------------------------------------------------------
def full_size_trajectories(lib, spec, params, batches, lr, steps_fn=None):
    """`len(batches)` fused Adam steps on the HIP path next to the numpy oracle in fp32 and fp64 (same initial weights,
    same batches).  Returns (model, p32, p64, flipped): `flipped` = names of the parameters whose gradient crossed a relu
    kink differently on the HIP path than in the fp64 replay (MLPEncoder family: a hidden MLP sees x only, so layer l of
    encoder e taints layers 0..l of that encoder and nothing else)."""
    import torch
    model = build_torch_model(spec, params, "cuda", lib)
    model.nan_policy = "device"
    opt = lib.optim.Adam(model.parameters(), lr=lr)
    p32 = {n: np.asarray(v, np.float32).copy() for n, v in params.items()}
    p64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    o32, o64 = O.Adam(lr), O.Adam(lr)
    B = len(batches[0][1])
    eng = model._get_engine(B)
    eng.begin_sequence()
    eng.epoch_reset()
    pairs = [(k, k) for k in range(spec.E)]
    alpha, beta = float(model.err_penalty), float(model.state_change_penalty)
    ML = lib.hip.MAX_LAYERS
    flipped = set()
    for xs, y in batches:
        dx = [torch.from_numpy(x).cuda() for x in xs]
        dy = torch.from_numpy(y).cuda()
        b = eng.make_batch(dx, dy, pairs, device_nan_flags=False)
        assert eng.local_step(b, alpha, beta, accumulate=True, optimizer=opt)
        opt.step()
        torch.cuda.synchronize()
        r64 = O.forward_backward(p64, spec, xs, y, dtype=np.float64, keep_states=True)
        for (e, l), h64 in r64.hidden.items():
            if spec.encoders[e].activation != O.ACT_RELU:
                continue
            h_hip = eng.debug_tensor(6, e * ML + l, eng.max_batch, h64.shape[1])[:B].cpu().numpy()
            if ((h_hip > 0) != (h64 > 0)).any():
                for j in range(l + 1):
                    flipped.update({f"encoders.{e}.layers.{j}.weight", f"encoders.{e}.layers.{j}.bias"})
        o64.step(p64, r64.grads)
        r32 = O.forward_backward(p32, spec, xs, y)
        o32.step(p32, r32.grads)
    return model, p32, p64, flipped


def assert_predictions_match(got, want, margins, what=""):
    """argmax predictions (multimodn.py:144,443): EQUAL, except where the float64 oracle's two outputs sit within
    TIE_MARGIN of each other (`margins` = |o_1 - o_0| from oracle.predict / per_sample_eval run in float64)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = (got != want) & (np.asarray(margins) >= TIE_MARGIN)
    assert not bad.any(), (what, int(bad.sum()), "predictions differ away from any tie")


def auc_slack(y_true, prob64):
    """How far the exact AUROC can move when scores within TIE_MARGIN of each other change order: the (positive, negative)
    pairs that close, over all pairs."""
    y_true, prob64 = np.asarray(y_true), np.asarray(prob64, np.float64)
    pos, neg = prob64[y_true == 1], prob64[y_true == 0]
    if len(pos) == 0 or len(neg) == 0:
        return 0.0
    close = (np.abs(pos[:, None] - neg[None, :]) < TIE_MARGIN).sum()
    return float(close) / (len(pos) * len(neg))


def featurewise_pipeline(g, device, lib, engine_factory=None):
    """The body of pipelines/titanic/titanic_featurewise_pipeline.py:44-73 / titanic_missingness_pipeline.py:46-74 on a
    feature-wise golden's data: FeatureWiseDataset over the [N, n_features] table -> stock DataLoader (no shuffle) ->
    one MLPFeatureEncoder per feature + one LogisticDecoder, built by the pipelines' constructor calls, the golden's initial
    weights loaded by name.  Returns (model, loader)."""
    import torch
    import torch.nn.functional as F
    c = g.cfg
    batches = g.batches()
    X = np.concatenate([np.concatenate(b[0], axis=1) for b in batches], axis=0)
    y = np.concatenate([b[1] for b in batches], axis=0)
    loader = torch.utils.data.DataLoader(lib.FeatureWiseDataset(X, y), c["B"])
    encoders = [lib.MLPFeatureEncoder(c["S"], c["H"][0], F.relu) for _ in c["F"]]
    decoders = [lib.LogisticDecoder(c["S"]) for _ in range(c["D"])]
    model = lib.MultiModN(c["S"], encoders, decoders, c["pen"][0], c["pen"][1], device=torch.device(device))
    if engine_factory is not None:
        model._engine_factory = engine_factory
    model.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in g.init_params().items()})
    return model, loader


def assert_history_matches_golden(hist, g, tol=2e-6):
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < tol
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < tol
    for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        got = np.stack(getattr(hist, k)["train"])
        assert got.dtype == z["hist/" + k].dtype and got.shape == z["hist/" + k].shape
        assert np.array_equal(got, z["hist/" + k]), k
