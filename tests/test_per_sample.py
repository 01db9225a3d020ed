"""Per-sample mode (BASELINE.json configs[4]): per-sample missing modalities (NaN rows)
and per-sample encoder order.  The reference only defines this at batch size 1 (multimodn.py:168,
518-523); the build-defined batch result is the mean over the samples of the reference's
batch-size-1 result, restated in oracle.per_sample_step (which literally loops the pinned
batch-size-1 oracle over the samples).  GPU tests compare the fused kernel's per-sample mode with it:
1e-5 relative on loss cells / state change, exact integer counters and row counts, 2e-5 of max|g| on
gradients."""
import numpy as np
import pytest
import torch

import multimodn_amd as mm
from helpers import (PerSampleGolden, TIE_MARGIN, adam_well_conditioned, assert_predictions_match, assert_within_fp32_noise, auc_slack, build_torch_model,
                     rel_err)
from oracle import multimodn_oracle as O
from oracle_engine import OracleEngine


def c5_like(B, E=4, F=8, S=32, D=2, H=(8, 8), seed=0, p_missing=0.3, permute=True):
    spec = O.ModelSpec(S, [O.EncoderSpec(F, H, O.ACT_RELU) for _ in range(E)], D, 1.0, 0.3)
    rng = np.random.default_rng(seed)
    xs, y = O.synthetic_batches(spec, B, B, seed=seed + 1)[0]
    # missing-not-at-random: modality e is missing more often when y_0 = 1
    pm = np.where(y[:, :1] == 1, 1.5 * p_missing, 0.5 * p_missing)
    miss = rng.random((B, E)) < pm
    xs = [x.copy() for x in xs]
    for e in range(E):
        xs[e][miss[:, e]] = np.nan
    seq = np.stack([rng.permutation(E) for _ in range(B)]).astype(np.int64) if permute else None
    return spec, xs, y, seq


def check(stats, grads, ref, tol=1e-5, tolg=2e-5):
    assert rel_err(stats["err_loss"], ref.err_loss) < tol
    assert rel_err(stats["state_change"], ref.state_change) < tol
    assert rel_err(stats["loss"], ref.loss) < tol
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    assert np.array_equal(stats["rows"].astype(np.int64), ref.row_counts)
    for n, g in ref.grads.items():
        got = grads[n]
        if g is None:
            assert np.abs(got).max() == 0.0, n
        else:
            assert rel_err(got.reshape(g.shape), g) < tolg, (n, rel_err(got.reshape(g.shape), g))


def run_step(model, xs, y, seq, torch_regroup=False):
    model.per_sample = True
    data = [torch.from_numpy(x) for x in xs]
    eng = model._get_engine(len(y))
    eng._torch_regroup = torch_regroup      # regrouping by torch ops instead of mmn_regroup's kernels
    eng.epoch_reset()
    _, keep = model._run_step_per_sample(eng, data, torch.from_numpy(y), None if seq is None else torch.from_numpy(seq))
    eng.assign_grads(None)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    stats = {k: np.array(v) for k, v in eng.step_values().items()}
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    return stats, grads


def test_per_sample_host_logic_on_checker_backend():
    spec, xs, y, seq = c5_like(24, E=3, seed=3)
    params = O.init_params(spec, 1)
    model = build_torch_model(spec, params, "cpu", mm)
    model._engine_factory = OracleEngine
    stats, grads = run_step(model, xs, y, seq)
    check(stats, grads, O.per_sample_step(params, spec, xs, y, seq), tol=1e-6, tolg=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [8, 6])      # 6: rows that are not whole float4s (the regrouping kernels' 4-byte path)
@pytest.mark.parametrize("B,E,permute", [(1, 2, True), (37, 3, True), (200, 4, True), (130, 4, False), (16, 1, True)])
def test_per_sample_step_matches_oracle(B, E, permute, F):
    mm.hip.load()
    spec, xs, y, seq = c5_like(B, E=E, F=F, seed=B)
    if not permute:
        seq = None
    params = O.init_params(spec, 2)
    model = build_torch_model(spec, params, "cuda", mm)
    stats, grads = run_step(model, xs, y, seq)
    check(stats, grads, O.per_sample_step(params, spec, xs, y, seq))


@pytest.mark.gpu
@pytest.mark.parametrize("B,E,F,H,S", [(37, 5, 8, (8, 8), 32), (200, 6, 1, (5,), 5), (700, 8, 6, (8,), 20), (64, 8, 72, (48,), 24), (1, 7, 4, (), 16)])
def test_per_sample_step_of_five_to_eight_encoders_matches_oracle(B, E, F, H, S):
    """Five to eight encoders (round 6; 200 x 6 x 1 x (5,) x 5: the reference's feature-wise missingness pipeline,
    pipelines/titanic/titanic_missingness_pipeline.py:26-74, as a BATCH instead of one passenger per step): per-sample
    missing modalities in the default encoder order - the regrouping's groups are the 2^E subsets - on the generic tier's
    tiled form, against the oracle's loop over the samples."""
    mm.hip.load()
    spec, xs, y, _ = c5_like(B, E=E, F=F, S=S, H=H, seed=B + E, permute=False)
    params = O.init_params(spec, 2)
    model = build_torch_model(spec, params, "cuda", mm)
    stats, grads = run_step(model, xs, y, None)
    assert model._engine._generic_tier
    check(stats, grads, O.per_sample_step(params, spec, xs, y, None), tolg=3e-5)
    again = run_step(build_torch_model(spec, params, "cuda", mm), xs, y, None)      # the layout is a pure function of the batch
    for n in grads:
        assert np.array_equal(grads[n], again[1][n]), n


@pytest.mark.gpu
def test_per_sample_step_matches_the_reference_run_at_batch_size_one():
    """tests/golden/per_sample_b1_mlp.npz: the reference fed 32 samples one per batch (own encoder order, NaN rows, frozen
    weights); the HIP per-sample step over the same 32 rows as ONE batch must give the means of the reference's results."""
    mm.hip.load()
    g = PerSampleGolden("per_sample_b1_mlp")
    model = build_torch_model(g.spec, g.init_params(), "cuda", mm)
    stats, grads = run_step(model, g.xs, g.y, g.seq)
    g.check(stats["err_loss"], stats["state_change"], stats["loss"], stats["rows"], grads)
    n = 1.0 + stats["rows"].astype(np.float64)[:, None]
    assert np.array_equal(stats["n_correct"] / n, g.z["hist/accuracy"])          # the reference's epoch accuracy, exactly


@pytest.mark.gpu
def test_hip_regrouping_equals_torch_regrouping():
    """mmn_regroup (k_ps_code / k_ps_hist / k_ps_layout / k_ps_gather: one workgroup per 512 rows) and the torch-op
    regrouping build the same layout: bit-identical statistics and gradients - also where a batch is not a whole number of
    512-row blocks, and beyond the 16,384 rows the one-workgroup layout of round 2 stopped at."""
    mm.hip.load()
    for B, E, perm in ((333, 4, True), (64, 3, False), (4096, 4, True), (1025, 4, True), (5000, 3, True), (20000, 4, True)):
        spec, xs, y, seq = c5_like(B, E=E, seed=7 + B)
        if not perm:
            seq = None
        params = O.init_params(spec, 2)
        a = run_step(build_torch_model(spec, params, "cuda", mm), xs, y, seq, torch_regroup=False)
        b = run_step(build_torch_model(spec, params, "cuda", mm), xs, y, seq, torch_regroup=True)
        for k in a[0]:
            assert np.array_equal(a[0][k], b[0][k]), k
        for n in a[1]:
            assert np.array_equal(a[1][n], b[1][n]), n


@pytest.mark.gpu
def test_per_sample_equals_batch_mode_when_nothing_varies():
    """No missing modality, one common order: per-sample mode must reproduce the ordinary batch step."""
    mm.hip.load()
    spec, xs, y, _ = c5_like(100, E=4, seed=5, p_missing=0.0, permute=False)
    params = O.init_params(spec, 4)
    model = build_torch_model(spec, params, "cuda", mm)
    stats, grads = run_step(model, xs, y, None)
    ref = O.forward_backward(params, spec, xs, y)
    assert rel_err(stats["err_loss"], ref.err_loss) < 1e-5 and rel_err(stats["loss"], ref.loss) < 1e-5
    for n, g in ref.grads.items():
        assert rel_err(grads[n].reshape(g.shape), g) < 2e-5, n


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c5_like", "featurewise6_host", "featurewise6_device"])
def test_per_sample_training_epochs_match_oracle(case):
    """train_epoch in per-sample mode (fused Adam included) against the oracle loop: History and weights.
    featurewise6: the reference's missingness pipeline (six MLPFeatureEncoder-shaped encoders of one feature each, state 5,
    pipelines/titanic/titanic_missingness_pipeline.py:26-74) trained in BATCHES of 32 passengers with their own missing features
    instead of one passenger per step - host batches, and device-resident ones (the look-ahead regrouping on a side stream)."""
    mm.hip.load()
    if case == "c5_like":
        spec, xs, y, seq = c5_like(96, E=4, seed=11)
    else:
        spec, xs, y, seq = c5_like(96, E=6, F=1, S=5, H=(5,), seed=11, permute=False)
    params = O.init_params(spec, 6)
    model = build_torch_model(spec, params, "cuda", mm)
    model.per_sample = True
    opt = mm.optim.Adam(list(model.parameters()), 1e-2)
    hist = mm.MultiModNHistory(["a", "b"])
    put = (lambda a: torch.from_numpy(a).cuda()) if case.endswith("_device") else torch.from_numpy
    loader = [([put(x[s:s + 32]) for x in xs], put(y[s:s + 32])) + ((put(seq[s:s + 32]),) if seq is not None else ())
              for s in range(0, 96, 32)]
    if seq is None:
        seq = np.tile(np.arange(spec.E, dtype=np.int64), (96, 1))            # (for the oracle: the default order, written out)
    oparams = {n: v.copy() for n, v in params.items()}
    oparams64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    oopt, oopt64 = O.Adam(1e-2), O.Adam(1e-2)
    cond = {}
    for ep in range(2):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        results, sizes = [], []
        for s in range(0, 96, 32):
            r = O.per_sample_step(oparams, spec, [x[s:s + 32] for x in xs], y[s:s + 32], seq[s:s + 32])
            oopt.step(oparams, {n: (g if g is not None else np.zeros_like(oparams[n])) for n, g in r.grads.items()})
            r64 = O.per_sample_step(oparams64, spec, [x[s:s + 32] for x in xs], y[s:s + 32], seq[s:s + 32], dtype=np.float64)
            oopt64.step(oparams64, {n: (g if g is not None else np.zeros_like(oparams64[n])) for n, g in r64.grads.items()})
            adam_well_conditioned(oopt64, into=cond)
            results.append(r); sizes.append(32)
        er = O.aggregate_epoch(spec.E, spec.D, results, sizes)
        assert rel_err(hist.loss["train"][ep], er.loss) < 1e-5
        assert rel_err(hist.state_change_loss[ep], er.state_change) < 1e-5
        assert np.abs(hist.accuracy["train"][ep] - er.accuracy).max() <= 1.0 / 96 + 1e-12
    # trained weights: 2e-5 of the fp32 oracle's outright, or no further from the float64 trajectory than 4x the fp32 oracle
    # is - on the coordinates where Adam's update is well conditioned (with 30 % of the modalities missing, whole rows of
    # weights see gradients of Adam's eps size: helpers.adam_well_conditioned)
    for n, p in model.named_parameters():
        assert cond[n].mean() > 0.5 or (case != "c5_like" and cond[n].any()), n     # (five hidden units: a dead relu unit is a fifth of a tensor)
        assert_within_fp32_noise(p.detach().cpu().numpy(), oparams[n], oparams64[n], n, mask=cond[n])


@pytest.mark.gpu
@pytest.mark.parametrize("fork", ["1", "0"])
@pytest.mark.parametrize("family", ["mlp", "mimic"])
def test_replayed_per_sample_groups_equal_eager_steps(family, fork, monkeypatch):
    """Round 4: groups of 8 device-resident per-sample batches are replayed as ONE hipGraph from their second sighting on
    (regrouping launches of every batch - on a graph branch of their own, or in line - the dropout draw and the step).
    Same launches in the same order as the eager loop: History and trained weights are BITWISE those of eager steps,
    device-drawn dropout included; 11 batches per epoch = one group + three eager steps."""
    mm.hip.load()
    monkeypatch.setenv("MMN_PS_GRAPH_FORK", fork)
    if family == "mlp":
        spec, xs, y, seq = c5_like(11 * 48, E=4, seed=31)
    else:
        spec = O.ModelSpec(128, [O.EncoderSpec(24, (32, 20), O.ACT_RELU, kind="mimic", dropout=0.2) for _ in range(3)], 2, 1.0, 0.3,
                           decoders=[O.DecoderSpec("mlp", (16,)) for _ in range(2)])
        rng = np.random.default_rng(3)
        xs, y = O.synthetic_batches(spec, 11 * 48, 11 * 48, seed=4)[0]
        xs = [x.copy() for x in xs]
        for e in range(3):
            xs[e][rng.random(len(y)) < 0.3] = np.nan
        seq = np.stack([rng.permutation(3) for _ in range(len(y))]).astype(np.int64)
    params = O.init_params(spec, 2)
    loader = [([torch.from_numpy(x[s:s + 48]).cuda() for x in xs], torch.from_numpy(y[s:s + 48]).cuda(), torch.from_numpy(seq[s:s + 48]).cuda())
              for s in range(0, 11 * 48, 48)]

    def train(replay):
        torch.manual_seed(7)
        model = build_torch_model(spec, params, "cuda", mm)
        model.per_sample = True
        model.replay_steps = replay
        opt = mm.optim.Adam(list(model.parameters()), 1e-2)
        hist = mm.MultiModNHistory(["a", "b", "c"][:spec.D])
        for _ in range(4):
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        torch.cuda.synchronize()
        eng = model._get_engine(48)
        return hist, {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}, int(getattr(eng, "_graph_hits", 0))

    h1, w1, hits1 = train(True)
    h0, w0, hits0 = train(False)
    assert hits1 >= 2 and hits0 == 0                        # epochs 3 and 4 replayed the group captured in epoch 2
    for ep in range(4):
        assert np.array_equal(h1.loss["train"][ep], h0.loss["train"][ep]), ep
        assert np.array_equal(h1.state_change_loss[ep], h0.state_change_loss[ep]), ep
        assert np.array_equal(h1.accuracy["train"][ep], h0.accuracy["train"][ep]), ep
    for n in w0:
        assert np.array_equal(w1[n], w0[n]), n
    assert np.isfinite(h1.loss["train"][-1]).all() and h1.loss["train"][-1].mean() < h1.loss["train"][0].mean()


@pytest.mark.gpu
def test_c5_full_size_against_oracle():
    """BASELINE configs[4] shape: 4 x 64 features, hidden (32,32), state 128, 3 tasks, batch 4096,
    30 % missing-not-at-random, a random encoder order per sample."""
    mm.hip.load()
    spec, xs, y, seq = c5_like(4096, E=4, F=64, S=128, D=3, H=(32, 32), seed=1)
    params = O.init_params(spec, 0)
    model = build_torch_model(spec, params, "cuda", mm)
    stats, grads = run_step(model, xs, y, seq)
    ref = O.per_sample_step(params, spec, xs, y, seq)
    ref64 = O.per_sample_step(params, spec, xs, y, seq, dtype=np.float64)
    assert rel_err(stats["err_loss"], ref64.err_loss) < 1e-5 and rel_err(stats["state_change"], ref64.state_change) < 1e-5
    assert np.array_equal(stats["rows"].astype(np.int64), ref.row_counts)
    for n, g in ref64.grads.items():
        assert rel_err(grads[n].reshape(g.shape), g) < 2e-5, (n, rel_err(grads[n].reshape(g.shape), g))


@pytest.mark.gpu
def test_per_sample_forward_only_entry_points():
    """test() / predict() / get_states() in per-sample mode against the per-sample loop of the oracle."""
    mm.hip.load()
    spec, xs, y, seq = c5_like(150, E=4, seed=21)
    params = O.init_params(spec, 8)
    model = build_torch_model(spec, params, "cuda", mm)
    model.per_sample = True
    ref, preds, states, last_out, last_idx = O.per_sample_eval(params, spec, xs, y, seq)
    marg = []
    p64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    _, _, _, last_out64, _ = O.per_sample_eval(p64, spec, xs, y, seq, dtype=np.float64, margins=marg)
    loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y), torch.from_numpy(seq))]
    hist = mm.MultiModNHistory(["a", "b"])
    results = model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="val")
    ep = O.aggregate_epoch(spec.E, spec.D, [ref], [150])
    assert rel_err(hist.loss["val"][0], ep.loss) < 1e-5
    assert np.array_equal(hist.accuracy["val"][0], ep.accuracy)
    assert np.array_equal(hist.balanced_accuracy["val"][0], ep.balanced_accuracy)
    # report of the decoders on the last encoder's state, over the samples that have it
    for d in range(spec.D):
        p = last_out[:, d, :] / last_out[:, d, :].sum(axis=1, keepdims=True)
        want = O.performance_metrics(y[last_idx, d], (p[:, 1] > p[:, 0]).astype(np.int64), p[:, 1])
        got = dict(zip(mm.metrics.performance_metrics, results[d]))
        # counts: equal, up to the predictions of this decoder whose float64 outputs tie; the exact AUROC: up to the
        # (positive, negative) pairs whose float64 scores tie
        p64d = last_out64[:, d, :] / last_out64[:, d, :].sum(axis=1, keepdims=True)
        ties = int((np.abs(last_out64[:, d, 1] - last_out64[:, d, 0]) < TIE_MARGIN).sum())
        assert abs(int(got["tp"]) - want["tp"]) <= ties
        assert abs(float(got["auc"]) - want["auc"]) <= auc_slack(y[last_idx, d], p64d[:, 1]) + 1e-6
    got_states = torch.stack(model.get_states(loader)).cpu().numpy()
    assert rel_err(got_states, states) < 1e-5
    got_pred = model.predict([torch.from_numpy(x) for x in xs], torch.from_numpy(seq))
    assert_predictions_match(got_pred, preds, marg[0], "per-sample predict()")


@pytest.mark.gpu
def test_per_sample_step_of_a_wide_model_matches_the_reference_run_at_batch_size_one():
    """tests/golden/per_sample_b1_wide.npz (round 6): an MLPEncoder model OUTSIDE the fused kernel's tiled form
    (72 features > 64, hidden 48 > 32) - the reference runs any shape at batch size 1 (multimodn/multimodn.py:509-531).  The
    engine plans the generic tier for it (mmn_model.flags, MMN_MODEL_GENERIC_TIER) and the step over the 32 rows as ONE batch
    gives the means of the reference's 32 one-sample results."""
    mm.hip.load()
    g = PerSampleGolden("per_sample_b1_wide")
    model = build_torch_model(g.spec, g.init_params(), "cuda", mm)
    stats, grads = run_step(model, g.xs, g.y, g.seq)
    assert model._engine._generic_tier and mm.hip.load().mmn_per_sample_supported(model._engine._plan)
    g.check(stats["err_loss"], stats["state_change"], stats["loss"], stats["rows"], grads)
    n = 1.0 + stats["rows"].astype(np.float64)[:, None]
    assert np.array_equal(stats["n_correct"] / n, g.z["hist/accuracy"])


@pytest.mark.gpu
def test_per_sample_batch_of_the_missingness_pipeline_matches_the_reference_run_at_batch_size_one():
    """tests/golden/per_sample_b1_feature6.npz (round 6): the reference's feature-wise missingness pipeline - six
    MLPFeatureEncoder(state 5, hidden 5), one LogisticDecoder, penalties 0.7 / 0.3, default encoder order
    (pipelines/titanic/titanic_missingness_pipeline.py:26-74) - fed 48 passengers ONE PER BATCH, as that pipeline does.  Built
    here from the pipeline's own constructor calls, the 48 passengers as ONE per-sample batch give the means of the reference's
    48 results: what lets that pipeline train in batches."""
    import torch.nn.functional as F
    mm.hip.load()
    g = PerSampleGolden("per_sample_b1_feature6")
    c = g.cfg
    assert np.array_equal(g.seq, np.tile(np.arange(6), (g.N, 1)))
    encoders = [mm.MLPFeatureEncoder(c["S"], c["H"][0], F.relu) for _ in c["F"]]
    model = mm.MultiModN(c["S"], encoders, [mm.LogisticDecoder(c["S"])], c["pen"][0], c["pen"][1], device=torch.device("cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.init_params().items()})
    stats, grads = run_step(model, g.xs, g.y, None)
    assert model._engine._generic_tier
    g.check(stats["err_loss"], stats["state_change"], stats["loss"], stats["rows"], grads)
    n = 1.0 + stats["rows"].astype(np.float64)[:, None]
    assert np.array_equal(stats["n_correct"] / n, g.z["hist/accuracy"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [dict(F=100, H=(64,), S=64), dict(F=8, H=(48, 16), S=32), dict(F=33, H=(), S=20), dict(F=128, H=(64, 64), S=128)])
def test_per_sample_mode_runs_every_mlp_encoder_shape(shape):
    """Per-sample steps of MLPEncoder shapes the fused kernel's tiled form does not take (wide features, wide / odd hidden
    layers, no hidden layer, a state that is no multiple of 16) against the oracle's loop over the samples; a model INSIDE
    the fused kernel's shapes keeps its plan (no generic tier)."""
    mm.hip.load()
    spec, xs, y, seq = c5_like(70, E=3, F=shape["F"], S=shape["S"], H=shape["H"], seed=5)
    params = O.init_params(spec, 2)
    model = build_torch_model(spec, params, "cuda", mm)
    stats, grads = run_step(model, xs, y, seq)
    assert model._engine._generic_tier
    check(stats, grads, O.per_sample_step(params, spec, xs, y, seq), tolg=3e-5)
    lean = build_torch_model(*(lambda sp: (sp, O.init_params(sp, 2)))(c5_like(16, E=2, seed=1)[0]), "cuda", mm)
    lean.per_sample = True
    assert not lean._get_engine(16)._generic_tier


@pytest.mark.gpu
def test_per_sample_training_of_a_wide_model_and_back_to_batch_mode():
    """ADVICE r4 / VERDICT r5 #8: the wide MLPEncoder model that per-sample mode used to refuse trains through the public
    train_epoch (host batches, fused Adam; the loss falls), evaluates, and - per_sample switched off again - trains in batch mode
    on its default plan (whole-batch semantics)."""
    lib = mm
    spec, xs, y, seq = c5_like(96, H=(48,), seed=2)
    model = build_torch_model(spec, O.init_params(spec, 1), "cuda", lib)
    model.per_sample = True
    opt = lib.optim.Adam(list(model.parameters()), 1e-2)
    hist = lib.MultiModNHistory(["a", "b"])
    loader = [([torch.from_numpy(x[s:s + 48]) for x in xs], torch.from_numpy(y[s:s + 48]), torch.from_numpy(seq[s:s + 48])) for s in (0, 48)]
    for _ in range(10):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    loss = np.stack(hist.loss["train"])
    assert np.isfinite(loss).all() and loss[-1].mean() < loss[0].mean()
    assert model._engine._generic_tier
    model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="val")
    assert np.isfinite(hist.loss["val"][0]).all()
    model.per_sample = False                                 # the same model trains in batch mode (whole-batch semantics)
    clean = [([torch.from_numpy(np.nan_to_num(x)) for x in xs], torch.from_numpy(y))]
    model.train_epoch(clean, opt, torch.nn.CrossEntropyLoss(), lib.MultiModNHistory(["a", "b"]))
    assert not model._engine._generic_tier


@pytest.mark.gpu
@pytest.mark.parametrize("E,with_seq", [(5, True), (9, False)])
def test_per_sample_mode_refuses_what_no_tier_runs_up_front(E, with_seq):
    """Per-sample encoder ORDER for more than four encoders (the regrouping's table of ordered subsets ends there; five to
    eight encoders run in the default order), and more than eight encoders at all (a tile's sequence code holds eight steps):
    refused where the engine is asked for - a message that says what per-sample mode covers, nothing launched or changed."""
    from multimodn_amd.engine import UnsupportedModelError
    spec, xs, y, seq = c5_like(48, E=E, seed=2)
    model = build_torch_model(spec, O.init_params(spec, 1), "cuda", mm)
    model.per_sample = True
    opt = mm.optim.Adam(list(model.parameters()), 1e-2)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) + ((torch.from_numpy(seq),) if with_seq else ())]
    with pytest.raises(UnsupportedModelError, match="per-sample mode"):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), mm.MultiModNHistory(["a", "b"]))
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["mlp", "mimic"])
def test_per_sample_stats_layouts_are_bitwise_equal(family, monkeypatch):
    """Round 5: the step statistics of a per-sample batch - loss cells, counters and the ROW COUNTS per grid row, summed over
    the tiles' sequence codes - are formed by one workgroup per 32 columns inside the k_wgrad launch (workgroup 0 counts the
    rows); MMN_STATS_COLS=0 keeps the single stats workgroup, MMN_SIDE=0 round 4's layout inside k_reduce.  History (the
    accuracies divide by the row counts) and trained weights must be identical bit for bit."""
    mm.hip.load()
    if family == "mlp":
        spec, xs, y, seq = c5_like(9 * 48, E=4, seed=13)
    else:
        spec = O.ModelSpec(128, [O.EncoderSpec(24, (32, 20), O.ACT_RELU, kind="mimic", dropout=0.0) for _ in range(3)], 2, 1.0, 0.3,
                           decoders=[O.DecoderSpec("mlp", (16,)) for _ in range(2)])
        rng = np.random.default_rng(5)
        xs, y = O.synthetic_batches(spec, 9 * 48, 9 * 48, seed=6)[0]
        xs = [x.copy() for x in xs]
        for e in range(3):
            xs[e][rng.random(len(y)) < 0.3] = np.nan
        seq = np.stack([rng.permutation(3) for _ in range(len(y))]).astype(np.int64)
    params = O.init_params(spec, 2)
    loader = [([torch.from_numpy(x[s:s + 48]).cuda() for x in xs], torch.from_numpy(y[s:s + 48]).cuda(), torch.from_numpy(seq[s:s + 48]).cuda())
              for s in range(0, 9 * 48, 48)]

    def train(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(7)
        model = build_torch_model(spec, params, "cuda", mm)
        model.per_sample = True
        opt = mm.optim.Adam(list(model.parameters()), 1e-2)
        hist = mm.MultiModNHistory(["a", "b", "c"][:spec.D])
        for _ in range(3):
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        torch.cuda.synchronize()
        return hist, {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    ref = train({"MMN_SIDE": "0", "MMN_STATS_COLS": "1"})
    for env in ({"MMN_SIDE": "1", "MMN_STATS_COLS": "0"}, {"MMN_SIDE": "1", "MMN_STATS_COLS": "1"}):
        got = train(env)
        for ep in range(3):
            for k in ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy"):
                assert np.array_equal(getattr(got[0], k)["train"][ep], getattr(ref[0], k)["train"][ep]), (env, k, ep)
            assert np.array_equal(got[0].state_change_loss[ep], ref[0].state_change_loss[ep]), (env, ep)
        for n in ref[1]:
            assert np.array_equal(got[1][n], ref[1][n]), (env, n)
