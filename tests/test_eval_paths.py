"""Forward-only entry points test() / predict() / get_states() (reference multimodn.py:255-492).

CPU part (not gpu): the oracle's restatement against the golden vectors the reference produced on
its own trained weights; the host logic of MultiModN.test/predict/get_states on the checker
backend; the native metrics (multimodn_amd/metrics.py) against the oracle's restatement of the
torchmetrics algorithms and against scikit-learn.
GPU part: the same entry points on the HIP engine against the golden vectors, every kernel tier.
Tolerance: 1e-5 relative on losses and states, predictions / accuracies / confusion-derived values exact."""
import numpy as np
import pytest
import torch

import multimodn_amd as mm
from helpers import GOLDEN_NAMES, Golden, assert_predictions_match, build_torch_model, rel_err
from oracle import multimodn_oracle as O
from oracle_engine import OracleEngine


def loader_of(g):
    out = []
    for b in g.batches():
        item = [[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]
        if len(b) > 2:
            item.append(torch.from_numpy(b[2]))
        out.append(tuple(item))
    return out


def trained_model(g, device, backend=None):
    model = build_torch_model(g.spec, {n: np.array(v) for n, v in g.final_params().items()}, device, mm)
    if backend is not None:
        model._engine_factory = backend
    return model


def check_eval_against_golden(g, model, tol=1e-5, flips=1):
    z = g.z
    loader = loader_of(g)
    hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    results = model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="test")
    assert rel_err(hist.loss["test"][0], z["eval/test_loss"]) < tol
    assert np.array_equal(hist.accuracy["test"][0], z["eval/test_accuracy"])
    assert np.array_equal(hist.sensitivity["test"][0], z["eval/test_sensitivity"])
    assert np.array_equal(hist.specificity["test"][0], z["eval/test_specificity"])
    assert np.array_equal(hist.balanced_accuracy["test"][0], z["eval/test_balanced_accuracy"])
    assert hist.loss["test"][0].dtype == np.float64 and hist.sensitivity["test"][0].dtype == np.float32
    # per-decoder report vs the oracle's restatement on the same weights
    _, outs = O.test_epoch({n: np.array(v) for n, v in g.final_params().items()}, g.spec, g.batches())
    assert len(results) == g.spec.D
    for d, (yt, yp, pr) in enumerate(outs):
        ref = O.performance_metrics(yt, yp, pr)
        got = dict(zip(mm.metrics.performance_metrics, results[d]))
        # these are functions of the RANKING / thresholding of fp32 scores: two implementations whose
        # scores differ in the last ulp may order one near-tied pair differently, which moves a count
        # by one sample and the AUC by 1 / (n_pos * n_neg); `flips` such events are tolerated
        # (0 for the checker backend, whose scores are the oracle's own)
        n, npos = len(yt), int(yt.sum())
        for k in ("tn", "fp", "fn", "tp"):
            assert abs(int(got[k]) - ref[k]) <= flips, (d, k)
        for k in ("f1", "accuracy", "sensitivity", "specificity"):
            assert abs(float(got[k]) - float(ref[k])) <= 1e-5 + 2.0 * flips / max(min(npos, n - npos), 1), (d, k, float(got[k]), ref[k])
        assert abs(float(got["auc"]) - ref["auc"]) <= 1e-5 + 2.0 * flips / max(npos * (n - npos), 1), (d, float(got["auc"]), ref["auc"])
        for k in ("fpr", "tpr", "precision", "recall", "thr_roc", "thr_pr"):
            a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
            # a curve has one point per DISTINCT score: fp32 scores that differ in the last ulp
            # between two implementations may merge / split points, so compare through the AUC
            # when the lengths differ, point-wise otherwise
            if a.shape == b.shape and flips == 0:
                assert np.allclose(a, b, atol=2e-5, equal_nan=True), (d, k)
    states = torch.stack(model.get_states(loader)).cpu().numpy()
    assert states.shape == z["eval/states"].shape
    assert rel_err(states, z["eval/states"]) < tol
    if "eval/predict" in z.files:
        b0 = g.batch(0)
        seq = torch.from_numpy(b0[2]) if len(b0) > 2 else None
        pred = model.predict([torch.from_numpy(x) for x in b0[0]], seq)
        assert pred.shape == z["eval/predict"].shape and pred.dtype == np.float64
        mism = (pred != z["eval/predict"]).mean()
        assert mism <= 1.0 / pred.shape[-1] + 1e-12      # a class can only flip when two sigmoids tie to rounding


# ------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_eval_matches_reference_golden(name):
    g = Golden(name)
    fin = {n: np.array(v) for n, v in g.final_params().items()}
    ep, _ = O.test_epoch(fin, g.spec, g.batches())
    assert rel_err(ep.loss, g.z["eval/test_loss"]) < 2e-6
    assert np.array_equal(ep.accuracy, g.z["eval/test_accuracy"])
    assert np.array_equal(ep.balanced_accuracy, g.z["eval/test_balanced_accuracy"])
    assert rel_err(O.get_states(fin, g.spec, g.batches()), g.z["eval/states"]) < 2e-6
    if "eval/predict" in g.z.files:
        b0 = g.batch(0)
        assert np.array_equal(O.predict(fin, g.spec, b0[0], b0[2] if len(b0) > 2 else None), g.z["eval/predict"])


@pytest.mark.parametrize("name", ["c2_split", "nan_skip", "seq_perm", "mlp_sigmoid"])
def test_host_logic_of_eval_entry_points(name):
    g = Golden(name)
    check_eval_against_golden(g, trained_model(g, "cpu", OracleEngine), tol=2e-6, flips=0)


def test_native_metrics_against_oracle_and_sklearn():
    from sklearn import metrics as M
    rng = np.random.default_rng(0)
    for n, q in ((500, 3), (64, 1), (1000, 4)):
        y = rng.integers(0, 2, n)
        p = np.round(rng.random(n), q).astype(np.float32)          # ties between scores on purpose
        pred = (p > 0.45).astype(np.int64)
        got = dict(zip(mm.metrics.performance_metrics,
                       mm.metrics.get_performance_metrics(torch.from_numpy(y), torch.from_numpy(pred), torch.from_numpy(p))))
        ref = O.performance_metrics(y, pred, p)
        assert abs(float(got["auc"]) - M.roc_auc_score(y, p)) < 1e-6
        assert abs(float(got["f1"]) - M.f1_score(y, (p > 0.5).astype(int))) < 1e-6
        assert abs(float(got["accuracy"]) - M.accuracy_score(y, pred)) < 1e-6
        fpr, tpr, thr = M.roc_curve(y, p, drop_intermediate=False)
        assert np.allclose(got["fpr"].numpy(), fpr, atol=1e-6) and np.allclose(got["tpr"].numpy(), tpr, atol=1e-6)
        assert np.allclose(got["thr_roc"].numpy()[1:], thr[1:])
        for k in ("fpr", "tpr", "precision", "recall", "thr_roc", "thr_pr"):
            assert np.allclose(np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64), atol=1e-6), k
        for k in ("tn", "fp", "fn", "tp"):
            assert int(got[k]) == ref[k]
    # degenerate: one class only -> undefined rates are 0, curves stay finite where defined
    y = np.zeros(10, np.int64); p = np.linspace(0, 1, 10).astype(np.float32)
    got = dict(zip(mm.metrics.performance_metrics,
                   mm.metrics.get_performance_metrics(torch.from_numpy(y), torch.from_numpy((p > 0.5).astype(np.int64)), torch.from_numpy(p))))
    assert got["sensitivity"] == 0 and float(got["f1"]) == 0.0


def test_last_epoch_returns_test_results():
    g = Golden("c2_split")
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    model._engine_factory = OracleEngine
    opt = torch.optim.Adam(list(model.parameters()), 0.01)
    res = model.train_epoch(loader_of(g), opt, torch.nn.CrossEntropyLoss(), None, last_epoch=True)
    assert isinstance(res, list) and len(res) == g.spec.D and len(res[0]) == 15


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fused8", "fast8", "seq16"])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_eval_entry_points_match_reference_golden(name, mode, monkeypatch):
    from test_hip_parity import set_mode
    mm.hip.load()
    set_mode(monkeypatch, mode)
    g = Golden(name)
    check_eval_against_golden(g, trained_model(g, "cuda"))


@pytest.mark.gpu
def test_hip_eval_device_nan_policy_and_large_predict(monkeypatch):
    mm.hip.load()
    g = Golden("nan_skip")
    model = trained_model(g, "cuda")
    model.nan_policy = "device"
    check_eval_against_golden(g, model)
    # predict on more rows than any batch so far: the engine re-plans, result equals the oracle
    spec = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.3)
    params = O.init_params(spec, 2)
    xs, _ = O.synthetic_batches(spec, 3000, 3000, seed=9)[0]
    m2 = build_torch_model(spec, params, "cuda", mm)
    pred = m2.predict([torch.from_numpy(x) for x in xs])
    ref = O.predict(params, spec, xs)
    marg = []
    O.predict({n: np.asarray(v, np.float64) for n, v in params.items()}, spec, xs, dtype=np.float64, margins=marg)
    assert_predictions_match(pred, ref, marg[0], "predict() on 3000 rows")     # equal, except where the float64 outputs tie


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2_split", "nan_skip", "seq_perm", "mimic_drop"])
def test_hip_test_collecting_loop_equals_general_loop(name):
    """test() over batches that already live on the device takes its one-call-per-step loop (MultiModN._test_steps_collected,
    mmn_eval_step_ex: the step's last-row outputs and its "row exists" flag are collected by the library call); the result -
    History arrays and every value of the per-decoder report - equals the general loop's bit for bit, also with a last
    batch that is smaller than the others and when the same loader is evaluated again (reused batch structs)."""
    mm.hip.load()
    g = Golden(name)
    loader = []
    for b in g.batches():
        if len(b) > 2:
            pytest.skip("explicit sequences go through the general loop")
        loader.append(([torch.from_numpy(x).cuda() for x in b[0]], torch.from_numpy(b[1]).cuda()))
    xs, y = loader[-1]
    loader.append(([x[:max(1, x.shape[0] // 2)].contiguous() for x in xs], y[:max(1, y.shape[0] // 2)].contiguous()))
    out = []
    for fast in (True, False):
        model = trained_model(g, "cuda")
        model.nan_policy = "device"
        model.collect_in_step = fast
        hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
        res = [model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="test") for _ in range(2)]
        assert ("_eval_bufs" in model._engine.__dict__) == fast          # the collecting loop really ran (or did not)
        out.append((hist, res))
    (h1, r1), (h0, r0) = out
    for k in ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        for a, b in zip(getattr(h1, k)["test"], getattr(h0, k)["test"]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), k
    for ra, rb in zip(r1, r0):
        for da, db in zip(ra, rb):
            for va, vb in zip(da, db):
                assert np.array_equal(np.asarray(va), np.asarray(vb))
