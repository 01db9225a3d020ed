"""The oracle against every golden vector the reference produced (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from helpers import GOLDEN_NAMES, HAIM_GOLDEN_NAMES, MIMIC_GOLDEN_NAMES, PER_SAMPLE_B1_NAMES, Golden, PerSampleGolden, assert_within_fp32_noise, fp64_trajectory, rel_err
from oracle import multimodn_oracle as O


@pytest.mark.parametrize("name", GOLDEN_NAMES + MIMIC_GOLDEN_NAMES + HAIM_GOLDEN_NAMES)
def test_oracle_reproduces_reference_run(name):
    g = Golden(name)
    params = g.init_params()
    opt = O.Adam(g.cfg["lr"])
    z = g.z
    s = 0
    for ep in range(g.epochs):
        results, sizes = [], []
        for bi in range(g.n_batches):
            b = g.batch(bi)
            r = O.forward_backward(params, g.spec, b[0], b[1], b[2] if len(b) > 2 else None,
                                   drop_masks=g.step_masks(s))
            assert abs(r.loss - z["step_loss"][s]) / abs(z["step_loss"][s]) < 2e-6
            r64 = None
            if g.has_step(s):
                none = set(str(x) for x in z[f"step{s}/grad_none"])
                for n, gr in r.grads.items():
                    assert (gr is None) == (n in none), (n, s)
                    if gr is not None:        # 1e-5 of the reference's gradient outright (north_star's bar), or within fp32 noise of float64
                        if r64 is None:
                            r64 = O.forward_backward({k: np.asarray(v, np.float64) for k, v in params.items()}, g.spec, b[0], b[1],
                                                     b[2] if len(b) > 2 else None, dtype=np.float64, drop_masks=g.step_masks(s))
                        assert_within_fp32_noise(gr, z[f"step{s}/grad/{n}"], r64.grads[n], (n, s), tight=1e-5)
            opt.step(params, r.grads)
            results.append(r)
            sizes.append(len(b[1]))
            s += 1
        er = O.aggregate_epoch(g.spec.E, g.spec.D, results, sizes)
        assert rel_err(er.loss, z["hist/loss"][ep]) < 2e-6
        assert rel_err(er.state_change, z["hist/state_change"][ep]) < 2e-6
        for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
            assert np.array_equal(getattr(er, k), z["hist/" + k][ep]), k
        assert er.loss.dtype == np.float64 and er.sensitivity.dtype == np.float32
    w64 = fp64_trajectory(g)[0]                          # 2e-5 of the reference's weights, or within fp32 noise of the fp64 replay
    for n, w in g.final_params().items():
        assert_within_fp32_noise(params[n], w, w64[n], n)


@pytest.mark.parametrize("name", PER_SAMPLE_B1_NAMES)
def test_per_sample_oracle_reproduces_the_reference_at_batch_size_one(name):
    """The per-sample extension pinned to a reference run: N samples fed to the reference one per batch (multimodn.py:167-171,
    509-531 at batch size 1, as pipelines/titanic/titanic_missingness_pipeline.py:35 runs it) against oracle.per_sample_step
    on the N-row batch - loss cells, state change, loss, executed rows, mean gradients, and the epoch's History ratios."""
    g = PerSampleGolden(name)
    r = O.per_sample_step(g.init_params(), g.spec, g.xs, g.y, g.seq, drop_masks=g.masks)
    g.check(r.err_loss, r.state_change, r.loss, r.row_counts, r.grads, tol=2e-6, tolg=1e-5)
    er = O.aggregate_epoch(g.spec.E, g.spec.D, [r], [g.N])
    for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        assert np.array_equal(getattr(er, k), g.z["hist/" + k]), k
    # three samples' own gradients: the oracle at batch size 1 IS the reference's step
    for b in (0, 1, 2):
        rb = O.forward_backward(g.init_params(), g.spec, [x[b:b + 1] for x in g.xs], g.y[b:b + 1], g.seq[b:b + 1],
                                drop_masks=None if g.masks is None else {e: m[b:b + 1] for e, m in g.masks.items()})
        assert abs(rb.loss - g.z["sample_loss"][b]) / g.z["sample_loss"][b] < 2e-6
        for n, gr in rb.grads.items():
            key = f"sample{b}/grad/{n}"
            assert (gr is None) == (key not in g.z.files), (b, n)
            if gr is not None:
                assert rel_err(gr.reshape(g.z[key].shape), g.z[key]) < 1e-5, (b, n)


def test_fp64_oracle_agrees_with_fp32():
    g = Golden("c3_small")
    b = g.batch(0)
    r32 = O.forward_backward(g.init_params(), g.spec, b[0], b[1])
    r64 = O.forward_backward(g.init_params(), g.spec, b[0], b[1], dtype=np.float64)
    assert abs(r32.loss - r64.loss) / r64.loss < 1e-6
    for n in r32.grads:
        assert rel_err(r32.grads[n], r64.grads[n]) < 1e-5


FD_SPECS = {
    "mlp": O.ModelSpec(6, [O.EncoderSpec(3, (4,), O.ACT_SIGMOID), O.EncoderSpec(2, (), O.ACT_RELU)], 2, 0.8, 50.0),
    # MIMIC family: sigmoid everywhere (smooth, so central differences are valid), explicit dropout masks
    "mimic": O.ModelSpec(6, [O.EncoderSpec(3, (4,), O.ACT_SIGMOID, kind="mimic", dropout=0.3),
                             O.EncoderSpec(2, (), O.ACT_SIGMOID, kind="mimic"),
                             O.EncoderSpec(2, (3,), O.ACT_SIGMOID)], 2, 0.8, 50.0,
                         decoders=[O.DecoderSpec("mlp", (4, 3), O.ACT_SIGMOID), O.DecoderSpec("mlp", ())]),
}


@pytest.mark.parametrize("family", ["mlp", "mimic"])
def test_gradients_by_finite_differences(family):
    spec = FD_SPECS[family]
    params = O.init_params(spec, 0, np.float64)
    xs, y = O.synthetic_batches(spec, 5, 5, seed=0)[0]
    masks = None
    if family == "mimic":
        keep = np.random.default_rng(5).random((5, 3 + 6)) >= 0.3
        masks = {0: keep / 0.7}
    import functools
    fb = functools.partial(O.forward_backward, drop_masks=masks)
    r = fb(params, spec, xs, y, dtype=np.float64)
    rng = np.random.default_rng(0)
    for n in spec.param_names():
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in params[n].shape)
            p1 = {k: v.copy() for k, v in params.items()}
            p2 = {k: v.copy() for k, v in params.items()}
            p1[n][idx] += 1e-6
            p2[n][idx] -= 1e-6
            fd = (fb(p1, spec, xs, y, dtype=np.float64, want_grads=False).loss
                  - fb(p2, spec, xs, y, dtype=np.float64, want_grads=False).loss) / 2e-6
            assert abs(fd - r.grads[n][idx]) < 1e-7 + 1e-5 * abs(fd), (n, idx)


def test_shard_sum_equals_full_batch():
    spec = O.ModelSpec(8, [O.EncoderSpec(3, (4,), O.ACT_RELU), O.EncoderSpec(2, (5, 3), O.ACT_RELU)], 2, 1.0, 0.5)
    params = O.init_params(spec, 1, np.float64)
    xs, y = O.synthetic_batches(spec, 12, 12, seed=2)[0]
    full = O.forward_backward(params, spec, xs, y, dtype=np.float64)
    parts = [O.forward_backward(params, spec, [x[lo:lo + 4] for x in xs], y[lo:lo + 4], batch_global=12,
                                dtype=np.float64) for lo in (0, 4, 8)]
    assert rel_err(sum(p.err_loss for p in parts), full.err_loss) < 1e-12
    for n in full.grads:
        assert rel_err(sum(p.grads[n] for p in parts), full.grads[n]) < 1e-12


def test_per_sample_extension_matches_batch_when_uniform():
    spec = O.ModelSpec(8, [O.EncoderSpec(3, (4,), O.ACT_RELU), O.EncoderSpec(2, (), O.ACT_RELU)], 2, 1.0, 0.5)
    params = O.init_params(spec, 1, np.float64)
    xs, y = O.synthetic_batches(spec, 6, 6, seed=2)[0]
    full = O.forward_backward(params, spec, xs, y, dtype=np.float64)
    ps = O.per_sample_step(params, spec, xs, y, None, dtype=np.float64)
    assert rel_err(ps.err_loss, full.err_loss) < 1e-12
    for n in full.grads:
        assert rel_err(ps.grads[n], full.grads[n]) < 1e-12


def test_sequence_rows_must_agree():
    with pytest.raises(ValueError):
        O.encoder_iterable(2, np.array([[0, 1], [1, 0]]))
