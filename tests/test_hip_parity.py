"""GPU parity tests: the HIP path (through the C ABI) against (a) the golden vectors produced by
the reference itself and (b) the numpy oracle on the same seeded inputs.

Tolerances (fp32 path, BASELINE.json north_star: 1e-5 relative):
  * per-step loss / loss grid / state change ............ 1e-5 relative
  * integer work (n_correct, tp/tn/fp/fn, row counts) .... bit-exact
  * gradients of one step vs reference ................... 2e-5 of the tensor's max |g|
    (the reference's own fp32 grads sit ~3e-7 from fp64 truth; see tests/golden/make_golden.py)
  * History loss / state-change after training ........... 1e-5 relative
  * History accuracy / sensitivity / specificity / balanced accuracy: EQUAL to the reference's (ratios of integer
    counts; helpers.assert_counts_match allows a count to move only by as many predictions as tie within 1e-6 in the
    fp64 replay - none in any golden)
  * trained weights ....................................... 2e-5 of the tensor's max |w| against the reference's, or -
    where Adam has amplified rounding-size gradient differences beyond that (it divides by sqrt(v)) - no further from
    the fp64 replay of the same run than 4x the reference's own fp32 run is (helpers.assert_within_fp32_noise)
"""
import numpy as np
import pytest
import torch

from helpers import (GOLDEN_NAMES, Golden, assert_counts_match, assert_within_fp32_noise, build_torch_model,
                     fp64_trajectory, rel_err)
from oracle import multimodn_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import multimodn_amd
    multimodn_amd.hip.load()          # fail loudly if the HIP library did not travel
    assert torch.cuda.is_available()
    return multimodn_amd


def to_dev(batch, device="cuda"):
    xs = [torch.from_numpy(np.ascontiguousarray(x)).to(device) for x in batch[0]]
    y = torch.from_numpy(np.ascontiguousarray(batch[1])).to(device)
    seq = batch[2] if len(batch) > 2 else None
    return xs, y, seq


def run_step(model, batch, batch_global=None, nan_policy="host"):
    """One engine step on a numpy batch; returns (stats dict, grads dict name->np or None)."""
    model.nan_policy = nan_policy
    data = [torch.from_numpy(np.ascontiguousarray(x)) for x in batch[0]]
    target = torch.from_numpy(np.ascontiguousarray(batch[1]))
    seq = torch.from_numpy(batch[2]) if len(batch) > 2 else None
    eng = model._get_engine(target.shape[0])
    eng.epoch_reset()
    executed, keep = model._run_step(eng, data, target, seq, train=True, batch_global=batch_global)
    eng.assign_grads(executed)
    torch.cuda.synchronize()
    stats = {k: np.array(v) for k, v in eng.step_values().items()}
    grads = {n: (None if p.grad is None else p.grad.detach().cpu().numpy().copy())
             for n, p in model.named_parameters()}
    return stats, grads, executed


def check_against(stats, grads, ref: O.StepResult, tol_loss=1e-5, tol_grad=2e-5, truth=None):
    assert rel_err(stats["loss"], ref.loss) < tol_loss
    assert rel_err(stats["err_loss"], ref.err_loss) < tol_loss
    if np.abs(ref.state_change).max() > 0:
        assert rel_err(stats["state_change"], ref.state_change) < tol_loss
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    for n, g in ref.grads.items():
        if g is None:
            assert grads[n] is None, n
        else:
            assert grads[n] is not None, n
            assert rel_err(grads[n].reshape(g.shape), g) < tol_grad, (n, rel_err(grads[n].reshape(g.shape), g))


@pytest.mark.parametrize("mode", ["fused8", "fused8p", "fast8", "seq16", "gen16"])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_first_step_matches_reference_golden(lib, name, mode, monkeypatch):
    set_mode(monkeypatch, mode)
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    stats, grads, _ = run_step(model, g.batch(0))
    assert rel_err(stats["loss"], g.z["step_loss"][0]) < 1e-5
    if g.has_step(0):
        ref_grads = g.step_grads(0)
        none = set(str(s) for s in g.z["step0/grad_none"])
        for n in g.spec.param_names():
            if n in none:
                assert grads[n] is None
            else:
                assert rel_err(grads[n], ref_grads[n]) < 2e-5, n
    # and against the oracle (fp32) for everything the golden file does not hold per step
    b = g.batch(0)
    ref = O.forward_backward(g.init_params(), g.spec, b[0], b[1], b[2] if len(b) > 2 else None)
    check_against(stats, grads, ref)


@pytest.mark.parametrize("optimizer", ["torch", "hip"])
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_training_matches_reference_golden(lib, name, optimizer):
    """Whole train_epoch loops through the public surface: History arrays and trained weights,
    with torch's Adam (what the reference pipelines build) and with the one-launch HIP Adam."""
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    Adam = torch.optim.Adam if optimizer == "torch" else lib.optim.Adam
    opt = Adam(list(model.parameters()), g.cfg["lr"])
    hist = lib.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    crit = torch.nn.CrossEntropyLoss()
    loader = []
    for b in g.batches():
        item = [[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]
        if len(b) > 2:
            item.append(torch.from_numpy(b[2]))
        loader.append(tuple(item))
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, crit, hist)
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < 1e-5
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 1e-5
    acc = np.stack(hist.accuracy["train"])
    assert acc.dtype == np.float64 and hist.sensitivity["train"][0].dtype == np.float32
    # counts are integers: a prediction can only flip when its two sigmoids tie (none does in the goldens)
    assert_counts_match(hist, z, g)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    # trained weights: within 2e-5 of the reference's, or as close to the fp64 replay of the run as the reference's own
    # fp32 run is (x4): Adam divides by sqrt(v), so rounding-size gradient differences on near-zero-gradient
    # coordinates become O(lr) weight differences in ANY fp32 implementation
    w64, l64, s64 = fp64_trajectory(g)[:3]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n], w, w64[n], n)
    assert_within_fp32_noise(np.stack(hist.loss["train"]), z["hist/loss"], l64, "History loss", tight=1e-5)


# mode -> (MMN_FAST8, MMN_FUSED, MMN_GENERIC)
KERNEL_MODES = {"fused8": ("1", "1", "0"), "fused8p": ("1", "1", "0"),      # fused8p: the fused kernel's pair-per-encoder form
                "fast8": ("1", "0", "0"),                                    # (MMN_FB8_LEAN=0), what wider shapes run
                "seq16": ("0", "0", "0"),
                # the generic tier (k_gen_fwd / k_gen_bwd: written for MIMIC_MLPEncoder / MLPDecoder models) forced
                # onto MLPEncoder + LogisticDecoder models: same results through different kernels and plan tables
                "gen16": ("0", "0", "1")}


def set_mode(monkeypatch, mode):
    """Kernel tier, read at plan creation: the fused 8-wave kernel (k_fb9, or k_fb8 for the shapes k_fb9 does not take),
    the two-launch 8-wave tier (k_fwd8 / k_bwd8), the sequential chain kernels, or the generic tier's sequential form
    (16-row tiles everywhere)."""
    fast8, fused, generic = KERNEL_MODES[mode]
    monkeypatch.setenv("MMN_GENERIC", generic)
    monkeypatch.setenv("MMN_FAST8", fast8)
    monkeypatch.setenv("MMN_FUSED", fused)          # forward+backward chain in one launch (E <= 4)
    monkeypatch.setenv("MMN_FB8_LEAN", "0" if mode == "fused8p" else "1")


@pytest.mark.parametrize("mode", list(KERNEL_MODES))
@pytest.mark.parametrize("B", [1, 15, 16, 17, 31, 32, 33, 257])
def test_ragged_batches_match_oracle(lib, B, mode, monkeypatch):
    set_mode(monkeypatch, mode)
    spec = O.ModelSpec(20, [O.EncoderSpec(7, (9, 6), O.ACT_RELU), O.EncoderSpec(3, (), O.ACT_RELU),
                            O.EncoderSpec(70, (33,), O.ACT_SIGMOID)], 3, 1.0, 0.7)
    params = O.init_params(spec, 3)
    batch = O.synthetic_batches(spec, B, B, seed=11)[0]
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref = O.forward_backward(params, spec, batch[0], batch[1])
    check_against(stats, grads, ref)


@pytest.mark.parametrize("mode", ["fast8", "seq16"])
@pytest.mark.parametrize("B", [1, 9, 16, 23, 130])
def test_mimic_like_shapes_ragged(lib, B, mode, monkeypatch):
    """Shapes inside the 8-wave tier's envelope: mixed hidden depths (0, 1, 2), a permuted sequence."""
    set_mode(monkeypatch, mode)
    spec = O.ModelSpec(48, [O.EncoderSpec(32, (32, 16), O.ACT_RELU), O.EncoderSpec(16, (), O.ACT_RELU),
                            O.EncoderSpec(64, (32,), O.ACT_SIGMOID), O.EncoderSpec(128, (16, 32), O.ACT_RELU),
                            O.EncoderSpec(8, (32, 32), O.ACT_IDENTITY)], 4, 1.0, 0.6)
    params = O.init_params(spec, 9)
    xs, y = O.synthetic_batches(spec, B, B, seed=4)[0]
    order = [3, 0, 4, 1, 2]
    batch = ([xs[e] for e in order], y, np.tile(np.array(order, np.int64), (B, 1)))
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref = O.forward_backward(params, spec, batch[0], batch[1], batch[2])
    check_against(stats, grads, ref)


@pytest.mark.parametrize("mode", ["fused8", "fast8"])
@pytest.mark.parametrize("B", [1, 9, 16, 23, 130])
def test_fused_chain_shapes_ragged(lib, B, mode, monkeypatch):
    """E <= 4 so that the fused forward+backward kernel applies: hidden depths 2, 0, 1, 2."""
    set_mode(monkeypatch, mode)
    spec = O.ModelSpec(48, [O.EncoderSpec(32, (32, 16), O.ACT_RELU), O.EncoderSpec(16, (), O.ACT_RELU),
                            O.EncoderSpec(64, (32,), O.ACT_SIGMOID), O.EncoderSpec(128, (16, 32), O.ACT_IDENTITY)],
                       4, 1.0, 0.6)
    params = O.init_params(spec, 9)
    xs, y = O.synthetic_batches(spec, B, B, seed=4)[0]
    order = [3, 0, 1, 2]
    batch = ([xs[e] for e in order], y, np.tile(np.array(order, np.int64), (B, 1)))
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref = O.forward_backward(params, spec, batch[0], batch[1], batch[2])
    check_against(stats, grads, ref)


def test_unaligned_feature_stride_and_big_dims(lib):
    # widest shapes this round's LDS budget admits (state / hidden <= 128), odd feature strides
    spec = O.ModelSpec(72, [O.EncoderSpec(130, (100, 17), O.ACT_RELU), O.EncoderSpec(1, (128,), O.ACT_RELU)],
                       8, 0.9, 0.4)
    params = O.init_params(spec, 5)
    batch = O.synthetic_batches(spec, 70, 70, seed=2)[0]
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref = O.forward_backward(params, spec, batch[0], batch[1])
    check_against(stats, grads, ref)


def c3_spec():
    return O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.3)


@pytest.mark.parametrize("mode", list(KERNEL_MODES))
def test_full_size_c3_step_matches_oracle(lib, mode, monkeypatch):
    set_mode(monkeypatch, mode)
    spec = c3_spec()
    params = O.init_params(spec, 0)
    batch = O.synthetic_batches(spec, 4096, 4096, seed=1)[0]
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref32 = O.forward_backward(params, spec, batch[0], batch[1])
    ref64 = O.forward_backward(params, spec, batch[0], batch[1], dtype=np.float64)
    # 4096-term sums that cancel to ~1e-4 (bias grads) move by ~5e-5 with the summation order
    # alone, so fp32-vs-fp32 gets a looser gradient bound here and fp64 truth is the real judge:
    check_against(stats, grads, ref32, tol_grad=3e-4)
    # the HIP path must be as close to fp64 truth as the fp32 CPU restatement is (x4 slack)
    for n, g in ref64.grads.items():
        e_hip = rel_err(grads[n].reshape(g.shape), g)
        e_cpu = rel_err(ref32.grads[n], g)
        assert e_hip < max(4 * e_cpu, 2e-6), (n, e_hip, e_cpu)
    assert rel_err(stats["err_loss"], ref64.err_loss) < 2e-6


def test_full_size_c3_five_adam_steps_within_fp32_noise(lib):
    """Trained weights at the BASELINE.json size (configs[2]: batch 4096, 5 fused Adam steps): every tensor sits no further
    from the float64 trajectory than 4x the fp32 numpy oracle does (or agrees with it to 2e-5 outright), tensor by tensor;
    tensors whose gradient crossed a relu kink differently than the float64 replay did are reported, not compared
    (helpers.full_size_trajectories).  The state-update layers, the decoders and the init state have no relu in front of
    them: they are always compared."""
    from helpers import full_size_trajectories
    spec = c3_spec()
    params = O.init_params(spec, 0)
    batches = O.synthetic_batches(spec, 5 * 4096, 4096, seed=11)
    model, p32, p64, flipped = full_size_trajectories(lib, spec, params, batches, 1e-3)
    always = [n for n in spec.param_names() if ".layers.2." in n or n.startswith("decoders.") or n.startswith("init_state.")]
    assert not (flipped & set(always))
    compared = 0
    for n, p in model.named_parameters():
        if n in flipped:
            continue
        assert_within_fp32_noise(p.detach().cpu().numpy(), p32[n], p64[n], n)
        compared += 1
    assert compared >= len(always)
    print(f"compared {compared} tensors, {len(flipped)} crossed a relu kink: {sorted(flipped)}")


@pytest.mark.parametrize("launch", ["step_path", "shipping_default"])
@pytest.mark.parametrize("which", ["c1", "c2"])
def test_titanic_configs_at_their_stated_sizes(lib, which, launch, monkeypatch):
    """BASELINE.json configs[0] / configs[1] at the sizes they state, through the public train_epoch: C1 = the Titanic MLP
    pipeline's shape (pipelines/titanic/titanic_mlp_pipeline.py:63-74: 570 training rows, batch 32 -> 17 full batches and
    one of 26 rows, E = 1, F = 6, H = (5, 5), S = 32, D = 1, Adam 0.01, penalties 0.7 / 0.3) for one whole epoch = 18 Adam
    steps; C2 = two encoders over the [3, 2] feature split (titanic_partitioned_pipeline.py:26-27), S = 64, D = 2, batch 512,
    5 Adam steps.  Against the numpy oracle's train_epoch on the same batches: History loss / state change 1e-5, the
    count ratios equal wherever the float64 replay has no near-tie, trained weights within fp32 noise of the float64
    trajectory (assert_within_fp32_noise).
    launch "shipping_default" (VERDICT r5): MMN_EPOCH_KERNEL as a user's process has it - unset - and the batches where a
    DeviceResidentLoader keeps them: C1 then runs as ONE launch of k_epoch_small (checked), C2 (512 rows: outside that kernel)
    the step path with replayed groups; "step_path": what the rest of this suite pins (tests/conftest.py)."""
    if launch == "shipping_default":
        monkeypatch.delenv("MMN_EPOCH_KERNEL", raising=False)
    if which == "c1":
        spec = O.ModelSpec(32, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3)
        rows, B, lr = 570, 32, 0.01
    else:
        spec = O.ModelSpec(64, [O.EncoderSpec(3, (5, 5), O.ACT_RELU), O.EncoderSpec(2, (5, 5), O.ACT_RELU)], 2, 0.7, 0.3)
        rows, B, lr = 5 * 512, 512, 0.01
    params = O.init_params(spec, 3)
    batches = O.synthetic_batches(spec, rows, B, seed=17)
    assert len(batches) == (18 if which == "c1" else 5) and len(batches[-1][1]) == (26 if which == "c1" else 512)
    model = build_torch_model(spec, params, "cuda", lib)
    opt = lib.optim.Adam(model.parameters(), lr=lr)
    hist = lib.MultiModNHistory([f"t{d}" for d in range(spec.D)])
    put = (lambda a: torch.from_numpy(a).cuda()) if launch == "shipping_default" else torch.from_numpy
    loader = [([put(x) for x in xs], put(y)) for xs, y in batches]
    model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    torch.cuda.synchronize()
    if launch == "shipping_default":
        assert bool(model.__dict__.get("_small_epochs")) == (which == "c1"), "C1 takes k_epoch_small by default, C2 does not fit it"
    p32 = {n: np.asarray(v, np.float32).copy() for n, v in params.items()}
    p64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    e32 = O.train_epoch(p32, spec, batches, O.Adam(lr))
    e64 = O.train_epoch(p64, spec, batches, O.Adam(lr), dtype=np.float64)
    assert rel_err(hist.loss["train"][0], e64.loss) < 1e-5 and rel_err(hist.loss["train"][0], e32.loss) < 1e-5
    assert rel_err(hist.state_change_loss[0], e64.state_change) < 1e-5
    if np.array_equal(e32.accuracy, e64.accuracy):          # (no prediction sits on a tie: the counts are exact)
        for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
            assert np.array_equal(getattr(hist, k)["train"][0], getattr(e32, k)), k
    for n, p in model.named_parameters():
        assert_within_fp32_noise(p.detach().cpu().numpy(), p32[n], p64[n], (which, n))


def test_shard_linearity_full_size(lib):
    """Size-independent property used by data parallelism: with batch_global fixed, the reduce
    buffer of the full batch equals the sum of the shards' buffers (grads and statistics)."""
    spec = c3_spec()
    params = O.init_params(spec, 7)
    batch = O.synthetic_batches(spec, 4096, 4096, seed=5)[0]
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    full = model._engine.reduce_buf.detach().cpu().numpy().copy()
    acc = np.zeros_like(full, dtype=np.float64)
    for lo in range(0, 4096, 1024):
        shard = ([x[lo:lo + 1024] for x in batch[0]], batch[1][lo:lo + 1024])
        run_step(model, shard, batch_global=4096)
        acc += model._engine.reduce_buf.detach().cpu().numpy()
    n = model._engine.n_params
    assert rel_err(acc[:n], full[:n]) < 1e-5
    st_full, st_sum = full[n:], acc[n:]
    k = model._engine.n_stats - 4 - (spec.E + 1)          # grid, sc and counters are additive
    assert rel_err(st_sum[:k], st_full[:k]) < 1e-5


def test_device_nan_policy_matches_host_policy(lib):
    g = Golden("nan_skip")
    b = g.batch(1)                                         # the batch that holds a NaN
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    s_host, g_host, ex = run_step(model, b, nan_policy="host")
    assert ex == [True, False, True]
    s_dev, g_dev, _ = run_step(model, b, nan_policy="device")
    assert np.array_equal(s_host["err_loss"], s_dev["err_loss"])
    assert np.array_equal(s_host["rows"], s_dev["rows"])
    for n in g_host:
        if g_host[n] is None:
            assert not np.any(g_dev[n])                    # device policy: zeros instead of None
        else:
            assert np.array_equal(g_host[n], g_dev[n])


def test_determinism(lib):
    spec = c3_spec()
    params = O.init_params(spec, 1)
    batch = O.synthetic_batches(spec, 1000, 1000, seed=3)[0]
    model = build_torch_model(spec, params, "cuda", lib)
    run_step(model, batch)
    a = model._engine.reduce_buf.detach().cpu().numpy().copy()
    run_step(model, batch)
    b = model._engine.reduce_buf.detach().cpu().numpy()
    assert np.array_equal(a, b)


def test_eval_step_matches_oracle_forward(lib):
    g = Golden("c2_split")
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    hist = lib.MultiModNHistory(["a", "b"])
    loader = [([torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])) for b in g.batches()]
    model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="val")
    spec_eval = O.ModelSpec(g.spec.state_size, g.spec.encoders, g.spec.D, 1.0, 0.0)
    res = [O.forward_backward(g.init_params(), spec_eval, b[0], b[1], want_grads=False) for b in g.batches()]
    ep = O.aggregate_epoch(g.spec.E, g.spec.D, res, [len(b[1]) for b in g.batches()])
    assert rel_err(hist.loss["val"][0], ep.loss) < 1e-5
    assert np.array_equal(hist.accuracy["val"][0], ep.accuracy)
    assert np.array_equal(hist.sensitivity["val"][0], ep.sensitivity)


def test_no_silent_fallback_off_gpu(lib):
    spec = O.ModelSpec(8, [O.EncoderSpec(4, (5,), O.ACT_RELU)], 1, 1.0, 0.0)
    model = build_torch_model(spec, O.init_params(spec, 0), "cpu", lib)
    loader = [([torch.zeros(4, 4)], torch.zeros(4, 1, dtype=torch.int64))]
    with pytest.raises(lib.hip.MmnError):
        model.train_epoch(loader, torch.optim.Adam(model.parameters()), torch.nn.CrossEntropyLoss())


def test_device_resident_loader_equals_host_batches(lib):
    """DeviceResidentLoader (data feed in HBM): same batches, same History and weights
    as a host-side list of the same batches; a PartitionDataset goes in without a per-sample loop."""
    g = Golden("c2_split")
    batches = g.batches()
    X = np.concatenate([np.concatenate(b[0], axis=1) for b in batches], axis=0)
    y = np.concatenate([b[1] for b in batches], axis=0)
    ds = lib.PartitionDataset(X, y, [x.shape[1] for x in batches[0][0]])
    B = batches[0][1].shape[0]
    dev_loader = lib.DeviceResidentLoader(ds, B, device="cuda")
    assert len(dev_loader) == len(batches)
    first = next(iter(dev_loader))
    assert first[0][0].is_cuda and first[1].dtype == torch.int64 and tuple(first[1].shape) == batches[0][1].shape
    host_loader = [([torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])) for b in batches]
    out = []
    for loader, policy in ((host_loader, "host"), (dev_loader, "device"), (dev_loader, "host")):
        model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
        model.nan_policy = policy
        opt = lib.optim.Adam(list(model.parameters()), g.cfg["lr"])
        hist = lib.MultiModNHistory(["a", "b"])
        for _ in range(2):
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        out.append((np.stack(hist.loss["train"]), {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}))
    for loss, params in out[1:]:
        assert np.array_equal(loss, out[0][0])
        for n in params:
            assert np.array_equal(params[n], out[0][1][n]), n
    # shuffled epochs: a permutation of the rows, reproducible from the generator
    sh = lib.DeviceResidentLoader(ds, B, shuffle=True, device="cuda", generator=torch.Generator().manual_seed(5))
    rows = torch.cat([b[1] for b in sh], dim=0).cpu().numpy()
    assert rows.shape == y.reshape(len(y), -1).shape and rows.sum() == y.sum()


@pytest.mark.parametrize("pipeline", ["mlp", "featurewise", "missingness"])
@pytest.mark.parametrize("device_loader", [False, True])
def test_reference_pipeline_shape_runs_end_to_end(lib, device_loader, pipeline):
    """examples/titanic_like_pipeline.py = the reference's Titanic MLP pipeline body with the import
    swapped: stock DataLoader over a PartitionDataset (per-sample tensors, default collate), train +
    val every epoch, pickling.  The loss must go down and the report must be sane."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "titanic_like_pipeline", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "examples", "titanic_like_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # (--featurewise / --missingness: the bodies of titanic_featurewise_pipeline.py / titanic_missingness_pipeline.py - six
    # MLPFeatureEncoder over a FeatureWiseDataset; the latter at batch size 1 with missing values kept as NaN)
    hist, results = mod.main(["--epochs", "12", "--quiet"] + (["--device-loader"] if device_loader else [])
                             + ([] if pipeline == "mlp" else ["--" + pipeline]))
    tr = np.stack(hist.loss["train"])
    rows = 2 if pipeline == "mlp" else 7
    assert tr.shape == (12, rows, 1) and np.stack(hist.loss["val"]).shape == (12, rows, 1)
    # the decoder on the last state learns (missingness: on the state behind 'Sex_male', which every passenger has - a History
    # row of a feature that is often missing averages in a 0 for every batch that skipped it, multimodn.py:236)
    row = 4 if pipeline == "missingness" else -1
    assert np.isfinite(tr).all() and tr[-1, row, 0] < tr[0, row, 0] - 0.02
    if pipeline != "missingness":                      # (there the report covers only the ~25 passengers with a known last feature)
        assert 0.6 < float(results[0][1]) <= 1.0       # AUC of the validation report


@pytest.mark.parametrize("feed", ["stock_loader", "device_loader"])
@pytest.mark.parametrize("name", ["titanic_featurewise", "titanic_missingness"])
def test_featurewise_pipelines_match_reference_golden(lib, name, feed, monkeypatch):
    """The reference's feature-wise Titanic pipelines (titanic_featurewise_pipeline.py: five MLPFeatureEncoder(5, 5), batch
    32; titanic_missingness_pipeline.py: six, batch size 1, missing values kept as NaN so that a sample's missing feature
    skips that encoder) written with this package's classes (tests/helpers.py::featurewise_pipeline), against the reference's
    own run of them.  stock_loader: FeatureWiseDataset -> torch DataLoader -> torch.optim.Adam, as the pipelines build them;
    device_loader: DeviceResidentLoader + multimodn_amd.optim.Adam with MMN_EPOCH_KERNEL as a user's process has it - the
    whole epoch is then one launch of k_epoch_small (checked), the missing features decided on the device."""
    from helpers import featurewise_pipeline
    g = Golden(name)
    model, loader = featurewise_pipeline(g, "cuda", lib)
    if feed == "device_loader":
        monkeypatch.delenv("MMN_EPOCH_KERNEL", raising=False)
        loader = lib.DeviceResidentLoader(loader.dataset, g.cfg["B"], device="cuda")
        opt = lib.optim.Adam(model.parameters(), lr=g.cfg["lr"])
    else:
        opt = torch.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = lib.MultiModNHistory(["Survived"])
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    torch.cuda.synchronize()
    if feed == "device_loader":
        assert model.__dict__.get("_small_epochs"), "a feature-wise Titanic epoch on device-resident batches is one k_epoch_small launch"
    z = g.z
    w64, l64, s64 = fp64_trajectory(g)[:3]
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 1e-5
    assert_within_fp32_noise(np.stack(hist.loss["train"]), z["hist/loss"], l64, "History loss", tight=1e-5)
    assert_counts_match(hist, z, g)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n], w, w64[n], n)
    th = lib.MultiModNHistory(["Survived"])
    res = model.test(loader, torch.nn.CrossEntropyLoss(), th)
    assert rel_err(th.loss["test"][0], z["eval/test_loss"]) < 2e-5
    assert np.array_equal(th.accuracy["test"][0], z["eval/test_accuracy"])
    n_last = sum(len(b[1]) for b in g.batches() if not np.isnan(b[0][-1]).any())
    tn, fp, fn, tp = (int(res[0][lib.metrics.performance_metrics.index(k)]) for k in ("tn", "fp", "fn", "tp"))
    assert tn + fp + fn + tp == n_last                 # the report covers the batches whose last encoder ran


def test_integration_md_stub_runs(lib):
    """The ctypes stub printed in INTEGRATION.md (what a reference maintainer would paste) is executed
    verbatim - only the library path is made absolute - on a model with the reference's attribute
    names, and its step is checked against the oracle."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libmmn_hip.so")', f'C.CDLL(r"{lib.hip.LIB_PATH}")')
    ns = {}
    exec(code, ns)
    spec = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.3)
    params = O.init_params(spec, 1)
    model = build_torch_model(spec, params, "cuda", lib)
    xs, y = O.synthetic_batches(spec, 256, 256, seed=2)[0]
    plan, ws, flat_g = ns["build_plan"](model, 256)
    data = [torch.from_numpy(x).cuda() for x in xs]
    target = torch.from_numpy(y).cuda()
    ns["step"](plan, data, target, [(k, k) for k in range(4)], 1.0, 0.003)
    torch.cuda.synchronize()
    ref = O.forward_backward(params, spec, xs, y)
    n_params = sum(p.numel() for p in model.parameters())
    stats = flat_g[n_params:].cpu().numpy()
    assert rel_err(stats[:15].reshape(5, 3), ref.err_loss) < 1e-5
    # a gradient element is a sum of per-sample terms: its fp32 rounding noise scales with the TERMS, not with the sum (a
    # decoder bias gradient is ~1000 terms of either sign that cancel to 1e-5 of their size), so the tolerance is 2e-5 of
    # the tensor's own largest element or 2e-6 of the model's largest gradient element, whichever is larger
    g_all = max(float(np.max(np.abs(g))) for g in ref.grads.values() if g is not None)
    for n, p in model.named_parameters():
        got, want = p.grad.cpu().numpy().reshape(ref.grads[n].shape), ref.grads[n]
        assert float(np.max(np.abs(got - want))) <= max(2e-5 * float(np.max(np.abs(want))), 2e-6 * g_all), n


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_shapes_against_oracle(lib, seed):
    """Seeded sweep over model shapes and batch sizes: whatever kernel tier the library picks for the
    shape (fused / 8-wave / 4-wave parallel / sequential) must reproduce the oracle's step."""
    rng = np.random.default_rng(1000 + seed)
    E = int(rng.integers(1, 9))
    S = int(rng.choice([4, 8, 12, 16, 32, 48, 64, 100, 128]))
    D = int(rng.integers(1, 5))
    B = int(rng.choice([1, 7, 16, 33, 100, 257]))
    hid_choices = [(), (8,), (16, 16), (32,), (5, 5), (32, 32), (7,), (24, 8)]
    encs = []
    for _ in range(E):
        H = hid_choices[int(rng.integers(0, len(hid_choices)))]
        act = int(rng.choice([O.ACT_RELU, O.ACT_SIGMOID])) if H else O.ACT_IDENTITY
        encs.append(O.EncoderSpec(int(rng.choice([1, 3, 4, 6, 16, 33, 64, 100])), H, act))
    spec = O.ModelSpec(S, encs, D, float(rng.choice([0.7, 1.0])), float(rng.choice([0.0, 0.3, 1.0])))
    params = O.init_params(spec, seed)
    xs, y = O.synthetic_batches(spec, B, B, seed=seed + 5)[0]
    order = rng.permutation(E)
    batch = ([xs[e] for e in order], y, np.tile(order.astype(np.int64), (B, 1)))
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch)
    ref = O.forward_backward(params, spec, batch[0], batch[1], batch[2])
    # (a gradient can be a near-total cancellation of per-sample terms ~1e-3: the fp32 noise of that sum,
    #  ~1e-8 absolute, does not shrink with the result - hence the absolute floor next to 3e-5 of max|g|)
    assert rel_err(stats["loss"], ref.loss) < 1e-5
    assert rel_err(stats["err_loss"], ref.err_loss) < 1e-5
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    for n, g in ref.grads.items():
        got = grads[n].reshape(g.shape)
        assert np.abs(got - g).max() <= 3e-5 * np.abs(g).max() + 2e-8, (n, np.abs(got - g).max(), np.abs(g).max())
