"""GPU parity of the MIMIC family: MIMIC_MLPEncoder (multimodn/encoders/mlp_encoder.py:9-47)
and MLPDecoder (multimodn/decoders/decoders.py:22-46) through the C ABI (k_gen_fwd / k_gen_bwd + k_wgrad +
k_reduce) against the golden vectors the reference produced with these modules (tests/golden/make_golden.py,
dropout masks recorded from the reference's own nn.Dropout draws) and against the numpy oracle.

Tolerances as in tests/test_hip_parity.py: loss / grid 1e-5 relative, integer counters exact, gradients 2e-5 of
the tensor's max |g| (5e-5 at batch 4096 against the fp64 oracle: 4096-term sums in another order), trained
weights 2e-5 of max |w| or within 4x the reference's own distance from the fp64 replay (helpers.assert_within_fp32_noise)."""
import sys

import numpy as np
import pytest
import torch

from helpers import (HAIM_GOLDEN_NAMES, MIMIC_GOLDEN_NAMES, Golden, assert_counts_match, assert_within_fp32_noise, build_torch_model,
                     fp64_trajectory, rel_err)
from oracle import multimodn_oracle as O
from test_hip_parity import check_against, lib  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def mask_provider(masks):
    """masks: {encoder id: numpy [B, F+S]} -> what MultiModN.dropout_mask_provider expects."""
    def provide(e, batch, width):
        m = masks.get(e)
        return None if m is None else torch.from_numpy(np.ascontiguousarray(m, np.float32))
    return provide


def run_step(model, batch, masks=None, batch_global=None, nan_policy="host", device_inputs=False):
    """device_inputs: every data slot its own device tensor (the host path packs a batch's slots into one staging buffer;
    under tests/efence a slot of its own stands flush against unmapped memory)."""
    model.nan_policy = nan_policy
    model.dropout_mask_provider = mask_provider(masks or {})
    data = [torch.from_numpy(np.ascontiguousarray(x)) for x in batch[0]]
    target = torch.from_numpy(np.ascontiguousarray(batch[1]))
    if device_inputs:
        data = [x.cuda() for x in data]
        target = target.cuda()
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


@pytest.mark.parametrize("form", ["split", "split_2rows", "split_genf2", "batched", "sequential"])
@pytest.mark.parametrize("name", MIMIC_GOLDEN_NAMES)
def test_first_step_matches_reference_golden(lib, name, form, monkeypatch):
    """The forms of the generic tier.  "split": the default - at the MIMIC pipelines' encoder shape (mimic_c3_small) the chain is
    k_mfwd / k_mbwd, the decoders run in k_dec_fb; "split_2rows": k_dec_fb's two-grid-rows-per-workgroup form (round 5: measured
    slower, opt-in, MMN_DEC_FB_ROWS=2); "split_genf2": the same split with k_genf2_fwd / k_genf2_bwd kept as the
    chain (MMN_MC=0); "batched": k_genf2_* with the decoders inside; "sequential": k_gen_* (what every model the batched
    forms do not take runs)."""
    monkeypatch.setenv("MMN_GEN_FAST", "0" if form == "sequential" else "1")
    monkeypatch.setenv("MMN_GEN_BATCHED", "1" if form in ("batched", "split", "split_2rows", "split_genf2") else "0")   # decoders of all grid rows at once (16-row tiles)
    monkeypatch.setenv("MMN_GEN_SPLIT", "1" if form in ("split", "split_2rows", "split_genf2") else "0")        # ... in a launch of their own (k_dec_fb)
    monkeypatch.setenv("MMN_DEC_FB_ROWS", "2" if form == "split_2rows" else "1")
    monkeypatch.setenv("MMN_MC", "0" if form == "split_genf2" else "1")
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    stats, grads, _ = run_step(model, g.batch(0), g.step_masks(0))
    assert rel_err(stats["loss"], g.z["step_loss"][0]) < 1e-5
    ref_grads = g.step_grads(0)
    none = set(str(s) for s in g.z["step0/grad_none"])
    for n in g.spec.param_names():
        if n in none:
            assert grads[n] is None
        else:
            assert rel_err(grads[n], ref_grads[n]) < 2e-5, (n, rel_err(grads[n], ref_grads[n]))
    b = g.batch(0)
    ref = O.forward_backward(g.init_params(), g.spec, b[0], b[1], b[2] if len(b) > 2 else None, drop_masks=g.step_masks(0))
    check_against(stats, grads, ref)


@pytest.mark.parametrize("form", ["default", "sequential", "decoders_inside", "x_part_inside"])
@pytest.mark.parametrize("name", HAIM_GOLDEN_NAMES)
def test_haim_shape_first_step_matches_reference_golden(lib, name, form, monkeypatch):
    """The reference's real MIMIC configuration: state 50, hidden (32, 32), dropout 0.2, batch 16, sources of
    width [6, 1024, 768, 99] (the pipeline's four) / all nine of the dataset - through whatever kernels the plan picks for the
    shape ("default": k_xpart + the sequential chain kernels + k_dec_fb since round 6), through the sequential form forced
    (k_gen_*), and through the forms the defaults replaced: the decoders inside the chain kernels (MMN_SEQ_SPLIT=0), the
    first layers' x parts inside them (MMN_XPART=0)."""
    if form == "sequential":
        monkeypatch.setenv("MMN_GEN_FAST", "0")
    if form == "decoders_inside":
        monkeypatch.setenv("MMN_SEQ_SPLIT", "0")
    if form == "x_part_inside":
        monkeypatch.setenv("MMN_XPART", "0")
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    stats, grads, _ = run_step(model, g.batch(0), g.step_masks(0))
    assert rel_err(stats["loss"], g.z["step_loss"][0]) < 1e-5
    ref_grads = g.step_grads(0)
    for n in g.spec.param_names():
        assert rel_err(grads[n], ref_grads[n]) < 2e-5, (n, rel_err(grads[n], ref_grads[n]))
    b = g.batch(0)
    ref = O.forward_backward(g.init_params(), g.spec, b[0], b[1], None, drop_masks=g.step_masks(0))
    check_against(stats, grads, ref)


@pytest.mark.parametrize("optimizer", ["torch", "hip"])
@pytest.mark.parametrize("name", MIMIC_GOLDEN_NAMES + HAIM_GOLDEN_NAMES)
def test_training_matches_reference_golden(lib, name, optimizer):
    """Whole train_epoch loops through the public surface with the masks the reference drew."""
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    Adam = torch.optim.Adam if optimizer == "torch" else lib.optim.Adam
    opt = Adam(list(model.parameters()), g.cfg["lr"])
    hist = lib.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    crit = torch.nn.CrossEntropyLoss()
    loader = [tuple([[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]) for b in g.batches()]

    def provide(e, batch, width):
        m = g.step_masks(model.train_steps_launched).get(e)
        return None if m is None else torch.from_numpy(m)


    model.dropout_mask_provider = provide
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, crit, hist)
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < 1e-5
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 1e-5
    assert_counts_match(hist, z, g)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    assert list(sd.keys()) == [str(n) for n in z["param_names"]]          # the reference's state_dict keys, in order
    w64 = fp64_trajectory(g)[0]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n], w, w64[n], n)


@pytest.mark.parametrize("name", MIMIC_GOLDEN_NAMES + HAIM_GOLDEN_NAMES)
def test_eval_entry_points_match_reference(lib, name):
    """test() History, predict() and get_states() on the reference's TRAINED weights (eval mode: no dropout)."""
    g = Golden(name)
    model = build_torch_model(g.spec, {n: w.copy() for n, w in g.final_params().items()}, "cuda", lib)
    z = g.z
    loader = [tuple([[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]) for b in g.batches()]
    hist = lib.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="test")
    assert rel_err(hist.loss["test"][0], z["eval/test_loss"]) < 1e-5
    assert np.abs(hist.accuracy["test"][0] - z["eval/test_accuracy"]).max() <= 1.0 / g.cfg["B"] + 1e-12
    states = torch.stack(model.get_states(loader)).cpu().numpy()
    assert rel_err(states, z["eval/states"]) < 1e-5
    if "eval/predict" in z.files:
        pred = model.predict(loader[0][0])
        assert (pred != z["eval/predict"]).mean() < 0.01               # a class can flip only on a rounding-level tie


def test_replayed_steps_with_host_batches(lib):
    """The same with HOST batches (a stock DataLoader's tensors): they are staged through a ring of three device
    buffers, whose addresses repeat every third batch with NEW content - a replayed step must read the fresh copy."""
    spec = O.ModelSpec(32, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3)
    params = O.init_params(spec, 3)
    batches = O.synthetic_batches(spec, 7 * 32, 32, seed=10)
    loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in batches]      # CPU tensors
    runs = {}
    for replay in (True, False):
        model = build_torch_model(spec, params, "cuda", lib)
        model.replay_steps = replay                          # nan_policy "auto" + the HIP Adam: device-side decision
        opt = lib.optim.Adam(list(model.parameters()), 1e-2)
        hist = lib.MultiModNHistory(["a"])
        for _ in range(3):
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        torch.cuda.synchronize()
        if replay:
            assert any(v[1] is not None for v in model._engine._step_graphs.values())
        runs[replay] = (np.stack(hist.loss["train"]), {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})
    assert np.array_equal(runs[True][0], runs[False][0])
    for k in runs[True][1]:
        assert np.array_equal(runs[True][1][k], runs[False][1][k]), k
    ref_p = {k: v.copy() for k, v in params.items()}
    oopt = O.Adam(1e-2)
    for _ in range(3):
        ep = O.train_epoch(ref_p, spec, batches, oopt)
    assert rel_err(runs[True][0][-1], ep.loss) < 1e-5          # and it is the oracle's trajectory


@pytest.mark.parametrize("family", ["classic", "mimic"])
def test_replayed_steps_equal_eager_steps(lib, family):
    """train_epoch with nan_policy "device" + multimodn_amd.optim.Adam captures a step into a hipGraph the second time
    it sees the same device buffers and replays it afterwards (engine.run_group).  Four epochs over the
    same device-resident batches, with replay on and off: History and trained weights must be identical bit for bit
    (deterministic kernels; Adam counters, epoch sums and the dropout draw index live on the device)."""
    if family == "mimic":
        spec = O.ModelSpec(32, [O.EncoderSpec(12, (16, 16), O.ACT_RELU, kind="mimic", dropout=0.2) for _ in range(3)], 2, 1.0, 0.3,
                           decoders=[O.DecoderSpec("mlp", (16,)) for _ in range(2)])
    else:
        spec = O.ModelSpec(32, [O.EncoderSpec(12, (16, 16), O.ACT_RELU) for _ in range(3)], 2, 1.0, 0.3)
    params = O.init_params(spec, 2)
    batches = O.synthetic_batches(spec, 5 * 48, 48, seed=9)
    loader = [([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda()) for xs, y in batches]
    runs = {}
    for replay in (True, False):
        torch.manual_seed(77)
        model = build_torch_model(spec, params, "cuda", lib)
        model.nan_policy = "device"
        model.replay_steps = replay
        opt = lib.optim.Adam(list(model.parameters()), 1e-2)
        hist = lib.MultiModNHistory(["a", "b"])
        for _ in range(4):
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        torch.cuda.synchronize()
        n_graphs = sum(1 for v in model._engine._step_graphs.values() if v[1] is not None)
        assert (n_graphs >= 1) == replay                    # (the five steps of an epoch are one group)
        runs[replay] = (np.stack(hist.loss["train"]), np.stack(hist.state_change_loss),
                        {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()},
                        {k: v["step"].item() for k, v in list(opt.state.items())[:1]})
    assert np.array_equal(runs[True][0], runs[False][0]) and np.array_equal(runs[True][1], runs[False][1])
    for k in runs[True][2]:
        assert np.array_equal(runs[True][2][k], runs[False][2][k]), k
    assert list(runs[True][3].values()) == list(runs[False][3].values()) == [20.0]
    assert runs[True][0][-1].mean() < runs[True][0][0].mean()


def test_mimic_pipeline_shape_runs_end_to_end(lib):
    """examples/mimic_like_pipeline.py = the body of the reference's MIMIC multi-task pipeline with the import swapped
    (stock DataLoader over Subsets of a PartitionDataset, MIMIC_MLPEncoder / MLPDecoder, dropout live in training,
    train_epoch(last_epoch=True), per-epoch validation report, best-checkpoint save / load, pickling, plot)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mimic_like_pipeline.py")
    spec = importlib.util.spec_from_file_location("mimic_like_pipeline", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist, train_report, rows, best_epoch = mod.main(["--epochs", "4", "--rows", "2048", "--batch-size", "64", "--quiet"])
    tr, va = np.stack(hist.loss["train"]), np.stack(hist.loss["val"])
    assert tr.shape == (4, 5, 3) and va.shape == (4, 5, 3)
    assert tr[-1, -1].mean() < tr[0, -1].mean() - 0.01              # the heads on the last state learn
    assert len(train_report) == 3 and len(train_report[0]) == 15      # last_epoch=True returns test()'s report on the train set
    assert 1 <= best_epoch <= 4
    for r in rows:
        assert 0.6 < float(r[2]) <= 1.0                               # test-split AUC on the best checkpoint


def test_reference_written_checkpoint_evaluates_on_hip(lib):
    """SURVEY 8f #4 on the GPU: load the reference-written checkpoint fixture, run test() on the HIP path."""
    import os
    from helpers import GOLDEN
    g = Golden("mimic_drop")
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    ck = torch.load(os.path.join(GOLDEN, "ref_checkpoint_mimic_drop.pt"), map_location="cuda")
    model.load_state_dict(ck["model_state_dict"], strict=True)
    loader = [tuple([[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]) for b in g.batches()]
    hist = lib.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="test")
    assert rel_err(hist.loss["test"][0], g.z["eval/test_loss"]) < 1e-5
    assert rel_err(torch.stack(model.get_states(loader)).cpu().numpy(), g.z["eval/states"]) < 1e-5


@pytest.mark.parametrize("B", [1, 15, 17, 33, 257])
def test_ragged_batches_match_oracle(lib, B):
    spec = O.ModelSpec(20, [O.EncoderSpec(7, (9, 6), O.ACT_RELU, kind="mimic", dropout=0.3),
                            O.EncoderSpec(3, (), O.ACT_SIGMOID, kind="mimic"),
                            O.EncoderSpec(70, (33,), O.ACT_SIGMOID),
                            O.EncoderSpec(150, (40,), O.ACT_RELU, kind="mimic", dropout=0.1)], 3, 1.0, 0.7,
                       decoders=[O.DecoderSpec("mlp", (12, 5)), O.DecoderSpec("class"), O.DecoderSpec("mlp", (7,), O.ACT_SIGMOID)])
    params = O.init_params(spec, 3)
    batch = O.synthetic_batches(spec, B, B, seed=11)[0]
    rng = np.random.default_rng(B)
    masks = {0: ((rng.random((B, 27)) >= 0.3) / 0.7).astype(np.float32),
             3: ((rng.random((B, 170)) >= 0.1) / 0.9).astype(np.float32)}
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch, masks)
    ref = O.forward_backward(params, spec, batch[0], batch[1], drop_masks=masks)
    check_against(stats, grads, ref)


def mimic_c3_spec(pen=(1.0, 0.3)):
    """The MIMIC pipelines' modules at BASELINE's MIMIC shape (mimic_multi_task_pipeline.py:118-119)."""
    return O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU, kind="mimic", dropout=0.2) for _ in range(4)], 3,
                       pen[0], pen[1], decoders=[O.DecoderSpec("mlp", (32, 32)) for _ in range(3)])


@pytest.mark.parametrize("B", [4096, 6000])
def test_full_size_against_fp64_oracle(lib, B):
    spec = mimic_c3_spec()
    params = O.init_params(spec, 5)
    batch = O.synthetic_batches(spec, B, B, seed=2)[0]
    rng = np.random.default_rng(7)
    masks = {e: ((rng.random((B, 192)) >= 0.2) / 0.8).astype(np.float32) for e in range(4)}
    model = build_torch_model(spec, params, "cuda", lib)
    stats, grads, _ = run_step(model, batch, masks)
    ref = O.forward_backward(params, spec, batch[0], batch[1], drop_masks=masks, dtype=np.float64)
    check_against(stats, grads, ref, tol_grad=5e-5)


def test_kernel_names_and_forms(lib, monkeypatch):
    """Which kernels run: the fast form for an all-MIMIC model within its limits, the sequential form otherwise."""
    import ctypes as C
    fast = build_torch_model(mimic_c3_spec(), O.init_params(mimic_c3_spec(), 0), "cuda", lib)
    eng = fast._get_engine(64)
    b = eng.make_batch([torch.zeros(64, 64, device="cuda") for _ in range(4)], torch.zeros(64, 3, dtype=torch.int64, device="cuda"),
                       [(k, k) for k in range(4)])
    assert eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(b), 0) == b"k_mfwd"          # the MIMIC pipelines' encoder shape: the 8-wave chain
    assert eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(b), 1) == b"k_mbwd"          # kernels with every operand in registers (round 4)
    assert eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(b), 2) == b""
    assert eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(b), 3) == b"k_dec_fb"        # the decoders' own launch (split form)
    wide = O.ModelSpec(128, [O.EncoderSpec(64, (64, 32), O.ACT_RELU, kind="mimic", dropout=0.2) for _ in range(4)], 3, 1.0, 0.3,
                       decoders=[O.DecoderSpec("mlp", (32, 32)) for _ in range(3)])       # a 64-wide hidden layer: outside k_mfwd's shapes
    eng1 = build_torch_model(wide, O.init_params(wide, 0), "cuda", lib)._get_engine(64)
    assert eng1.lib.mmn_chain_kernel_name(eng1._plan, C.byref(b), 0) in (b"k_genf2_fwd", b"k_gen_fwd")    # the batched form, or the sequential
    assert eng1.lib.mmn_chain_kernel_name(eng1._plan, C.byref(b), 1) in (b"k_genf2_bwd", b"k_gen_bwd")   # one where its LDS carve does not fit
    g = Golden("mimic_mixed")                               # one MLPEncoder among the MIMIC ones: sequential form
    mixed = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    eng2 = mixed._get_engine(16)
    xs, y, _ = [torch.from_numpy(x).cuda() for x in g.batch(0)[0]], torch.from_numpy(g.batch(0)[1]).cuda(), None
    b2 = eng2.make_batch(xs, y, [(k, k) for k in range(3)])
    assert eng2.lib.mmn_chain_kernel_name(eng2._plan, C.byref(b2), 0) == b"k_gen_fwd"


def test_shard_sum_equals_full_batch(lib):
    """Size-independent property at full size: four 1024-row shards with the global divisor sum to the full batch."""
    spec = mimic_c3_spec()
    params = O.init_params(spec, 6)
    B = 4096
    batch = O.synthetic_batches(spec, B, B, seed=3)[0]
    rng = np.random.default_rng(8)
    masks = {e: ((rng.random((B, 192)) >= 0.2) / 0.8).astype(np.float32) for e in range(4)}
    model = build_torch_model(spec, params, "cuda", lib)
    full_stats, full_grads, _ = run_step(model, batch, masks)
    acc = None
    for lo in range(0, B, 1024):
        sl = slice(lo, lo + 1024)
        st, gr, _ = run_step(model, ([x[sl] for x in batch[0]], batch[1][sl]), {e: m[sl] for e, m in masks.items()},
                             batch_global=B)
        if acc is None:
            acc = (st["err_loss"].copy(), {n: g.copy() for n, g in gr.items()})
        else:
            acc = (acc[0] + st["err_loss"], {n: acc[1][n] + g for n, g in gr.items()})
    assert rel_err(acc[0], full_stats["err_loss"]) < 1e-5
    for n, g in full_grads.items():
        assert rel_err(acc[1][n], g) < 2e-5, n


def philox4x32_10(counter, key):
    """numpy restatement of the published Philox4x32-10 (Salmon et al., SC'11): counter [n, 4] uint32, key (k0, k1)."""
    c = counter.astype(np.uint64).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = (p1 >> np.uint64(32)) ^ c[:, 1] ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c[:, 3] ^ k1
        c = np.stack([n0 & MASK, p1 & MASK, n2 & MASK, p0 & MASK], axis=1)
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c.astype(np.uint32)


def expected_masks(seed, draw, sizes, ps):
    """What k_dropout writes (include/mmn_hip.h, mmn_draw_dropout): encoder after encoder, group g of 4 floats from
    counter (g, 0, draw lo, draw hi)."""
    total4 = sum((n + 3) // 4 for n in sizes)
    ctr = np.zeros((total4, 4), np.uint32)
    ctr[:, 0] = np.arange(total4, dtype=np.uint32)
    ctr[:, 2], ctr[:, 3] = draw & 0xFFFFFFFF, draw >> 32
    u = (philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    out, g0 = [], 0
    for n, p in zip(sizes, ps):
        g1 = g0 + (n + 3) // 4
        m = np.where(u[g0:g1].reshape(-1) >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0))
        out.append(m[:n].astype(np.float32))
        g0 = g1
    return out


def test_device_dropout_draw(lib):
    """Without a provider the multipliers come from k_dropout: bit-exact against a numpy Philox4x32-10, restarted by
    torch.manual_seed, advancing by one draw per launch (also inside a replayed hipGraph), right keep rate, and the
    training step on THOSE multipliers (read back from the buffer) equals the oracle's."""
    spec = mimic_c3_spec()
    params = O.init_params(spec, 1)
    model = build_torch_model(spec, params, "cuda", lib)
    B = 2048
    batch = O.synthetic_batches(spec, B, B, seed=4)[0]
    data = [torch.from_numpy(x) for x in batch[0]]
    target = torch.from_numpy(batch[1])
    eng = model._get_engine(B)

    def step():
        eng.epoch_reset()
        executed, keep = model._run_step(eng, data, target, None, train=True)
        eng.assign_grads(executed)
        torch.cuda.synchronize()
        stats = {k: np.array(v) for k, v in eng.step_values().items()}
        grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
        return stats, grads, [m.cpu().numpy().copy() for m in keep[2]]

    torch.manual_seed(123)
    s1, g1, m1 = step()
    s2, g2, m2 = step()
    torch.manual_seed(124)
    s3, _, m3 = step()
    torch.manual_seed(123)
    s4, _, m4 = step()
    assert len(m1) == 4 and all(m.shape == (B, 192) for m in m1)
    sizes, ps = [B * 192] * 4, [0.2] * 4
    for draw, got, seed in ((0, m1, 123), (1, m2, 123), (0, m3, 124), (0, m4, 123)):
        want = expected_masks(seed, draw, sizes, ps)
        for a, w in zip(got, want):
            assert np.array_equal(a.reshape(-1), w)
    assert float(s1["loss"]) == float(s4["loss"]) and float(s1["loss"]) != float(s2["loss"]) != float(s3["loss"])
    for mk in m1:
        assert abs(float((mk > 0).mean()) - 0.8) < 0.01 and set(np.unique(mk).tolist()) <= {0.0, 1.25}
    # the step that ran on the first draw against the oracle on the same multipliers
    # (relu on every layer: a pre-activation within rounding of zero switches a unit for one sample, which moves a
    #  gradient by ~1/B of its scale whatever the precision - the fp32 numpy oracle itself sits 4.5e-3 from fp64 on
    #  this case; the HIP path's fixed-order fp32 FMA chains measured 1e-4)
    ref = O.forward_backward(params, spec, batch[0], batch[1], drop_masks={e: m1[e] for e in range(4)}, dtype=np.float64)
    check_against(s1, g1, ref, tol_grad=5e-4)
    # graph replay: every replay of a captured draw advances the draw index on the device
    eng.reset_dropout()
    b = eng.make_batch([d.cuda() for d in data], target.cuda(), [(k, k) for k in range(4)])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.draw_dropout_masks(b)                                    # warm-up outside capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            views = eng.draw_dropout_masks(b)
    torch.cuda.current_stream().wait_stream(side)
    seen = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        seen.append(views[0].cpu().numpy().copy())
    want = [expected_masks(123, d, sizes, ps)[0].reshape(B, 192) for d in (1, 2, 3)]     # draw 0 was the warm-up
    for a, w in zip(seen, want):
        assert np.array_equal(a, w)
    del graph, views                                         # release the capture's private pool before the engine goes
    torch.cuda.synchronize()


def test_dropout_drawn_by_the_previous_step(lib):
    """Look-ahead: a step's last launch (k_reduce) also draws the NEXT step's multipliers (mmn_step_opts.next_drop_p) and
    the next step adopts them without a launch (mmn_dropout_adopt).  They must be the multipliers a k_dropout launch in
    front of that step draws - same generator, same draw index, bit for bit - the index must advance once per consumed
    draw, and a pre-drawn set nobody adopts must not advance it."""
    spec = mimic_c3_spec()
    model = build_torch_model(spec, O.init_params(spec, 3), "cuda", lib)
    B = 272                                                   # 17 tiles
    eng = model._get_engine(B)
    bs = []
    for seed in (11, 12):
        xs, y = O.synthetic_batches(spec, B, B, seed=seed)[0]
        keep = ([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda())
        bs.append((eng.make_batch(keep[0], keep[1], [(k, k) for k in range(4)], device_nan_flags=True), keep))
    (b1, _k1), (b2, _k2) = bs
    sizes, ps = [B * 192] * 4, [0.2] * 4

    def same(views, draw):
        torch.cuda.synchronize()
        for a, w in zip(views, expected_masks(321, draw, sizes, ps)):
            assert np.array_equal(a.cpu().numpy().reshape(-1), w)

    torch.manual_seed(321)
    eng.begin_sequence()
    same(eng.draw_dropout_masks(b1), 0)                       # k_dropout, draw 0
    eng.local_step(b1, 1.0, 0.003, accumulate=True, next_batch=b2, predraw_next=True)
    assert eng._predrawn is b2
    same(eng.draw_dropout_masks(b2), 1)                       # adopted: drawn by b1's k_reduce with draw index 1
    assert eng._predrawn is None
    eng.local_step(b2, 1.0, 0.003, accumulate=True)
    same(eng.draw_dropout_masks(b1), 2)                       # k_dropout again: the adopted draw advanced the index
    eng.local_step(b1, 1.0, 0.003, accumulate=True, next_batch=b2, predraw_next=True)
    same(eng.draw_dropout_masks(b1), 3)                       # b2's pre-drawn set is dropped: b1 draws with the next index
    eng.local_step(b1, 1.0, 0.003, accumulate=True)
    same(eng.draw_dropout_masks(b2), 4)
    eng.local_step(b2, 1.0, 0.003, accumulate=True)
    torch.cuda.synchronize()


def _per_sample_case(spec, B, seed, p_missing=0.3):
    """Synthetic per-sample batch (BASELINE configs[4]): missing-not-at-random NaN rows, a random encoder order per sample."""
    rng = np.random.default_rng(seed)
    xs, y = O.synthetic_batches(spec, B, B, seed=seed + 1)[0]
    pm = np.where(y[:, :1] == 1, 1.5 * p_missing, 0.5 * p_missing)
    miss = rng.random((B, spec.E)) < pm
    xs = [x.copy() for x in xs]
    for e in range(spec.E):
        xs[e][miss[:, e]] = np.nan
    seq = np.stack([rng.permutation(spec.E) for _ in range(B)]).astype(np.int64)
    masks = {e: ((rng.random((B, enc.n_features + spec.state_size)) >= enc.dropout) / (1.0 - enc.dropout)).astype(np.float32)
             for e, enc in enumerate(spec.encoders) if enc.kind == "mimic" and enc.dropout > 0}
    return xs, y, seq, masks


@pytest.mark.parametrize("B", [37, 4096])
@pytest.mark.parametrize("family", ["mimic", "mixed"])
def test_per_sample_mode_of_the_mimic_modules(lib, family, B):
    """BASELINE configs[4] with the modules the reference's MNAR pipeline builds (MIMIC_MLPEncoder + MLPDecoder,
    pipelines/mimic/mimic_single_task_mnar_missingness_pipeline.py:163-165): per-sample missing modalities and encoder order
    (the pipelines' shape at batch 4096: k_mfwd / k_dec_fb / k_mbwd on the regrouped 16-row tiles; the small and the mixed
    models: the generic tier's sequential form on the same tiles), dropout multipliers of the reference's kind handed in
    per ORIGINAL row, against the oracle's sample-by-sample loop: loss cells 1e-5, exact row counts and counters,
    gradients 2e-5 of float64 truth."""
    if family == "mimic":
        F = 64 if B == 4096 else 12
        S = 128 if B == 4096 else 32
        H = (32, 32) if B == 4096 else (16,)
        spec = O.ModelSpec(S, [O.EncoderSpec(F, H, O.ACT_RELU, kind="mimic", dropout=0.2) for _ in range(4)], 3, 1.0, 0.3,
                           decoders=[O.DecoderSpec("mlp", (32, 32) if B == 4096 else (16,)) for _ in range(3)])
    else:
        spec = O.ModelSpec(32, [O.EncoderSpec(12, (16,), O.ACT_RELU, kind="mimic", dropout=0.25), O.EncoderSpec(12, (8, 8), O.ACT_RELU),
                                O.EncoderSpec(12, (16, 16), O.ACT_RELU, kind="mimic", dropout=0.0)],
                           2, 1.0, 0.3, decoders=[O.DecoderSpec("mlp", (16, 8)), O.DecoderSpec()])
    params = O.init_params(spec, 3)
    xs, y, seq, masks = _per_sample_case(spec, B, seed=21)
    model = build_torch_model(spec, params, "cuda", lib)
    model.per_sample = True
    model.train()
    model.dropout_mask_provider = mask_provider(masks)
    eng = model._get_engine(B)
    eng.epoch_reset()
    model._run_step_per_sample(eng, [torch.from_numpy(x) for x in xs], torch.from_numpy(y), torch.from_numpy(seq))
    eng.assign_grads(None)
    torch.cuda.synchronize()
    stats = {k: np.array(v) for k, v in eng.step_values().items()}
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    ref = O.per_sample_step(params, spec, xs, y, seq, drop_masks=masks)
    p64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    ref64 = O.per_sample_step(p64, spec, xs, y, seq, dtype=np.float64, drop_masks=masks)
    # (against float64: the fp32 oracle ADDS 4096 per-sample results one by one and is itself 3e-5 off at this size)
    assert rel_err(stats["err_loss"], ref64.err_loss) < 1e-5 and rel_err(stats["state_change"], ref64.state_change) < 1e-5
    assert np.array_equal(stats["rows"].astype(np.int64), ref.row_counts)
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    g_all = max(float(np.max(np.abs(g))) for g in ref64.grads.values() if g is not None)
    for n, g in ref64.grads.items():
        got = grads[n].reshape(np.asarray(params[n]).shape)
        if g is None:
            assert np.abs(got).max() == 0.0, n
        else:
            assert float(np.max(np.abs(got - g))) <= max(2e-5 * float(np.max(np.abs(g))), 2e-6 * g_all), n


def _per_sample_step_against_oracle(lib, spec, B, seed, expect_kernel=None, with_order=True):
    import ctypes as C
    params = O.init_params(spec, 3)
    xs, y, seq, masks = _per_sample_case(spec, B, seed=seed)
    if not with_order:                                      # (modalities of different widths: slot k can only feed encoder k)
        seq = None
    model = build_torch_model(spec, params, "cuda", lib)
    model.per_sample = True
    model.train()
    model.dropout_mask_provider = mask_provider(masks)
    eng = model._get_engine(B)
    eng.epoch_reset()
    if expect_kernel is not None:
        bout, keep = eng.per_sample_batch([torch.from_numpy(x).cuda() for x in xs], torch.from_numpy(y).cuda(), None if seq is None else torch.from_numpy(seq))
        names = tuple(eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(bout), k) for k in (0, 3, 1))
        assert names == expect_kernel, names
    model._run_step_per_sample(eng, [torch.from_numpy(x) for x in xs], torch.from_numpy(y), None if seq is None else torch.from_numpy(seq))
    eng.assign_grads(None)
    torch.cuda.synchronize()
    stats = {k: np.array(v) for k, v in eng.step_values().items()}
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    ref = O.per_sample_step(params, spec, xs, y, seq, drop_masks=masks)
    p64 = {n: np.asarray(v, np.float64) for n, v in params.items()}
    ref64 = O.per_sample_step(p64, spec, xs, y, seq, dtype=np.float64, drop_masks=masks)
    assert rel_err(stats["err_loss"], ref64.err_loss) < 1e-5 and rel_err(stats["state_change"], ref64.state_change) < 1e-5
    assert np.array_equal(stats["rows"].astype(np.int64), ref.row_counts)
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    g_all = max(float(np.max(np.abs(g))) for g in ref64.grads.values() if g is not None)
    for n, g in ref64.grads.items():
        got = grads[n].reshape(np.asarray(params[n]).shape)
        if g is None:
            assert np.abs(got).max() == 0.0, n
        else:
            assert float(np.max(np.abs(got - g))) <= max(2e-5 * float(np.max(np.abs(g))), 2e-6 * g_all), n
    return stats, grads


@pytest.mark.parametrize("form", ["tiles", "sequential"])
@pytest.mark.parametrize("shape", ["pipeline", "odd_widths"])
@pytest.mark.parametrize("B", [5, 37, 300])
def test_per_sample_mimic_pipeline_shapes_on_the_chain_kernels(lib, B, shape, form, monkeypatch):
    """Round 4: per-sample batches of the MIMIC pipelines' shape class run k_mfwd / k_dec_fb / k_mbwd on the regrouped tiles
    (slot k = the tile's k-th executed encoder; zeros for what a tile does not execute and for its padding rows) instead
    of the sequential k_gen_* form.  Small batches: most tiles are short and most of the tile list is padding.  Both forms
    against the oracle's sample-by-sample loop, and against each other."""
    if shape == "pipeline":
        spec = mimic_c3_spec()
    else:
        spec = O.ModelSpec(128, [O.EncoderSpec(24, h, O.ACT_RELU, kind="mimic", dropout=d)
                                 for h, d in (((20, 24), 0.25), ((32, 28), 0.0), ((28, 32), 0.1))], 2, 1.0, 0.3,
                           decoders=[O.DecoderSpec("mlp", (16,)), O.DecoderSpec("mlp", (24, 8))])
    if form == "sequential":
        monkeypatch.setenv("MMN_MC_TILED", "0")
    expect = (b"k_mfwd", b"k_dec_fb", b"k_mbwd") if form == "tiles" else (b"k_gen_fwd", b"k_dec_fb", b"k_gen_bwd")   # (round 6: the decoders' own launch beside the sequential chain too)
    _per_sample_step_against_oracle(lib, spec, B, seed=40 + B, expect_kernel=expect)


@pytest.mark.parametrize("B", [16, 100])
def test_per_sample_step_of_the_real_mimic_shape(lib, B):
    """The reference's real MIMIC configuration (state 50, sources of 6 / 1024 / 768 / 99 features, hidden (32, 32), MLPDecoder
    (32, 32): pipelines/mimic/mimic_multi_task_pipeline.py:53-83) in per-sample mode - round 6: k_xpart forms the x parts of
    the regrouped tiles too, the decoders run in k_dec_fb beside the sequential chain kernels (a state size that is no
    multiple of 4), short encoders take the chain kernels' request-everything-first steps - against the oracle's loop."""
    spec = O.ModelSpec(50, [O.EncoderSpec(f, (32, 32), O.ACT_RELU, kind="mimic", dropout=0.2) for f in (6, 1024, 768, 99)], 2, 1.0, 0.3,
                       decoders=[O.DecoderSpec("mlp", (32, 32)) for _ in range(2)])
    _per_sample_step_against_oracle(lib, spec, B, seed=70 + B, expect_kernel=(b"k_gen_fwd", b"k_dec_fb", b"k_gen_bwd"), with_order=False)


def test_per_sample_mimic_step_matches_the_reference_run_at_batch_size_one(lib):
    """tests/golden/per_sample_b1_mimic.npz: the reference's MIMIC_MLPEncoder + MLPDecoder model fed 32 samples one per batch
    (own encoder order, NaN rows, its recorded dropout masks, frozen weights) against the HIP per-sample step on the 32-row batch."""
    from helpers import PerSampleGolden
    g = PerSampleGolden("per_sample_b1_mimic")
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    model.per_sample = True
    model.train()
    model.dropout_mask_provider = mask_provider(g.masks)
    eng = model._get_engine(g.N)
    eng.epoch_reset()
    model._run_step_per_sample(eng, [torch.from_numpy(x) for x in g.xs], torch.from_numpy(g.y), torch.from_numpy(g.seq))
    eng.assign_grads(None)
    torch.cuda.synchronize()
    stats = {k: np.array(v) for k, v in eng.step_values().items()}
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    g.check(stats["err_loss"], stats["state_change"], stats["loss"], stats["rows"], grads)
    n = 1.0 + stats["rows"].astype(np.float64)[:, None]
    assert np.array_equal(stats["n_correct"] / n, g.z["hist/accuracy"])


def test_per_sample_training_of_the_mimic_modules(lib):
    """train_epoch in per-sample mode with MIMIC modules (device-drawn dropout, fused Adam): runs, the loss falls, test()
    and predict() agree with each other on the regrouped path."""
    spec = O.ModelSpec(32, [O.EncoderSpec(12, (16,), O.ACT_RELU, kind="mimic", dropout=0.1) for _ in range(3)], 2, 1.0, 0.3,
                       decoders=[O.DecoderSpec("mlp", (16,)) for _ in range(2)])
    xs, y, seq, _ = _per_sample_case(spec, 192, seed=5)
    model = build_torch_model(spec, O.init_params(spec, 1), "cuda", lib)
    model.per_sample = True
    opt = lib.optim.Adam(model.parameters(), lr=1e-2)
    hist = lib.MultiModNHistory(["a", "b"])
    loader = [([torch.from_numpy(x[s:s + 64]) for x in xs], torch.from_numpy(y[s:s + 64]), torch.from_numpy(seq[s:s + 64]))
              for s in range(0, 192, 64)]
    for _ in range(12):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    loss = np.stack(hist.loss["train"])
    assert np.isfinite(loss).all() and loss[-1].mean() < loss[0].mean()
    model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="val")
    assert np.isfinite(hist.loss["val"][0]).all()


import collections
SWEEP_FORMS = collections.Counter()


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_models_against_oracle(lib, seed):
    """Seeded sweep over generic-tier models: encoder kinds (all MIMIC in two thirds of the cases, so that the fast
    form runs with 1, 2 and 3 layers, narrow and 64-wide hidden layers, up to 8 encoders / 8 decoders), decoder kinds
    and depths, dropout on some encoders, a NaN-skipped modality, a permuted encoder sequence, ragged batches."""
    _sweep_case(lib, seed, aligned=False)


@pytest.mark.parametrize("seed", list(range(100, 132)))
def test_random_aligned_models_against_oracle(lib, seed, monkeypatch):
    """The same sweep over all-MIMIC models whose widths are multiples of 4: the shapes the batched backward
    (k_genf2_bwd: every tile by LDS-DMA, whole float4s only) accepts - 1 to 3 layers, 0 to 3 decoder layers, narrow
    tiles (a state of 8 is half a DMA request), ragged batches, skipped modalities, permuted sequences.  Even seeds with
    the decoders in their own launch (k_dec_fb, the default), odd seeds with the decoders inside the chain kernels."""
    monkeypatch.setenv("MMN_GEN_SPLIT", "0" if seed % 2 else "1")
    _sweep_case(lib, seed, aligned=True)


def _sweep_case(lib, seed, aligned, device_inputs=False):
    rng = np.random.default_rng(5000 + seed)
    all_mimic = aligned or seed % 3 != 0
    E = int(rng.integers(1, 6 if aligned else 9))
    S = int(rng.choice([8, 16, 24, 48, 64, 100, 128] if aligned else [4, 8, 16, 24, 48, 64, 100, 128]))
    D = int(rng.integers(1, 9 if all_mimic else 5))
    B = int(rng.choice([1, 7, 16, 33, 100, 257]))
    hid_choices = ([(), (8,), (16, 16), (32,), (32, 32), (24, 8), (64, 64), (64,), (8, 32)] if aligned else
                   [(), (8,), (16, 16), (32,), (5, 5), (32, 32), (7,), (24, 8), (64, 64), (64,)])
    encs = []
    for _ in range(E):
        H = hid_choices[int(rng.integers(0, len(hid_choices)))]
        kind = "mimic" if (all_mimic or rng.random() < 0.6) else "mlp"
        act = int(rng.choice([O.ACT_RELU, O.ACT_SIGMOID])) if (H or kind == "mimic") else O.ACT_IDENTITY
        F = int(rng.choice([4, 8, 16, 32, 64, 100] if aligned else [1, 3, 4, 6, 16, 33, 64, 100]))
        if kind == "mimic" and (F + 15) // 16 + (S + 15) // 16 > 12:
            F = 16                                            # keep some all-MIMIC models inside the fast form's layer-0 limit
        drop = float(rng.choice([0.0, 0.2, 0.5])) if kind == "mimic" else 0.0
        encs.append(O.EncoderSpec(F, H, act, kind=kind, dropout=drop))
    dec_h = [(), (8,), (16, 16), (32, 32), (5,), (12, 7, 9)]
    decs = []
    for _ in range(D):
        if rng.random() < 0.25:
            decs.append(O.DecoderSpec("class"))
        else:
            decs.append(O.DecoderSpec("mlp", dec_h[int(rng.integers(0, len(dec_h)))], int(rng.choice([O.ACT_RELU, O.ACT_SIGMOID]))))
    spec = O.ModelSpec(S, encs, D, float(rng.choice([0.7, 1.0])), float(rng.choice([0.0, 0.3, 1.0])), decoders=decs)
    params = O.init_params(spec, seed)
    xs, y = O.synthetic_batches(spec, B, B, seed=seed + 5)[0]
    order = rng.permutation(E)
    xs_o = [xs[e].copy() for e in order]
    if E > 1 and rng.random() < 0.3:
        xs_o[int(rng.integers(0, E))][int(rng.integers(0, B)), 0] = np.nan      # that modality is skipped for the batch
    batch = (xs_o, y, np.tile(order.astype(np.int64), (B, 1)))
    masks = {e: ((rng.random((B, enc.n_features + S)) >= enc.dropout) / (1 - enc.dropout)).astype(np.float32)
             for e, enc in enumerate(encs) if enc.dropout > 0}
    model = build_torch_model(spec, params, "cuda", lib)
    print(f"[sweep] seed {seed} aligned {int(aligned)}: E {E} S {S} D {D} B {B} order {order.tolist()} encoders "
          f"{[(e.kind, e.n_features, tuple(e.hidden), e.dropout) for e in encs]} decoders {[(d.kind, tuple(d.hidden)) for d in decs]}",
          file=sys.stderr, flush=True)                        # (a fault kills the process: the tail of stderr names the case)
    stats, grads, _ = run_step(model, batch, masks, device_inputs=device_inputs)
    import ctypes as C
    eng = model._get_engine(B)
    probe = eng.make_batch([torch.zeros(B, e.n_features, device="cuda") for e in encs],
                           torch.zeros(B, D, dtype=torch.int64, device="cuda"), [(k, k) for k in range(E)])
    SWEEP_FORMS[eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(probe), 0).decode()] += 1
    SWEEP_FORMS[eng.lib.mmn_chain_kernel_name(eng._plan, C.byref(probe), 1).decode()] += 1
    ref = O.forward_backward(params, spec, batch[0], batch[1], batch[2], drop_masks=masks)
    assert rel_err(stats["loss"], ref.loss) < 1e-5
    assert rel_err(stats["err_loss"], ref.err_loss) < 1e-5
    for k in ("n_correct", "tp", "tn", "fp", "fn"):
        assert np.array_equal(stats[k].astype(np.int64), getattr(ref, k)), k
    for n, g in ref.grads.items():
        if g is None:
            assert grads[n] is None, n
            continue
        got = grads[n].reshape(g.shape)
        assert np.abs(got - g).max() <= 3e-5 * np.abs(g).max() + 2e-8, (n, np.abs(got - g).max(), np.abs(g).max())


def test_sweep_ran_both_forms():
    """The sweep above must have exercised both forms of the generic tier (it runs before this test)."""
    if sum(SWEEP_FORMS.values()) < 48:
        pytest.skip("sweep not run in this session")
    assert SWEEP_FORMS["k_genf2_fwd"] + SWEEP_FORMS["k_mfwd"] >= 8 and SWEEP_FORMS["k_gen_fwd"] >= 8, dict(SWEEP_FORMS)
    assert SWEEP_FORMS["k_genf2_bwd"] + SWEEP_FORMS["k_mbwd"] >= 8 and SWEEP_FORMS["k_gen_bwd"] >= 8, dict(SWEEP_FORMS)
