"""The one-launch epoch kernel (csrc/mmn_epoch_small.inc, mmn_train_epoch_small): `for batch in loader:
forward, loss grid, backward, optimizer.step()` of multimodn/multimodn.py:117-212 for the reference's Titanic-sized
models (pipelines/titanic/titanic_mlp_pipeline.py:63-85) as ONE launch of one workgroup per train_epoch call.

Checked against the REFERENCE's own runs (tests/golden/*.npz: History arrays and trained weights of the reference's
train_epoch loops, same rules as test_hip_parity.py::test_training_matches_reference_golden), against the step-by-step
HIP path on the same batches (History within 1e-5, Adam step counts equal - a NaN batch skips its encoder's tensors),
and for what the surface promises: the path is taken only where it applies, `model.epoch_kernel = False` keeps the
step-by-step path, `.grad` holds the last step's gradients."""
import numpy as np
import pytest
import torch

from helpers import Golden, assert_counts_match, assert_within_fp32_noise, build_torch_model, fp64_trajectory, rel_err
from oracle import multimodn_oracle as O
from test_hip_parity import lib  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

# goldens whose model and batches the kernel takes (c2_split / c3_small: more than 4,096 parameters; seq_perm: its own encoder order)
EPOCH_GOLDENS = ["c1_titanic", "c1_curve20", "nan_skip", "slp_sigmoid", "mlp_sigmoid", "mlp_identity"]


def _device_loader(batches):
    return [([torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in b[0]], torch.from_numpy(b[1]).cuda()) for b in batches]


@pytest.mark.parametrize("name", EPOCH_GOLDENS)
def test_epoch_kernel_matches_reference_golden(lib, name, monkeypatch):
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", lib)
    opt = lib.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = lib.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    loader = _device_loader(g.batches())
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    torch.cuda.synchronize()
    assert model.__dict__.get("_small_epochs"), "the one-launch epoch path did not run"
    assert model.train_steps_launched == g.epochs * len(loader)
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < 1e-5
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 1e-5
    assert_counts_match(hist, z, g)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    w64, l64, _ = fp64_trajectory(g)[:3]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n], w, w64[n], n)
    assert_within_fp32_noise(np.stack(hist.loss["train"]), z["hist/loss"], l64, "History loss", tight=1e-5)


def _run(lib, spec, loader, epochs, use_kernel, lr=1e-2):
    torch.manual_seed(3)
    model = build_torch_model(spec, O.init_params(spec, 4), "cuda", lib)
    model.epoch_kernel = use_kernel
    opt = lib.optim.Adam(list(model.parameters()), lr)
    hist = lib.MultiModNHistory([f"t{d}" for d in range(spec.D)])
    for _ in range(epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    torch.cuda.synchronize()
    steps = {i: int(st["step"]) for i, st in opt.state_dict()["state"].items()}
    grads = [p.grad.detach().cpu().numpy().copy() for p in model.parameters()]
    return model, hist, steps, grads


@pytest.mark.parametrize("shape", ["titanic", "titanic_table", "titanic2", "titanic_s1", "titanic_s20", "two_enc", "deep", "titanic_b50", "two_enc_b64",
                                   "wide_b64", "feature6", "feature8_b1"])
def test_epoch_kernel_equals_step_path(lib, shape, monkeypatch):
    """Same batches through both paths: 28 batches of 32 rows (the last one ragged), NaN batches in the middle that skip
    an encoder (multimodn.py:168: no gradient, no Adam step for its tensors - their step counts stay behind)."""
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    # "titanic": the one-encoder form (round 6: k_epoch_small<.., true>, the step written out by hand - state 32, batches of at
    # most 32 rows); "titanic_table": the same model through the table-driven form (MMN_EPS_ONE=0); "titanic2": the one-encoder
    # form with two decoders, sigmoid hidden layers and other widths; "titanic_b50": 50-row batches - outside the one-encoder
    # form, the table-driven one takes it
    monkeypatch.setenv("MMN_EPS_ONE", "0" if shape == "titanic_table" else "1")
    shape = shape.replace("titanic_table", "titanic")
    spec = {"titanic": O.ModelSpec(32, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3),
            "titanic2": O.ModelSpec(32, [O.EncoderSpec(8, (7, 3), O.ACT_SIGMOID)], 2, 1.0, 1.0),
            # state_size = 1: what pipelines/titanic/titanic_mlp_pipeline.py:37 really sets (BASELINE configs[0] says 32); 20: a state
            # that is neither
            "titanic_s1": O.ModelSpec(1, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3),
            "titanic_s20": O.ModelSpec(20, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 2, 0.7, 0.3),
            "two_enc": O.ModelSpec(24, [O.EncoderSpec(3, (5, 5), O.ACT_RELU), O.EncoderSpec(2, (7,), O.ACT_SIGMOID)], 2, 0.7, 0.3),
            "deep": O.ModelSpec(16, [O.EncoderSpec(9, (8, 6, 4), O.ACT_RELU), O.EncoderSpec(4, (), O.ACT_IDENTITY),
                                     O.EncoderSpec(5, (6,), O.ACT_RELU)], 3, 1.0, 1.0),
            # 64 rows x 48 features = 3,072 x values per batch: more than eight waves fetch at 4 a thread - the kernel's
            # 8-per-thread instantiation (round 6: it replaces the 1024-thread form, the one kernel that had scratch)
            # the reference's feature-wise Titanic pipelines (titanic_featurewise_pipeline.py:70, titanic_missingness_pipeline.py:35,71):
            # one MLPFeatureEncoder(state 5, hidden 5) per feature, six features; eight encoders at batch size 1 = the kernel's most
            "feature6": O.ModelSpec(5, [O.EncoderSpec(1, (5,), O.ACT_RELU) for _ in range(6)], 1, 0.7, 0.3),
            "feature8": O.ModelSpec(5, [O.EncoderSpec(1, (5,), O.ACT_RELU) for _ in range(8)], 1, 0.7, 0.3),
            "wide": O.ModelSpec(16, [O.EncoderSpec(12, (6,), O.ACT_RELU) for _ in range(4)], 2, 0.7, 0.3)}[shape.split("_b")[0]]
    B, nb = (int(shape.split("_b")[1]) if "_b" in shape else 32), 28        # (50: every batch ends inside a 16-row tile; 64: the kernel's largest)
    batches = O.synthetic_batches(spec, nb * B - (11 if B > 11 else 0), B, seed=9)        # (the last batch has 21 rows)
    for bi, slot in ((3, 0), (9, spec.E - 1), (10, spec.E - 1), (nb - 1, 0)):
        batches[bi][0][slot][min(1, B - 1), 0] = np.nan
    loader = _device_loader(batches)
    m_k, h_k, st_k, g_k = _run(lib, spec, loader, 3, True)
    m_s, h_s, st_s, g_s = _run(lib, spec, loader, 3, False)
    assert m_k.__dict__.get("_small_epochs") and not m_s.__dict__.get("_small_epochs")
    assert st_k == st_s and len(set(st_k.values())) > 1, st_k          # skipped tensors lag behind, identically
    for a, b in ((h_k.loss["train"], h_s.loss["train"]), (h_k.state_change_loss, h_s.state_change_loss)):
        assert rel_err(np.stack(a), np.stack(b)) < 1e-5
    for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        assert np.abs(np.stack(getattr(h_k, k)["train"]) - np.stack(getattr(h_s, k)["train"])).max() <= 2.0 / (nb * B)
    # trained weights after 84 Adam steps: against the fp64 replay, no further than 4x the fp32 oracle's own replay is
    p32 = {n: v.copy() for n, v in O.init_params(spec, 4).items()}
    p64 = {n: np.asarray(v, np.float64) for n, v in O.init_params(spec, 4).items()}
    o32, o64 = O.Adam(1e-2), O.Adam(1e-2)
    for _ in range(3):
        O.train_epoch(p32, spec, batches, o32, dtype=np.float32)
        O.train_epoch(p64, spec, batches, o64, dtype=np.float64)
    for n, p in m_k.named_parameters():
        assert_within_fp32_noise(p.detach().cpu().numpy(), p32[n], p64[n], n)
    # .grad = the last step's gradients on both paths (a skipped encoder's: whatever an earlier step left)
    last_skips_enc0 = True
    for (n, _), a, b in zip(m_k.named_parameters(), g_k, g_s):
        if last_skips_enc0 and n.startswith("encoders.0."):
            continue
        assert np.abs(a - b).max() <= 3e-5 * max(np.abs(b).max(), 1e-6) + 2e-8, n


def test_epoch_kernel_scope(lib, monkeypatch):
    """Where it does not apply the step-by-step path runs, silently and with the same results contract: batches above 64
    rows, models outside its shapes (MIMIC modules; more than 4,096 parameters), a stock torch optimizer, host batches."""
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    small = O.ModelSpec(32, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3)
    big = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.3)
    mimic = O.ModelSpec(32, [O.EncoderSpec(6, (8,), O.ACT_RELU, kind="mimic", dropout=0.0)], 1, 1.0, 0.3,
                        decoders=[O.DecoderSpec("mlp", (8,))])

    def took(spec, B, opt_cls=None, host=False):
        batches = O.synthetic_batches(spec, 4 * B, B, seed=2)
        loader = _device_loader(batches) if not host else [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in batches]
        model = build_torch_model(spec, O.init_params(spec, 1), "cuda", lib)
        opt = (opt_cls or lib.optim.Adam)(list(model.parameters()), 1e-2)
        hist = lib.MultiModNHistory([f"t{d}" for d in range(spec.D)])
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
        torch.cuda.synchronize()
        assert np.isfinite(np.stack(hist.loss["train"])).all()
        return bool(model.__dict__.get("_small_epochs"))
    assert took(small, 32)
    assert took(small, 64)
    assert not took(small, 65)
    assert not took(big, 32)
    assert not took(mimic, 16)
    assert not took(small, 32, opt_cls=torch.optim.Adam)
    assert not took(small, 32, host=True)


def test_epoch_kernel_then_eval_and_optimizer_options(lib, monkeypatch):
    """What follows an epoch on the kernel path must see its results: test() right behind train_epoch (the chain kernels'
    weight copies were not touched by the epoch kernel: they are rebuilt), an LR change between epochs, weight decay, and
    four encoders of different depths under three decoders - all against the step-by-step path on the same batches."""
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    spec = O.ModelSpec(20, [O.EncoderSpec(7, (6, 5), O.ACT_RELU), O.EncoderSpec(3, (), O.ACT_IDENTITY),
                            O.EncoderSpec(5, (4, 4, 4), O.ACT_SIGMOID), O.EncoderSpec(2, (9,), O.ACT_RELU)], 3, 0.9, 0.4)
    batches = O.synthetic_batches(spec, 9 * 24 - 5, 24, seed=31)
    batches[4][0][2][3, 1] = np.nan
    loader = _device_loader(batches)

    def run(use_kernel):
        torch.manual_seed(5)
        model = build_torch_model(spec, O.init_params(spec, 6), "cuda", lib)
        model.epoch_kernel = use_kernel
        opt = lib.optim.Adam(list(model.parameters()), 1e-2, weight_decay=1e-3)
        hist = lib.MultiModNHistory([f"t{d}" for d in range(spec.D)])
        reports = []
        for ep in range(4):
            if ep == 2:
                for gp in opt.param_groups:
                    gp["lr"] = 3e-3
            model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
            reports.append(model.test(loader, torch.nn.CrossEntropyLoss(), hist, tag="val"))
        torch.cuda.synchronize()
        return model, hist, reports
    m_k, h_k, r_k = run(True)
    m_s, h_s, r_s = run(False)
    assert m_k.__dict__.get("_small_epochs") and not m_s.__dict__.get("_small_epochs")
    for tag in ("train", "val"):
        assert rel_err(np.stack(h_k.loss[tag]), np.stack(h_s.loss[tag])) < 2e-5, tag
        assert np.abs(np.stack(h_k.accuracy[tag]) - np.stack(h_s.accuracy[tag])).max() <= 2.0 / (9 * 24)
    for (n, a), (_, b) in zip(m_k.named_parameters(), m_s.named_parameters()):
        a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
        assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-3), (n, np.abs(a - b).max(), np.abs(b).max())


def test_epoch_kernel_sees_new_optimizer_buffers_and_edited_batches(lib, monkeypatch):
    """The path records addresses (batch descriptors in device memory): an optimizer whose state was re-loaded (new moment /
    step buffers), a batch whose tensor was replaced in its list, and parameters loaded from a checkpoint must all be seen by
    the next call - same training as the step-by-step path doing the same things."""
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    spec = O.ModelSpec(32, [O.EncoderSpec(6, (5, 5), O.ACT_RELU)], 1, 0.7, 0.3)
    batches = O.synthetic_batches(spec, 6 * 32, 32, seed=77)
    other = O.synthetic_batches(spec, 32, 32, seed=78)[0]

    def run(use_kernel):
        torch.manual_seed(1)
        loader = _device_loader(batches)
        model = build_torch_model(spec, O.init_params(spec, 8), "cuda", lib)
        model.epoch_kernel = use_kernel
        opt = lib.optim.Adam(list(model.parameters()), 1e-2)
        hist = lib.MultiModNHistory(["t0"])
        crit = torch.nn.CrossEntropyLoss()
        for ep in range(6):
            if ep == 2:                                      # new optimizer buffers
                sd = opt.state_dict()
                opt.load_state_dict({"state": {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                                               for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]})
            if ep == 3:                                      # a batch's tensor replaced in place of the old one
                loader[2][0][0] = torch.from_numpy(other[0][0]).cuda()
            if ep == 4:                                      # parameters written from outside (a checkpoint)
                model.load_state_dict({k: v * 0.5 for k, v in model.state_dict().items()})
            model.train_epoch(loader, opt, crit, hist)
        torch.cuda.synchronize()
        return model, np.stack(hist.loss["train"])
    m_k, l_k = run(True)
    m_s, l_s = run(False)
    assert m_k.__dict__.get("_small_epochs") and not m_s.__dict__.get("_small_epochs")
    assert rel_err(l_k, l_s) < 2e-5, (l_k, l_s)
    for (n, a), (_, b) in zip(m_k.named_parameters(), m_s.named_parameters()):
        a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
        assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-3), n
    # the reference's pipelines pickle the whole model (titanic_mlp_pipeline.py:96): the path's descriptor cache stays behind
    import copy
    import pickle
    m2 = pickle.loads(pickle.dumps(m_k))
    m3 = copy.deepcopy(m_k)
    for k, v in m_k.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k].cpu()) and torch.equal(v.cpu(), m3.state_dict()[k].cpu()), k
    assert not m2.__dict__.get("_small_epochs")


@pytest.mark.parametrize("seed", list(range(16)))
def test_epoch_kernel_random_shapes_against_step_path(lib, seed, monkeypatch):
    """Seeded sweep over what the kernel's scope admits - 1 to 4 encoders of 0 to 3 hidden layers, any widths, state sizes
    that are not multiples of anything, 1 to 4 decoders, batches of 1 to 64 rows with a ragged last one, NaN batches - two
    epochs on both paths: History within 2e-5, Adam step counts equal."""
    monkeypatch.setenv("MMN_EPOCH_KERNEL", "1")
    rng = np.random.default_rng(500 + seed)
    E = int(rng.integers(1, 5))
    S = int(rng.choice([3, 8, 13, 16, 21, 32, 40]))
    D = int(rng.integers(1, 5))
    B = int(rng.choice([1, 5, 16, 17, 32, 47, 64]))
    encs = []
    for _ in range(E):
        nh = int(rng.integers(0, 4))
        H = tuple(int(rng.integers(1, 20)) for _ in range(nh))
        act = int(rng.choice([O.ACT_RELU, O.ACT_SIGMOID, O.ACT_IDENTITY])) if H else O.ACT_IDENTITY
        encs.append(O.EncoderSpec(int(rng.integers(1, 24)), H, act))
    spec = O.ModelSpec(S, encs, D, float(rng.choice([0.7, 1.0])), float(rng.choice([0.0, 0.3, 1.0])))
    nb = 7
    batches = O.synthetic_batches(spec, nb * B - (B // 3), B, seed=seed)
    for bi in rng.choice(nb, size=2, replace=False):
        slot = int(rng.integers(0, E))
        if batches[bi][0][slot].shape[0] > 0:
            batches[bi][0][slot][0, 0] = np.nan
    loader = _device_loader(batches)
    m_k, h_k, st_k, _ = _run(lib, spec, loader, 2, True)
    m_s, h_s, st_s, _ = _run(lib, spec, loader, 2, False)
    if not m_k.__dict__.get("_small_epochs"):
        pytest.skip("outside the kernel's scope (LDS image)")
    assert st_k == st_s, (st_k, st_s)
    for a, b in ((h_k.loss["train"], h_s.loss["train"]), (h_k.state_change_loss, h_s.state_change_loss)):
        assert rel_err(np.stack(a), np.stack(b)) < 2e-5, (spec, B)
    assert np.abs(np.stack(h_k.accuracy["train"]) - np.stack(h_s.accuracy["train"])).max() <= 2.0 / max(nb * B, 1)
