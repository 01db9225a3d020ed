"""Host-side logic of the package on a GPU-less machine: the public surface driven through a
test-only oracle backend (tests/oracle_engine.py) must reproduce the reference's golden History
arrays and trained weights; plus the plugin/dataset/history classes themselves."""
import io
import pickle

import numpy as np
import pytest
import torch

import multimodn_amd as mm
from helpers import (GOLDEN_NAMES, MIMIC_GOLDEN_NAMES, Golden, assert_within_fp32_noise, build_torch_model, fp64_trajectory,
                     rel_err)
from oracle_engine import OracleEngine


def make_loader(g):
    loader = []
    for b in g.batches():
        item = [[torch.from_numpy(x) for x in b[0]], torch.from_numpy(b[1])]
        if len(b) > 2:
            item.append(torch.from_numpy(b[2]))
        loader.append(tuple(item))
    return loader


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_train_epoch_host_logic_reproduces_golden_history(name):
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    model._engine_factory = OracleEngine
    opt = torch.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    for _ in range(g.epochs):
        assert model.train_epoch(make_loader(g), opt, torch.nn.CrossEntropyLoss(), hist) is None
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < 2e-6
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 2e-6
    for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        got = np.stack(getattr(hist, k)["train"])
        assert got.dtype == z["hist/" + k].dtype and got.shape == z["hist/" + k].shape
        assert np.array_equal(got, z["hist/" + k]), k
    sd = model.state_dict()
    w64 = fp64_trajectory(g)[0]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n].numpy(), w, w64[n], n)


@pytest.mark.parametrize("name", MIMIC_GOLDEN_NAMES)
def test_mimic_family_host_logic_reproduces_golden_history(name):
    """MIMIC_MLPEncoder + MLPDecoder through the public surface: state_dict keys of the reference (Dropout at
    layers.0), train-mode dropout masks handed to the engine per step (here: the ones the reference drew), none
    in eval mode, History and trained weights of the reference run."""
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    assert list(model.state_dict().keys()) == [str(n) for n in g.z["param_names"]]
    model._engine_factory = OracleEngine
    opt = torch.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])


    def provide(e, batch, width):
        m = g.step_masks(model.train_steps_launched).get(e)
        assert m is None or m.shape == (batch, width)
        return None if m is None else torch.from_numpy(m)

    model.dropout_mask_provider = provide
    for _ in range(g.epochs):
        model.train_epoch(make_loader(g), opt, torch.nn.CrossEntropyLoss(), hist)
    z = g.z
    assert rel_err(np.stack(hist.loss["train"]), z["hist/loss"]) < 2e-6
    assert rel_err(np.stack(hist.state_change_loss), z["hist/state_change"]) < 2e-6
    for k in ("accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        assert np.array_equal(np.stack(getattr(hist, k)["train"]), z["hist/" + k]), k
    sd = model.state_dict()
    w64 = fp64_trajectory(g)[0]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n].numpy(), w, w64[n], n)
    # eval mode: nn.Dropout is the identity, no masks are drawn
    model.dropout_mask_provider = lambda *a: pytest.fail("dropout mask requested in eval mode")
    th = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    model.test(make_loader(g), torch.nn.CrossEntropyLoss(), th, tag="test")
    assert rel_err(th.loss["test"][0], z["eval/test_loss"]) < 2e-6


def test_reference_written_checkpoint_loads():
    """the checkpoint the MIMIC pipelines write (mimic_multi_task_pipeline.py:150-154), produced by
    the REFERENCE's trained model (tests/golden/ref_checkpoint_mimic_drop.pt, tensors only), loads into the build with
    strict key matching and evaluates to the reference's test() numbers.  The other direction (a build checkpoint
    loading into the reference) is asserted in tests/golden/make_golden.py, the only place that has the reference."""
    import os
    from helpers import GOLDEN
    g = Golden("mimic_drop")
    ck = torch.load(os.path.join(GOLDEN, "ref_checkpoint_mimic_drop.pt"))
    assert set(ck) == {"epoch", "model_state_dict", "auc_bac_val_cum"} and ck["epoch"] == g.epochs
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    missing = model.load_state_dict(ck["model_state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for n, w in g.final_params().items():
        assert np.array_equal(model.state_dict()[n].numpy(), w), n
    model._engine_factory = OracleEngine
    th = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    model.test(make_loader(g), torch.nn.CrossEntropyLoss(), th, tag="test")
    assert rel_err(th.loss["test"][0], g.z["eval/test_loss"]) < 2e-6
    assert np.array_equal(th.accuracy["test"][0], g.z["eval/test_accuracy"])


def test_mimic_plugin_forward_contract():
    """The plugins' own forward (module-level contract) equals the oracle's restatement of the reference modules."""
    torch.manual_seed(0)
    S, F = 6, 4
    enc = mm.MIMIC_MLPEncoder(S, F, (5, 3), dropout=0.0)
    dec = mm.MLPDecoder(S, (4,), 2)
    assert isinstance(enc.layers[0], torch.nn.Dropout) and enc.layers[1].in_features == F + S
    assert [n for n, _ in dec.named_parameters()] == ["layers.0.weight", "layers.0.bias", "layers.1.weight", "layers.1.bias"]
    s, x = torch.randn(7, S), torch.randn(7, F)
    h = torch.cat([x, s], 1)
    for lin in enc.linears:
        h = torch.relu(lin(h))
    assert torch.allclose(enc.eval()(s, x), h) and float(h.min()) >= 0
    out = dec(s)
    assert out.shape == (7, 2) and 0 < float(out.min()) and float(out.max()) < 1
    enc_t = mm.MIMIC_MLPEncoder(S, F, (5,), dropout=0.5).train()
    assert not torch.allclose(enc_t(s, x), enc_t(s, x))     # training mode draws a new mask per call


def test_nan_skip_leaves_grad_none_and_adam_untouched():
    g = Golden("nan_skip")
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    model._engine_factory = OracleEngine
    opt = torch.optim.Adam(list(model.parameters()), 0.01)
    loader = make_loader(g)[1:2]                       # the batch with a NaN in data slot 1
    before = model.encoders[1].layers[0].weight.detach().clone()
    model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss())
    assert model.encoders[1].layers[0].weight.grad is None
    assert torch.equal(before, model.encoders[1].layers[0].weight.detach())
    assert model.encoders[0].layers[0].weight.grad is not None


def test_state_dict_keys_and_pickle_roundtrip():
    g = Golden("c2_split")
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    keys = list(model.state_dict().keys())
    assert keys[0] == "init_state.state_value"
    assert "encoders.1.layers.2.bias" in keys and "decoders.1.fc.weight" in keys
    assert keys == g.spec.param_names()
    model._engine_factory = OracleEngine
    model.train_epoch(make_loader(g), torch.optim.Adam(model.parameters(), 0.01), torch.nn.CrossEntropyLoss())
    clone = pickle.loads(pickle.dumps(model))           # pipelines pkl.dump the whole model
    assert clone._engine is None
    for (n1, p1), (n2, p2) in zip(model.state_dict().items(), clone.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)
    buf = io.BytesIO()
    torch.save({"epoch": 1, "model_state_dict": model.state_dict()}, buf)      # MIMIC pipelines' checkpoint
    buf.seek(0)
    fresh = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    fresh.load_state_dict(torch.load(buf)["model_state_dict"])
    assert torch.equal(fresh.decoders[0].fc.weight, model.decoders[0].fc.weight)


def test_encoder_iterable_semantics():
    g = Golden("seq_perm")
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    assert model.get_encoder_iterable(None, False, True) == [(0, 0), (1, 1), (2, 2), (3, 3)]
    seq = torch.tensor([[2, 0, 1, 3]] * 5)
    assert model.get_encoder_iterable(seq, False, True) == [(0, 2), (1, 0), (2, 1), (3, 3)]
    with pytest.raises(ValueError, match="different values across the batch"):
        model.get_encoder_iterable(torch.tensor([[0, 1, 2, 3], [1, 0, 2, 3]]), False, True)
    model.shuffle_mode = True
    assert sorted(model.get_encoder_iterable(None, True, True)) == [(0, 0), (1, 1), (2, 2), (3, 3)]
    assert model.get_encoder_iterable(None, True, False) == [(0, 0), (1, 1), (2, 2), (3, 3)]


def test_state_change_penalty_is_scaled_and_signature_matches():
    import inspect
    model = mm.MultiModN(4, [mm.MLPEncoder(4, 3, (2,))], [mm.LogisticDecoder(4)], 0.7, 0.3, device=torch.device("cpu"))
    assert abs(model.state_change_penalty - 0.003) < 1e-12 and model.err_penalty == 0.7
    assert list(inspect.signature(mm.MultiModN.__init__).parameters)[1:] == [
        "state_size", "encoders", "decoders", "err_penalty", "state_change_penalty", "shuffle_mode", "init_state", "device"]
    assert list(inspect.signature(mm.MultiModN.train_epoch).parameters)[1:] == [
        "train_loader", "optimizer", "criterion", "history", "log_interval", "logger", "last_epoch"]
    assert list(inspect.signature(mm.MultiModN.test).parameters)[1:] == [
        "test_loader", "criterion", "history", "tag", "log_results", "logger"]


def test_unsupported_plugins_and_criteria_are_refused():
    class Odd(mm.MultiModEncoder):
        def forward(self, state, x):
            return state
    model = mm.MultiModN(4, [Odd(4)], [mm.LogisticDecoder(4)], 1.0, 0.0, device=torch.device("cpu"))
    model._engine_factory = OracleEngine
    loader = [([torch.zeros(2, 3)], torch.zeros(2, 1, dtype=torch.int64))]
    with pytest.raises(mm.UnsupportedModelError):
        model.train_epoch(loader, torch.optim.Adam(model.parameters()), torch.nn.CrossEntropyLoss())
    ok = mm.MultiModN(4, [mm.MLPEncoder(4, 3, (2,))], [mm.LogisticDecoder(4)], 1.0, 0.0, device=torch.device("cpu"))
    with pytest.raises(mm.UnsupportedModelError):
        ok.train_epoch(loader, torch.optim.Adam(ok.parameters()), torch.nn.CrossEntropyLoss(reduction="sum"))


def test_product_engine_refuses_cpu_device():
    ok = mm.MultiModN(4, [mm.MLPEncoder(4, 3, (2,))], [mm.LogisticDecoder(4)], 1.0, 0.0, device=torch.device("cpu"))
    loader = [([torch.zeros(2, 3)], torch.zeros(2, 1, dtype=torch.int64))]
    with pytest.raises(mm.hip.MmnError, match="no CPU fallback"):
        ok.train_epoch(loader, torch.optim.Adam(ok.parameters()), torch.nn.CrossEntropyLoss())


def test_plugin_forward_contract():
    torch.manual_seed(0)
    enc = mm.MLPEncoder(5, 3, (4, 2))
    s, x = torch.randn(6, 5), torch.randn(6, 3)
    h = torch.relu(enc.layers[1](torch.relu(enc.layers[0](x))))
    assert torch.allclose(enc(s, x), enc.layers[2](torch.cat([h, s], 1)))
    assert enc.layers[2].in_features == 2 + 5
    slp = mm.LogisticEncoder(5, 3)
    assert len(slp.layers) == 1 and slp.layers[0].in_features == 8
    dec = mm.LogisticDecoder(5)
    assert dec.n_classes == 2 and dec(s).shape == (6, 2) and float(dec(s).max()) < 1
    init = mm.TrainableInitState(5)
    assert init(3).shape == (3, 5) and torch.equal(init(3)[0], init.state_value[0])


@pytest.mark.parametrize("name", ["titanic_featurewise", "titanic_missingness"])
def test_featurewise_pipelines_reproduce_the_reference_run(name):
    """The reference's feature-wise Titanic pipelines, written with this package's classes (MLPFeatureEncoder per feature
    over a FeatureWiseDataset through a stock DataLoader; batch 32, and batch size 1 with missing values kept as NaN):
    parameter names, History arrays, trained weights and the forward-only epoch equal the reference's own run."""
    from helpers import assert_history_matches_golden, featurewise_pipeline
    g = Golden(name)
    model, loader = featurewise_pipeline(g, "cpu", mm, OracleEngine)
    assert list(model.state_dict().keys()) == list(g.spec.param_names())
    b0 = next(iter(loader))
    assert [tuple(x.shape) for x in b0[0]] == [(g.cfg["B"], 1)] * len(g.cfg["F"]) and b0[0][0].dtype == torch.float32
    opt = torch.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = mm.MultiModNHistory(["Survived"])
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    assert_history_matches_golden(hist, g)
    sd = model.state_dict()
    w64 = fp64_trajectory(g)[0]
    for n, w in g.final_params().items():
        assert_within_fp32_noise(sd[n].numpy(), w, w64[n], n)
    # test(): History arrays as the reference's; the per-decoder report covers the batches whose LAST encoder ran (the
    # reference's own report raises once a batch's last feature is missing: the fixture records that, eval/test_report_raised)
    th = mm.MultiModNHistory(["Survived"])
    res = model.test(loader, torch.nn.CrossEntropyLoss(), th)
    assert rel_err(th.loss["test"][0], g.z["eval/test_loss"]) < 2e-5
    assert np.array_equal(th.accuracy["test"][0], g.z["eval/test_accuracy"])
    n_last = sum(len(b[1]) for b in g.batches() if not np.isnan(b[0][-1]).any())
    assert ("eval/test_report_raised" in g.z.files) == (n_last < g.cfg["N"])
    tn, fp, fn, tp = (int(res[0][mm.metrics.performance_metrics.index(k)]) for k in ("tn", "fp", "fn", "tp"))
    assert tn + fp + fn + tp == n_last


def test_feature_encoder_is_the_references_constructor():
    torch.manual_seed(0)
    enc = mm.MLPFeatureEncoder(5, 4, torch.sigmoid)              # mlp_encoder.py:84-91: (state_size, hidden_size, activation, device)
    assert isinstance(enc, mm.MLPEncoder) and enc.n_features == 1 and enc.hidden_layers == (4,) and enc.activation is torch.sigmoid
    assert [tuple(l.weight.shape) for l in enc.layers] == [(4, 1), (5, 9)]
    s, x = torch.randn(6, 5), torch.randn(6, 1)
    assert torch.equal(enc(s, x), mm.MLPEncoder.forward(enc, s, x))
    assert torch.equal(enc(s, x.double().numpy()), enc(s, x))   # whatever the loader hands over becomes float32 (Tensor(x), :94)


def test_display_arch_and_module_level_helpers(capsys):
    """multimodn.py:494-507 (display_arch: one table per encoder / decoder for one sample's features) and the two
    module-level helpers a pipeline may import from the model's module (pipelines/mimic/haim_api.py:12)."""
    from multimodn_amd.multimodn import compute_metrics, get_performance_metrics
    assert get_performance_metrics is mm.metrics.get_performance_metrics
    model = mm.MultiModN(8, [mm.MLPEncoder(8, 6, (5, 5)), mm.MIMIC_MLPEncoder(8, 4, (6,)), mm.MLPFeatureEncoder(8, 3)],
                         [mm.LogisticDecoder(8), mm.MLPDecoder(8, (4,), 2)], 0.7, 0.3, device=torch.device("cpu"))
    model.display_arch([np.zeros(6), np.zeros(4), np.zeros(1)])
    out = capsys.readouterr().out
    assert out.count("Encoder ") == 3 and out.count("Decoder ") == 2
    assert "Total params: 177" in out and "Total params: 134" in out and "Total params: 18" in out     # encoder 0 / 1, decoder 0
    grids = [torch.zeros(2, 1) for _ in range(4)]
    compute_metrics(*grids, torch.tensor([[3, 1], [2, 4]]), 1, 0)                # cm[true][pred]
    assert [float(g[1, 0]) for g in grids] == [4.0, 3.0, 1.0, 2.0]               # tp, tn, fp, fn
    compute_metrics(*grids, None, 0, 0)
    assert all(torch.isnan(g[0, 0]) for g in grids)


def test_partition_dataset_and_collate():
    X = np.arange(40, dtype=np.float32).reshape(8, 5)
    y = np.arange(16).reshape(8, 2) % 2
    ds = mm.PartitionDataset(X, y, [3, 2])
    xs, t = ds[2]
    assert [tuple(v.shape) for v in xs] == [(3,), (2,)] and np.array_equal(t, y[2])
    with pytest.raises(ValueError):
        mm.PartitionDataset(X, y, [3, 3])
    batch = next(iter(torch.utils.data.DataLoader(ds, 4)))
    assert batch[0][0].shape == (4, 3) and batch[0][1].shape == (4, 2) and batch[1].shape == (4, 2)
    parts = ds.random_split((0.5, 0.5), seed=0, balanced_target_idx=0)
    assert sorted(parts[0].indices + parts[1].indices) == list(range(8))
    assert len(mm.FeatureWiseDataset(X, y)[0][0]) == 5


def test_partition_dataset_batched_fetch_equals_per_sample_collate():
    """`PartitionDataset.__getitems__` (what torch's DataLoader calls per batch; datasets/multimod_dataset.py:55-88 has only
    `__getitem__`): every batch equals - structure, dtypes, values, order under a seeded shuffle - what default_collate
    makes of the per-sample items; through `Subset` (random_split), with a remainder batch, 1-D targets, worker processes,
    and a custom collate_fn still sees `(List[Tensor], target)` samples."""
    from torch.utils.data import DataLoader, Dataset
    rng = np.random.default_rng(0)
    X = rng.random((700, 10)); y = (rng.random((700, 3)) > 0.5).astype(np.int64)      # float64 features: items are float32 (Tensor(...))
    ds = mm.PartitionDataset(X, y, [4, 6])

    class PerSample(Dataset):                                # hides __getitems__: the reference's path
        def __init__(self, d): self.d = d
        def __len__(self): return len(self.d)
        def __getitem__(self, i): return self.d[i]

    def same(a, b):
        assert type(a) is type(b) and type(a[0]) is type(b[0]) and len(a[0]) == len(b[0])
        for u, v in zip(a[0], b[0]):
            assert u.dtype == v.dtype and torch.equal(u, v)
        assert a[1].dtype == b[1].dtype and torch.equal(a[1], b[1])
    for kw in (dict(batch_size=256), dict(batch_size=64, shuffle=True), dict(batch_size=33, drop_last=True), dict(batch_size=128, num_workers=2)):
        A = list(DataLoader(ds, generator=torch.Generator().manual_seed(3), **kw))
        Bs = list(DataLoader(PerSample(ds), generator=torch.Generator().manual_seed(3), **kw))
        assert len(A) == len(Bs)
        for a, b in zip(A, Bs):
            same(a, b)
    sub = ds.random_split([0.7, 0.3], seed=1)[0]
    same(next(iter(DataLoader(sub, batch_size=100))), next(iter(DataLoader(PerSample(sub), batch_size=100))))
    ds1 = mm.PartitionDataset(X, y[:, 0], [10])
    same(next(iter(DataLoader(ds1, batch_size=64))), next(iter(DataLoader(PerSample(ds1), batch_size=64))))
    seen = next(iter(DataLoader(ds, batch_size=5, collate_fn=lambda samples: [(len(xs), tuple(xs[1].shape), tuple(t.shape)) for xs, t in samples])))
    assert seen == [(2, (6,), (3,))] * 5
    rows = ds.__getitems__([5, 2, 9])
    assert torch.equal(rows[1][0][0], ds[2][0][0]) and np.array_equal(rows[2][1], ds[9][1]) and len(rows[0]) == 2

    class Augmented(mm.PartitionDataset):                    # a user subclass that overrides __getitem__ must see every sample
        def __getitem__(self, i):                            # (the batched fetch silently bypassed it)
            xs, t = super().__getitem__(i)
            return [x * 2 for x in xs], t, np.int64(i)
    aug = Augmented(X, y, [4, 6])
    xs2, t2, idx = next(iter(DataLoader(aug, batch_size=8)))
    assert idx.tolist() == list(range(8)) and torch.equal(xs2[0], 2 * next(iter(DataLoader(ds, batch_size=8)))[0][0])


def test_history_results_table(tmp_path):
    h = mm.MultiModNHistory(["a", "b"])
    h.state_change_loss.append(np.array([0.5, 0.25]))
    for k in ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy"):
        getattr(h, k)["train"].append(np.arange(6, dtype=float).reshape(3, 2))
    df = h.get_results()
    assert list(df.index) == ["a", "b"] and df.shape == (2, 6)
    assert df["State change loss"].tolist() == [0.25, 0.25] and df["Train loss"].tolist() == [4.0, 5.0]
    h.save_results(tmp_path / "r.csv")
    assert (tmp_path / "r.csv").read_text().startswith("Target,State change loss,Train loss")


def test_hip_adam_refuses_cpu_parameters():
    """No torch-op fallback for optimizer.step(): CPU parameters raise instead of silently running."""
    import multimodn_amd as mm
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    opt = mm.optim.Adam([p], 1e-3)
    with pytest.raises(mm.hip.MmnError):
        opt.step()
    with pytest.raises(NotImplementedError):
        mm.optim.Adam([p], amsgrad=True)


def test_history_list_settles_pending_epochs_before_anything_moves_entries():
    """HistoryList entries of an epoch in flight are placeholders that the epoch later fills in BY POSITION: every list
    operation that moves, removes or searches entries resolves them first (pop / del / insert / sort / index / in / extend)."""
    from multimodn_amd.history import HistoryList, PendingEpoch

    def pending(v):
        return PendingEpoch(lambda: None, lambda: {"loss": np.full(2, float(v))})

    def fresh():
        lst = HistoryList()
        lst.append(np.zeros(2))
        for v in (1, 2):
            lst.append_pending(pending(v), "loss")
        return lst

    lst = fresh()
    del lst[0]
    assert [float(a[0]) for a in lst] == [1.0, 2.0]
    lst = fresh()
    assert float(lst.pop(0)[0]) == 0.0 and float(lst.pop()[0]) == 2.0 and len(lst) == 1 and float(lst[0][0]) == 1.0
    lst = fresh()
    lst.insert(0, np.full(2, 9.0))
    assert [float(a[0]) for a in lst] == [9.0, 0.0, 1.0, 2.0]
    lst = fresh()
    lst.reverse()
    assert [float(a[0]) for a in lst] == [2.0, 1.0, 0.0]
    lst = fresh()
    other = fresh()
    lst.extend(other)
    assert [float(a[0]) for a in lst] == [0.0, 1.0, 2.0, 0.0, 1.0, 2.0]
    assert all(isinstance(a, np.ndarray) for a in list.__iter__(lst))
    lst = fresh()
    lst.clear()
    assert len(lst) == 0


def test_loader_prefetch_keeps_order_forwards_exceptions_and_stops():
    """multimodn.py::_LoaderPrefetch (round 6): the batch loop's helper thread hands batches on in order, re-raises what the
    loader raises where next() is called, ends with StopIteration, and stops pulling once it is closed."""
    from multimodn_amd.multimodn import _LoaderPrefetch
    assert list(_LoaderPrefetch(iter(range(100)))) == list(range(100))

    def bad():
        yield 1
        yield 2
        raise ValueError("broken sample")
    p = _LoaderPrefetch(bad())
    assert next(p) == 1 and next(p) == 2
    with pytest.raises(ValueError, match="broken sample"):
        next(p)
    pulled = []

    def slow():
        for i in range(1000):
            pulled.append(i)
            yield i
    q = _LoaderPrefetch(slow(), depth=2)
    assert next(q) == 0
    q.close()
    assert not q.thread.is_alive() and len(pulled) <= 5       # (the batch in hand + the queue's depth + one in the making)
