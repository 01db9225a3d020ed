"""k_adam (mmn_adam_step) against torch.optim.Adam on the same gradients: the optimizer.step() of
the reference's batch loop (multimodn.py:204).  Parameters and moments within 1e-5 relative (fp32),
step counters exact, skipped tensors bit-identical."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def make_flat(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    n = sum(int(np.prod(s)) for s in shapes)
    flat = torch.randn(n, generator=g).cuda()
    gflat = torch.zeros(n, device="cuda")
    ps, gs, off = [], [], 0
    for s in shapes:
        k = int(np.prod(s))
        p = torch.nn.Parameter(flat[off:off + k].view(s))
        ps.append(p)
        gs.append(gflat[off:off + k].view(s))
        off += k
    return flat, gflat, ps, gs


SHAPES = [(128,), (32, 64), (32,), (2, 128), (2,), (3,), (128, 160), (128,), (5, 7), (1,)]


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_matches_torch_adam_with_skipped_tensors(wd):
    import multimodn_amd as mm
    flat, gflat, ps, gs = make_flat(SHAPES, 0)
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = mm.optim.Adam(ps, 3e-3, betas=(0.8, 0.99), eps=1e-7, weight_decay=wd)
    ref = torch.optim.Adam(ref_ps, 3e-3, betas=(0.8, 0.99), eps=1e-7, weight_decay=wd)
    gen = torch.Generator().manual_seed(1)
    for it in range(12):
        gflat.copy_(torch.randn(gflat.numel(), generator=gen).cuda() * (10.0 ** (it % 3 - 1)))
        skip = {1, 2} if it in (3, 4, 7) else ({6} if it == 5 else set())
        for i, (p, g, rp) in enumerate(zip(ps, gs, ref_ps)):
            p.grad = None if i in skip else g
            rp.grad = None if i in skip else g.clone()
        before = [p.detach().clone() for p in ps]
        rbefore = [rp.detach().clone() for rp in ref_ps]
        opt.step()
        ref.step()
        for i, (p, rp, b, rb) in enumerate(zip(ps, ref_ps, before, rbefore)):
            upd, rupd = (p.detach() - b).cpu().numpy(), (rp.detach() - rb).cpu().numpy()
            if i in skip:
                assert np.array_equal(upd, np.zeros_like(upd))
            else:
                # parameters to fp32 rounding; the update itself (a difference of O(1) values, so
                # it carries the parameters' rounding: 2.4e-7 / 2e-3) to 5e-4
                assert rel_err(p.detach().cpu().numpy(), rp.detach().cpu().numpy()) < 1e-6, (it, i)
                assert rel_err(upd, rupd) < 5e-4, (it, i, rel_err(upd, rupd))
    for p, rp in zip(ps, ref_ps):
        assert float(opt.state[p]["step"]) == float(ref.state[rp]["step"])
        assert rel_err(opt.state[p]["exp_avg"].cpu().numpy(), ref.state[rp]["exp_avg"].cpu().numpy()) < 1e-5
        assert rel_err(opt.state[p]["exp_avg_sq"].cpu().numpy(), ref.state[rp]["exp_avg_sq"].cpu().numpy()) < 1e-5


def test_state_dict_round_trip_and_scattered_params():
    """state_dict layout equals torch.optim.Adam's (checkpoints move both ways); parameters that
    are NOT adjacent in memory still work (one launch per run)."""
    import multimodn_amd as mm
    ps = [torch.nn.Parameter(torch.randn(s).cuda()) for s in [(8, 4), (4,), (16,)]]
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ref = mm.optim.Adam(ps, 1e-2), torch.optim.Adam(ref_ps, 1e-2)
    for it in range(3):
        for p, rp in zip(ps, ref_ps):
            p.grad = torch.randn_like(p)
            rp.grad = p.grad.clone()
        opt.step()
        ref.step()
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    # torch's optimizer state -> ours, then continue in lock step
    opt2 = mm.optim.Adam(ps, 1e-2)
    opt2.load_state_dict(ref.state_dict())
    for it in range(3):
        grads = [torch.randn_like(p) for p in ps]      # fresh buffers: the runs must follow them
        for p, rp, g in zip(ps, ref_ps, grads):
            p.grad = g
            rp.grad = g.clone()
        opt2.step()
        ref.step()
    for p, rp in zip(ps, ref_ps):
        assert rel_err(p.detach().cpu().numpy(), rp.detach().cpu().numpy()) < 1e-5
        assert float(opt2.state[p]["step"]) == 6.0


def test_graph_replay_advances_steps():
    import multimodn_amd as mm
    flat, gflat, ps, gs = make_flat([(64, 32), (32,)], 2)
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ref = mm.optim.Adam(ps, 1e-3), torch.optim.Adam(ref_ps, 1e-3)
    for p, g in zip(ps, gs):
        p.grad = g
    gflat.normal_()
    opt.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    for rp, g in zip(ref_ps, gs):
        rp.grad = g.clone()
    ref.step()                       # the eager step above (capture itself does not execute)
    for it in range(4):
        gflat.normal_()
        for rp, g in zip(ref_ps, gs):
            rp.grad = g.clone()
        graph.replay()
        ref.step()
    torch.cuda.synchronize()
    assert float(opt.state[ps[0]]["step"]) == 5.0
    for p, rp in zip(ps, ref_ps):
        assert rel_err(p.detach().cpu().numpy(), rp.detach().cpu().numpy()) < 1e-5


def _c3_like_model(mm, seed=0, B=192):
    from oracle import multimodn_oracle as O
    from helpers import build_torch_model
    spec = O.ModelSpec(64, [O.EncoderSpec(16, (16,), O.ACT_RELU), O.EncoderSpec(32, (32, 16), O.ACT_RELU),
                            O.EncoderSpec(8, (), O.ACT_RELU)], 2, 1.0, 0.4)
    params = O.init_params(spec, seed)
    batches = O.synthetic_batches(spec, 3 * B, B, seed=5)
    return spec, params, batches, build_torch_model(spec, params, "cuda", mm)


@pytest.mark.parametrize("nan_policy", ["host", "device"])
def test_fused_step_equals_separate_step_and_skips_like_torch(nan_policy):
    """optimizer.step() fused into the last launch of the training step: bit-identical parameters
    to train step + k_adam, and an encoder skipped on a NaN batch keeps parameters, moments and
    step count untouched (torch leaves grad None parameters alone), under both NaN policies."""
    import multimodn_amd as mm
    spec, params, batches, model_f = _c3_like_model(mm)
    _, _, _, model_s = _c3_like_model(mm)
    model_f.nan_policy = model_s.nan_policy = nan_policy
    opt_f = mm.optim.Adam(list(model_f.parameters()), 2e-3)
    opt_s = mm.optim.Adam(list(model_s.parameters()), 2e-3)
    opt_s.fused_descriptor = lambda engine: None           # force the separate k_adam launch
    crit = torch.nn.CrossEntropyLoss()

    def loader(nan_at=None):
        out = []
        for i, (xs, y) in enumerate(batches):
            xs = [x.copy() for x in xs]
            if nan_at is not None and i == nan_at:
                xs[1][3, 2] = np.nan                       # encoder 1 is skipped for this whole batch
            out.append(([torch.from_numpy(x) for x in xs], torch.from_numpy(y)))
        return out

    model_f.train_epoch(loader(), opt_f, crit)
    model_s.train_epoch(loader(), opt_s, crit)
    before = {n: p.detach().clone() for n, p in model_f.named_parameters()}
    steps_before = {n: float(opt_f.state[p]["step"]) for n, p in model_f.named_parameters()}
    model_f.train_epoch(loader(nan_at=0)[:1], opt_f, crit)
    model_s.train_epoch(loader(nan_at=0)[:1], opt_s, crit)
    torch.cuda.synchronize()
    for (n, pf), (_, ps) in zip(model_f.named_parameters(), model_s.named_parameters()):
        skipped = n.startswith("encoders.1.")
        # (device policy + a separate optimizer sees ZERO gradients for the skipped encoder, not
        # None, so only the fused path can leave it untouched there)
        if not (skipped and nan_policy == "device"):
            assert torch.equal(pf, ps), n
            assert float(opt_f.state[pf]["step"]) == float(opt_s.state[ps]["step"])
        assert float(opt_f.state[pf]["step"]) == steps_before[n] + (0 if skipped else 1), n
        assert torch.equal(pf, before[n]) == skipped, n


def test_fused_path_is_taken_and_matches_torch_adam():
    import multimodn_amd as mm
    spec, params, batches, model = _c3_like_model(mm, seed=3)
    _, _, _, ref = _c3_like_model(mm, seed=3)
    opt = mm.optim.Adam(list(model.parameters()), 1e-3)
    topt = torch.optim.Adam(list(ref.parameters()), 1e-3)
    crit = torch.nn.CrossEntropyLoss()
    loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in batches]
    taken = []
    orig = opt.mark_fused_step
    opt.mark_fused_step = lambda: (taken.append(1), orig())[1]
    for _ in range(2):
        model.train_epoch(loader, opt, crit)
        ref.train_epoch(loader, topt, crit)
    assert len(taken) == 2 * len(loader)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert rel_err(p.detach().cpu().numpy(), q.detach().cpu().numpy()) < 1e-5, n


def test_data_parallel_tail_in_one_launch_equals_two():
    """mmn_adam_step_accumulate (epoch accumulation + Adam after the all-reduce, one launch) against
    mmn_epoch_accumulate followed by k_adam: bit-identical parameters and epoch sums."""
    import multimodn_amd as mm
    spec, params, batches, m1 = _c3_like_model(mm, seed=5)
    _, _, _, m2 = _c3_like_model(mm, seed=5)
    outs = []
    for model, fused_tail in ((m1, True), (m2, False)):
        model.nan_policy = "device"
        opt = mm.optim.Adam(list(model.parameters()), 1e-3)
        eng = model._get_engine(batches[0][1].shape[0])
        eng.epoch_reset()
        eng.assign_grads(None)
        for xs, y in batches:
            dx = [torch.from_numpy(x).cuda() for x in xs]
            dy = torch.from_numpy(y).cuda()
            b = eng.make_batch(dx, dy, [(i, i) for i in range(spec.E)], device_nan_flags=True)
            eng.local_step(b, 1.0, 0.004, accumulate=False)          # what a rank does before the all-reduce
            if fused_tail:
                assert eng.accumulate_and_step(1.0, 0.004, opt)
            else:
                eng.accumulate(1.0, 0.004)
            opt.step()
        torch.cuda.synchronize()
        outs.append(({n: p.detach().clone() for n, p in model.named_parameters()}, eng.epoch_read()))
    for n in outs[0][0]:
        assert torch.equal(outs[0][0][n], outs[1][0][n]), n
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def test_graph_replayed_fused_training_step_equals_eager():
    """bench.py's mode of operation: the whole fused step (prepare, k_fb8, k_wgrad, k_reduce + Adam)
    captured into a hipGraph and replayed: bit-identical parameters, moments and step counts to the
    same number of eager steps."""
    import multimodn_amd as mm
    spec, params, batches, m1 = _c3_like_model(mm, seed=9)
    _, _, _, m2 = _c3_like_model(mm, seed=9)
    xs, y = batches[0]
    res = []
    for model, use_graph in ((m1, True), (m2, False)):
        model.nan_policy = "device"
        opt = mm.optim.Adam(list(model.parameters()), 3e-3)
        eng = model._get_engine(len(y))
        eng.epoch_reset()
        eng.assign_grads(None)
        dx = [torch.from_numpy(x).cuda() for x in xs]
        dy = torch.from_numpy(y).cuda()
        b = eng.make_batch(dx, dy, [(i, i) for i in range(spec.E)], device_nan_flags=True)

        def step():
            assert eng.local_step(b, 1.0, 0.004, accumulate=True, optimizer=opt)
            opt.step()
        step()                                                   # eager: builds the optimizer's flat state
        torch.cuda.synchronize()
        if use_graph:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(5):
                g.replay()
        else:
            for _ in range(5):
                step()
        torch.cuda.synchronize()
        res.append(({n: p.detach().clone() for n, p in model.named_parameters()},
                    {n: float(opt.state[p]["step"]) for n, p in model.named_parameters()}, eng.epoch_read()))
    for n in res[0][0]:
        assert torch.equal(res[0][0][n], res[1][0][n]), n
        assert res[0][1][n] == res[1][1][n] == 6.0
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
