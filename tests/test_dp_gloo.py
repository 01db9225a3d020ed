"""Data-parallel protocol on two CPU ranks (gloo): every rank feeds its shard of each global
mini-batch, ONE all-reduce of [grads | stats] per step, identical Adam on every rank.  The result
must equal the single-process run on the concatenated batches.  Arithmetic comes from the test-only
oracle backend; what is under test is the package's DP logic (multimodn.py _ingest/_run_step)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import Golden, build_torch_model, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_dir):
    import torch.distributed as dist
    import multimodn_amd as mm
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cpu", mm)
    model._engine_factory = OracleEngine
    model.enable_data_parallel()
    opt = torch.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    loader = []
    for b in g.batches():
        n = len(b[1])
        assert n % world == 0
        lo, hi = rank * n // world, (rank + 1) * n // world
        item = [[torch.from_numpy(x[lo:hi]) for x in b[0]], torch.from_numpy(b[1][lo:hi])]
        if len(b) > 2:
            item.append(torch.from_numpy(b[2][lo:hi]))
        loader.append(tuple(item))
    # MIMIC family: every rank feeds ITS rows of the dropout masks the reference drew for the global batch


    def provide(e, batch, width):
        m = g.step_masks(model.train_steps_launched).get(e)
        if m is None:
            return None
        n = m.shape[0]
        return torch.from_numpy(m[rank * n // world:(rank + 1) * n // world])

    model.dropout_mask_provider = provide
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=np.stack(hist.loss["train"]),
             acc=np.stack(hist.accuracy["train"]), sc=np.stack(hist.state_change_loss),
             **{"p/" + k: v.numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["seq_perm", "nan_skip", "mimic_drop"])
def test_two_rank_dp_equals_single_process(name, tmp_path):
    g = Golden(name)
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:                                   # replicas stay bit-identical
        assert np.array_equal(r0[k], r1[k]), k
    z = g.z
    assert rel_err(r0["loss"], z["hist/loss"]) < 5e-6
    assert rel_err(r0["sc"], z["hist/state_change"]) < 5e-6
    assert np.abs(r0["acc"] - z["hist/accuracy"]).max() <= 1.0 / g.cfg["B"] + 1e-12
    for n, w in g.final_params().items():
        assert rel_err(r0["p/" + n], w) < 1e-4, n


# ------------------------------------------------------------------------------------------------
# The same protocol on the HIP engine: two ranks sharing ONE GPU (gloo moves the reduce buffer through
# the host, RCCL refuses two ranks on one device).  Checks the real kernels' shard arithmetic
# (batch_global divisors), the one-buffer all-reduce, the global NaN decision under both policies and
# the one-launch data-parallel tail.
# ------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, name, policy, out_dir):
    import torch.distributed as dist
    import multimodn_amd as mm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = Golden(name)
    model = build_torch_model(g.spec, g.init_params(), "cuda", mm)
    model.nan_policy = policy
    model.enable_data_parallel()
    opt = mm.optim.Adam(list(model.parameters()), g.cfg["lr"])
    hist = mm.MultiModNHistory([f"t{d}" for d in range(g.spec.D)])
    loader = []
    for b in g.batches():
        n = len(b[1])
        lo, hi = rank * n // world, (rank + 1) * n // world
        item = [[torch.from_numpy(x[lo:hi]) for x in b[0]], torch.from_numpy(b[1][lo:hi])]
        if len(b) > 2:
            item.append(torch.from_numpy(b[2][lo:hi]))
        loader.append(tuple(item))


    def provide(e, batch, width):                         # MIMIC family: this rank's rows of the reference's masks
        m = g.step_masks(model.train_steps_launched).get(e)
        if m is None:
            return None
        n = m.shape[0]
        return torch.from_numpy(m[rank * n // world:(rank + 1) * n // world])

    model.dropout_mask_provider = provide
    for _ in range(g.epochs):
        model.train_epoch(loader, opt, torch.nn.CrossEntropyLoss(), hist)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=np.stack(hist.loss["train"]),
             acc=np.stack(hist.accuracy["train"]), sc=np.stack(hist.state_change_loss),
             **{"p/" + k: v.cpu().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name,policy", [("seq_perm", "host"), ("nan_skip", "host"), ("nan_skip", "device"), ("c2_split", "device"),
                                         ("mimic_drop", "device"), ("mimic_mixed", "host")])
def test_two_rank_dp_on_one_gpu_equals_reference_golden(name, policy, tmp_path):
    g = Golden(name)
    mp.spawn(_gpu_worker, args=(2, _free_port(), name, policy, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:                                   # replicas stay bit-identical
        assert np.array_equal(r0[k], r1[k]), k
    z = g.z
    assert rel_err(r0["loss"], z["hist/loss"]) < 1e-5
    assert rel_err(r0["sc"], z["hist/state_change"]) < 1e-5
    assert np.abs(r0["acc"] - z["hist/accuracy"]).max() <= 1.0 / g.cfg["B"] + 1e-12
    # (device policy: the one-launch tail leaves a skipped encoder's parameters untouched, like the
    #  reference's grad-None parameters, so the trained weights agree under both policies)
    for n, w in g.final_params().items():
        assert rel_err(r0["p/" + n], w) < 1e-4, n
