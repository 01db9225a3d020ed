"""MultiModN: the reference's model-driver surface (multimodn/multimodn.py:65-531) with the
training step executed by hand-written HIP kernels on MI355X.

Drop-in contract: same constructor and method signatures, same attribute
and state_dict names, same History arrays (names, shapes, dtypes).  What differs is inside one
mini-batch: the body of the reference's batch loop (multimodn.py:137-203) is one call into
libmmn_hip.so (multimodn_amd/engine.py); per-step grids stay on the device and are read once per
epoch.  There is no torch-op fallback for that body: without the HIP library or off an AMD GPU
`train_epoch` raises.
"""
from __future__ import annotations

import os
import random
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor
from torch.optim import Optimizer
from torch.utils.data import DataLoader

import ctypes as C

from . import hip
from .decoders import MultiModDecoder
from .encoders import MultiModEncoder
from .engine import HipChainEngine, check_criterion
from .history import HistoryList, MultiModNHistory, PendingEpoch
from .metrics import compute_metrics, get_performance_metrics      # noqa: F401  (importable from here as from multimodn/multimodn.py)
from .state import InitState, TrainableInitState


def _has_nan_host(t: Tensor) -> bool:
    """`any(t.isnan().flatten())` (multimodn.py:168) for a host tensor: a sum propagates NaN, so one
    reduction answers "no NaN" for clean data; only a NaN sum (a real NaN, or +inf and -inf
    cancelling) pays for the exact element test."""
    if t.device.type != "cpu" or not t.is_floating_point():
        return bool(torch.isnan(t).any())
    if not bool(torch.isnan(t.sum())):
        return False
    return bool(torch.isnan(t).any())


class _HostStager:
    """Host batches -> device through ONE pinned staging buffer and ONE copy per batch: the slots of
    a host batch are small separate pageable tensors, and a pageable copy runs at ~1 GB/s on this
    stack (3 ms for the 4 MB of a 4096-row MIMIC-shaped batch); packed into pinned memory the same
    bytes move at PCIe speed.  Three buffers rotate; a buffer is reused only after its copy's event
    has completed."""

    def __init__(self, device: torch.device, depth: int = 3):
        self.device, self.depth = device, depth
        self.slots: List[Optional[tuple]] = [None] * depth
        self.turn = 0
        #: the copies run on a stream of their own (round 6): batch t + 1's packed copy overlaps step t instead
        #: of standing in front of it on the one stream - `MMN_COPY_STREAM=0` keeps them on the current stream
        self.copy_stream: Optional["torch.cuda.Stream"] = None
        self.use_copy_stream = os.environ.get("MMN_COPY_STREAM", "1") != "0"
        self.last_event: Optional["torch.cuda.Event"] = None   # the event of the last stage() whose wait was left to the caller


    def stage(self, data: Sequence[Tensor], y: Tensor, defer_wait: bool = False) -> Tuple[List[Tensor], Tensor]:
        """`defer_wait`: the caller makes the stream that runs the step wait for `self.last_event` itself, right in front of
        that step's launches (the batch loop, which stages batch t + 1 BEFORE it launches step t: waiting here would put
        step t behind copy t + 1).  Otherwise the current stream waits here and the call behaves like a copy in line.
        Order that keeps the ring safe: a copy waits for everything the current stream has been given so far - the buffer it
        overwrites was last read `depth` steps ago, by a step that was launched before this call."""
        # (the views into a slot's pinned and device buffers are made once per batch shape: at the reference pipelines' 16-row
        #  batches the shape arithmetic and the fourteen views of a batch cost more host time than its copies)
        # (pinned inputs copied straight from where they are - one DMA per tensor, no packing - were measured and are SLOWER:
        #  five 1 MB copies take 141 us per 4096-row batch against 127 us for the packed one on the same box, tools/time_h2d.py)
        use_cs = self.use_copy_stream and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        sig = (tuple(tuple(t.shape) for t in data), tuple(y.shape))
        i = self.turn
        self.turn = (self.turn + 1) % self.depth
        ent = self.slots[i]
        if ent is not None:
            ent[2].synchronize()                                 # its previous copy has left the buffer
        if ent is None or ent[3] != sig:
            sizes = [int(t.numel()) for t in data]
            n_f, n_y = sum(sizes), int(y.numel())
            off_y = (4 * n_f + 15) // 16 * 16
            used = off_y + 8 * n_y
            if ent is None or ent[0].numel() < used + 64:
                pinned = torch.empty(used + 64, dtype=torch.uint8).pin_memory()
                dev = torch.empty(used + 64, dtype=torch.uint8, device=self.device)
                ev = torch.cuda.Event()
            else:
                pinned, dev, ev = ent[0], ent[1], ent[2]
            pf, df = pinned[:4 * n_f].view(torch.float32), dev[:4 * n_f].view(torch.float32)
            p_views, d_views, o = [], [], 0
            for t, k in zip(data, sizes):
                p_views.append(pf[o:o + k].view(t.shape))
                d_views.append(df[o:o + k].view(t.shape))
                o += k
            p_y = pinned[off_y:off_y + 8 * n_y].view(torch.int64).view(y.shape)
            d_y = dev[off_y:off_y + 8 * n_y].view(torch.int64).view(y.shape)
            ent = (pinned, dev, ev, sig, p_views, d_views, p_y, d_y, pinned[:used], dev[:used])
            self.slots[i] = ent
        _, _, ev, _, p_views, d_views, p_y, d_y, p_used, d_used = ent
        for pv, t in zip(p_views, data):
            pv.copy_(t)                                          # converts dtype if needed
        p_y.copy_(y)
        self.last_event = None
        if use_cs:
            main = torch.cuda.current_stream(self.device)
            if self.copy_stream is None:
                self.copy_stream = torch.cuda.Stream(device=self.device)
            cs = self.copy_stream
            cs.wait_stream(main)
            with torch.cuda.stream(cs):
                d_used.copy_(p_used, non_blocking=True)
                ev.record(cs)
            if defer_wait:
                self.last_event = ev
            else:
                main.wait_event(ev)
        else:
            d_used.copy_(p_used, non_blocking=True)
            ev.record()
        return list(d_views), d_y


class _LoaderPrefetch:
    """`next(loader)` one batch ahead, on a helper thread (round 6): a stock `torch.utils.data.DataLoader` without workers
    builds every batch on the calling thread - 0.4 - 0.85 ms per 4096-row batch of four partitions (fresh host arrays: page
    faults), a third of an import-swap pipeline's step (bench.py `stock_path`) - while the training loop's own host work
    (packing into the staging ring, the host NaN test, a stock optimizer's step) waits.  numpy's and torch's copies release
    the GIL, so the two overlap.  Only the helper thread ever touches the iterator; an exception inside the loader is
    re-raised where `next()` is called; `close()` (the loop's `finally`) stops the thread at its next batch.
    OPT-IN (`model.prefetch_loader = True` / MMN_PREFETCH=1): measured 1,163 - 1,260 -> 1,073 - 1,078 us per step on the
    import-swap pipeline without shuffling and no gain with it (tools/time_stock.py) - not enough to run a user's dataset
    code on a thread it was not written for by default."""

    def __init__(self, it, depth: int = 2):
        import queue
        import threading
        self.it, self.q, self.stop = it, queue.Queue(maxsize=depth), False
        self.thread = threading.Thread(target=self._run, name="mmn-loader-prefetch", daemon=True)
        self.thread.start()

    def _put(self, item) -> bool:
        import queue
        while not self.stop:
            try:
                self.q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _run(self) -> None:
        try:
            for batch in self.it:
                if not self._put((0, batch)):
                    return
            self._put((1, None))
        except BaseException as ex:                          # noqa: BLE001  (handed to the consumer)
            self._put((2, ex))

    def __iter__(self):
        return self

    def __next__(self):
        kind, val = self.q.get()
        if kind == 0:
            return val
        self.stop = True
        if kind == 1:
            raise StopIteration
        raise val

    def close(self) -> None:
        self.stop = True
        import queue
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=5.0)


_BATCH_CACHE_MAX = 4096


def _stable_batches(loader) -> bool:
    """True if `loader` hands the same batch OBJECTS back every epoch: a list / tuple of batches (it keeps them alive
    itself), or a loader that promises it (`stable_batches`: DeviceResidentLoader without shuffling)."""
    return isinstance(loader, (list, tuple)) or bool(getattr(loader, "stable_batches", False))


class MultiModN(nn.Module):
    def __init__(
            self,
            state_size: int,
            encoders: List[MultiModEncoder],
            decoders: List[MultiModDecoder],
            err_penalty: float,
            state_change_penalty: float,
            shuffle_mode: Optional[bool] = False,
            init_state: Optional[InitState] = None,
            device: Optional[torch.device] = None,
    ):
        super().__init__()
        self.shuffle_mode = shuffle_mode
        self.device = device if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.init_state = init_state if init_state else TrainableInitState(state_size, self.device)
        self.encoders = nn.ModuleList(encoders)
        self.decoders = nn.ModuleList(decoders)
        self.err_penalty = err_penalty
        self.state_change_penalty = 0.01 * state_change_penalty      # multimodn.py:86
        self.to(self.device)
        # not part of the reference surface -----------------------------------------------------
        self._engine = None
        self._engine_factory: Callable = HipChainEngine    # tests may inject a checker backend
        self._dp_group = None
        self._dp_world = 1
        self._stager: Optional[_HostStager] = None
        #: "host": decide NaN-skips on the host like the reference (exact grad=None semantics, one readback per step);
        #: "device": keep the decision on the GPU (no sync; a skipped encoder's .grad reads zeros instead of None);
        #: "auto" (default): "device" whenever the result is the reference's anyway - forward-only entry points, and
        #: training with multimodn_amd.optim.Adam, which leaves a skipped encoder's parameters, moments and step counts
        #: untouched exactly as torch does for grad None - and "host" for every other optimizer
        self.nan_policy = "auto"
        #: build-defined extension (BASELINE.json configs[4]): every SAMPLE may miss modalities (NaN
        #: rows) and carry its own encoder order; the batch result is the mean over the samples of the
        #: reference's batch-size-1 result (the only case the reference defines, multimodn.py:168,518-523)
        self.per_sample = False
        # tests: callable (encoder id, batch, width) -> [batch, width] multipliers replacing the device draw
        # of the MIMIC encoders' dropout masks (parity runs feed the masks the reference drew); it is asked when a step
        # is LAUNCHED - train_steps_launched is then the index of that step (batches are ingested ahead of their launch)
        self.dropout_mask_provider = None
        self.train_steps_launched = 0
        #: single GPU, device-side NaN decision, multimodn_amd.optim.Adam: steps whose device buffers were seen before are
        #: captured into a hipGraph once and replayed afterwards (engine.run_group) - groups of REPLAY_GROUP steps when
        #: the batches are device-resident (any batch size), single steps of at most REPLAY_MAX_ROWS rows when they are
        #: staged from the host (the reference pipelines train with 16-32 rows: host-bound otherwise)
        self.replay_steps = True
        self.REPLAY_MAX_ROWS = 256
        #: False (default): every train_epoch / test / predict / get_states call rebuilds the kernels' fragment-order copies
        #: of the weights from the parameters, as the reference re-reads its parameters on every call - one small launch per
        #: call.  True: skip that while torch's `_version` counters, the library's write counter and the storage addresses
        #: say nobody wrote the parameters since the last call; writers those cannot see (`p.data.mul_(..)`, raw pointers)
        #: must then call `model._engine.invalidate_weights()` themselves.
        self.trust_param_versions = False

    # nn.Module pickling: the engine holds raw device handles and is rebuilt on demand
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        state.pop("_ps_stream", None)
        state["_batch_cache"] = {}
        state.pop("_eval_batch_cache", None)
        state["_epoch_plans"] = {}
        state["_small_epochs"] = {}                          # (batch descriptors: ctypes structs with raw device pointers)
        state["_dp_group"] = None
        state["_dp_world"] = 1
        state["_dp_rank"] = 0
        state["_stager"] = None
        return state

    # ------------------------------------------------------------------------------------------
    #: divisor every rank uses when shards may be uneven: a power of two (exact division), corrected to the true
    #: global batch after the all-reduce (engine.dp_rescale)
    DP_NOMINAL_BATCH = 1 << 16

    def enable_data_parallel(self, process_group=None, uneven_shards: bool = False, oneshot: Optional[bool] = None) -> None:
        """One process per GPU: every rank feeds its shard of each global mini-batch; gradients and
        the per-step statistics are summed with ONE all-reduce (RCCL over xGMI) per step.

        By default every rank must feed the same number of rows per step (global batch = rows x world: what
        DistributedSampler's padding guarantees).  `uneven_shards=True` lifts that (a last batch of 4096 + 4095 + ...
        rows; every rank still needs at least one row per step): all ranks divide by the same nominal batch and the
        true global row count, which rides in the step's one all-reduce anyway, corrects the means afterwards - one
        more small launch per step, no further collective."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dp_nominal = int(self.DP_NOMINAL_BATCH) if uneven_shards else 0
        self._dp_group = process_group if process_group is not None else dist.group.WORLD
        self._dp_world = dist.get_world_size(self._dp_group)
        self._dp_rank = dist.get_rank(self._dp_group)
        # `oneshot` (default: the environment's MMN_DP_ONESHOT=1, else off): training steps with multimodn_amd.optim.Adam
        # exchange [grads | stats] through buffers every rank's process has mapped (hipIpc; xGMI between GPUs) INSIDE the
        # launch that applies Adam, instead of one RCCL all-reduce + that launch: no collective per step, one launch less.
        # Even shards only, one node (<= 8 ranks).  Opt-in until a multi-GPU box has run it.
        if oneshot is None:
            oneshot = os.environ.get("MMN_DP_ONESHOT", "0") not in ("", "0")
        self._dp_oneshot = bool(oneshot) and not uneven_shards and self._dp_world <= 8

    def _get_engine(self, batch: int):
        if self._engine is None:
            self._engine = self._engine_factory(self, max(int(batch), 1))
        self._engine.ensure(int(batch))
        if hasattr(self._engine, "set_per_sample") and (self.per_sample or self._engine._generic_tier):
            self._engine.set_per_sample(bool(self.per_sample))   # (a plan whose kernels take regrouped tiles; re-plans when that changes)
        # data parallel: every rank draws ITS rows' dropout multipliers from its own stream (same torch seed on all
        # ranks is the usual discipline; identical multipliers on different rows would correlate the shards)
        self._engine.dropout_salt = int(getattr(self, "_dp_rank", 0)) if self._dp_group is not None else 0
        return self._engine

    def get_encoder_iterable(self, encoder_sequence, shuffle_mode: bool, train: bool) -> List[Tuple[int, int]]:
        """(data_idx, enc_idx) pairs for one batch (multimodn.py:509-531)."""
        if encoder_sequence is None:
            pairs = [(i, i) for i in range(len(self.encoders))]
        else:
            seq = encoder_sequence.cpu().numpy() if isinstance(encoder_sequence, Tensor) else np.asarray(encoder_sequence)
            first = seq[0]
            if not (seq == first).all():
                raise ValueError("Encoder sequence has different values across the batch. "
                                 "Hint: set batch size to 1 to avoid this error.")
            pairs = [(k, int(e)) for k, e in enumerate(first)]
        if shuffle_mode and train:
            random.shuffle(pairs)
        return pairs

    # ------------------------------------------------------------------------------------------
    def _nan_mode(self, eng, optimizer, train: bool, desc=None) -> str:
        """How a step decides which encoders a NaN batch skips (multimodn.py:168):
        "device"  on the GPU, nothing is read back (a skipped encoder's .grad then reads zeros instead of None: harmless for
                  multimodn_amd.optim.Adam, whose fused step leaves such an encoder untouched exactly as torch does for
                  grad None, and for the forward-only entry points);
        "host"    on the host before the step (skipped encoders leave the sequence; their .grad is None): the
                  reference's exact semantics for ANY optimizer, one readback per step when the batch lives on the device;
        "readback" on the GPU, and the executed rows are read back after the step so that skipped encoders still get grad
                  None: what "host" becomes under data parallel, where the decision belongs to the GLOBAL batch and the
                  flags travel inside the step's one all-reduce.
        nan_policy "auto" picks "device" only if that gives the reference's result: forward-only, or the optimizer's step
        really is fused with this engine (a multimodn_amd.optim.Adam over exactly this model's parameters in one group;
        with several groups, a subset or another order fused_descriptor() is None and the host policy applies)."""
        policy = getattr(self, "nan_policy", "auto")
        if policy == "auto":
            fused = train and hasattr(optimizer, "fused_descriptor") and hasattr(eng, "adam_fusable") and \
                eng.adam_fusable(optimizer, desc)
            policy = "device" if (not train or fused) else "host"
        if policy == "host" and self._dp_group is not None:
            return "readback"
        return policy

    def _to_device(self, data: Sequence[Tensor], target, defer_wait: bool = False):
        """multimodn.py:132-135: the batch on the model's device, float32 features / int64 targets.  Host batches go
        through one pinned staging buffer and ONE copy (on the stager's copy stream; `defer_wait`: _HostStager.stage)."""
        def here(t):                                        # (a device without an index means the current one)
            return t.device.type == self.device.type and (self.device.index is None or t.device.index == self.device.index)
        if self.device.type != "cpu" and isinstance(target, Tensor) and here(target) and target.dtype == torch.int64 \
                and target.dim() == 2 and target.is_contiguous() \
                and all(here(t) and t.dtype == torch.float32 and t.is_contiguous() for t in data):
            return list(data), target, False                # device-resident batch: nothing to move, nothing to convert
        if not isinstance(target, Tensor):
            target = torch.as_tensor(np.asarray(target))
        if target.dim() == 1:
            target = target.view(-1, 1)
        on_host = all(t.device.type == "cpu" for t in data) and target.device.type == "cpu"
        if on_host and self.device.type == "cuda" and data:
            # Host tensors of this size are copied / scanned by torch's intra-op pool; on a many-core
            # host (128 threads here) that pool turns a 100 us copy into milliseconds, so the handful of
            # small host ops of one batch run with at most 16 threads (model.stage_threads).
            # (torch runs tensors below its grain size - 32,768 elements - on the calling thread anyway: the reference
            #  pipelines' 16-row batches skip the two set_num_threads calls, which cost more than their copies)
            big = sum(int(t.numel()) for t in data) >= 32768
            n_thr = torch.get_num_threads() if (big and not self.__dict__.get("_threads_capped")) else 0
            cap = int(getattr(self, "stage_threads", 16))     # (tools/time_h2d.py, EPYC 9575F: 4 / 8 / 16 / 32 threads -> 121 / 123 / 114 / 149 us per step)
            if n_thr > cap:
                torch.set_num_threads(cap)
            try:
                if self._stager is None:
                    self._stager = _HostStager(self.device)
                xs, y = self._stager.stage(data, target.to(torch.int64), defer_wait)
            finally:
                if n_thr > cap:
                    torch.set_num_threads(n_thr)
        else:
            xs = [t.to(self.device, dtype=torch.float32, non_blocking=True).contiguous() for t in data]
            y = target.to(torch.int64).to(self.device, non_blocking=True).contiguous()
        return xs, y, on_host

    def _ingest(self, data: Sequence[Tensor], target, pairs, mode: str = "host", defer_wait: bool = False):
        """Host half of multimodn.py:132-135,168: move the batch to the device and, under the host policy, decide the
        NaN skips.  Returns (xs_dev, y_dev, executed_pairs, executed list or None, batch came from the host)."""
        present: Optional[List[bool]] = None
        if mode == "host":
            on_host = all(t.device.type == "cpu" for t in data)
            n_thr = torch.get_num_threads() if (on_host and not self.__dict__.get("_threads_capped")
                                                and sum(int(t.numel()) for t in data) >= 32768) else 0
            if n_thr > 8:
                torch.set_num_threads(8)
            try:
                if on_host or not pairs:
                    present = [not _has_nan_host(data[k]) for k, _ in pairs]
                else:                                                # device tensors: ONE readback for all slots
                    flags = torch.stack([torch.isnan(data[k]).any() for k, _ in pairs]).tolist()
                    present = [not f for f in flags]
            finally:
                if n_thr > 8:
                    torch.set_num_threads(n_thr)
        xs, y, on_host = self._to_device(data, target, defer_wait)
        if present is None:
            return xs, y, list(pairs), None, on_host
        exec_pairs = [pe for pe, ok in zip(pairs, present) if ok]
        executed = [False] * len(self.encoders)
        for _, e in exec_pairs:
            executed[e] = True
        return xs, y, exec_pairs, executed, on_host

    def _fusion_setup(self, eng, optimizer, mode: str):
        """optimizer.fused_descriptor(eng) if the engine can apply this optimizer's step itself (and the NaN decision
        stays on the device), else None.  In the first case every .grad is pointed at its slice of the engine's flat
        gradient buffer once - where loss.backward() would have left it - instead of once per step.  To be called again
        whenever the engine re-planned: the descriptor names the plan's gradient buffer."""
        if not (hasattr(optimizer, "fused_descriptor") and hasattr(eng, "lib")):
            return None
        fd = optimizer.fused_descriptor(eng)
        if fd is None or not eng.adam_fusable(optimizer, fd):
            return None
        if mode == "device" and (eng.params[0].grad is not eng.grad_views[0] or eng.params[-1].grad is not eng.grad_views[-1]):
            eng.assign_grads(None)
        return fd

    def _dp_group_tail(self, eng, optimizer):
        """(key, fn, process group) for engine.run_group when a data-parallel step can sit in a captured group, else None: the one-shot
        exchange (its launch is the library's own), or torch's RCCL all-reduce inside the capture (torch's NCCL binding is
        capturable; round 4, one rank on one GPU with real RCCL launches: 62.1 us/step in captured groups of 8 steps against
        71.0 us eager - the eager data-parallel step is host-bound; MMN_DP_GRAPH=0 keeps the steps eager)."""
        if getattr(self, "_dp_nominal", 0) or not hasattr(eng, "run_group"):
            return None
        alpha, beta = float(self.err_penalty), float(self.state_change_penalty)
        if self._oneshot_ready(eng):
            return ("oneshot", lambda d: eng.accumulate_and_step_oneshot(alpha, beta, optimizer, desc=d), self._dp_group)
        # OPT-IN (MMN_DP_GRAPH=1) until a box with more than one GPU has run it: no test here can put two RCCL ranks on the
        # one GPU of a box (RCCL refuses duplicate devices), so the capture of an all-reduce has only met a one-rank
        # communicator.  The default data-parallel step is eager: the protocol the 2- and 8-rank tests verify.
        if os.environ.get("MMN_DP_GRAPH", "0") not in ("", "0"):
            import torch.distributed as dist
            if dist.get_backend(self._dp_group) == "nccl":
                def tail(d):
                    dist.all_reduce(eng.reduce_buf, group=self._dp_group)
                    return eng.accumulate_and_step(alpha, beta, optimizer, desc=d)
                return ("rccl", tail, self._dp_group)
        return None

    def _oneshot_ready(self, eng) -> bool:
        """One-shot exchange requested and possible with this engine: attaches it (a collective: every rank gets here in
        the same step, the ranks' plans being the same) the first time and after a re-plan."""
        if not getattr(self, "_dp_oneshot", False) or not hasattr(eng, "attach_oneshot"):
            return False
        if not eng.oneshot_attached():
            if getattr(eng, "_oneshot_refused", None) == eng._plan.value:   # (every rank refused together: the all-reduce path)
                return False
            if not eng.attach_oneshot(self._dp_group, self._dp_world, self._dp_rank,
                                      spin_ms=int(os.environ.get("MMN_DP_SPIN_MS", "5000"))):
                return False
        eng.oneshot_check()
        return True

    def _global_rows(self, local_rows: int) -> int:
        """The divisor of this step's means: the global batch (local rows x ranks), or the nominal batch all ranks
        agree on when shards may be uneven (enable_data_parallel)."""
        nominal = getattr(self, "_dp_nominal", 0) if self._dp_group is not None else 0
        return nominal if nominal else local_rows * self._dp_world

    def _dp_all_reduce(self, buf: Tensor) -> None:
        """Sum over the ranks, in place.  RCCL takes device buffers; any other backend (gloo in the tests) gets the
        buffer through the host."""
        import torch.distributed as dist
        if buf.is_cuda and dist.get_backend(self._dp_group) != "nccl":
            host_buf = buf.cpu()
            dist.all_reduce(host_buf, group=self._dp_group)
            buf.copy_(host_buf)
        else:
            dist.all_reduce(buf, group=self._dp_group)

    class _Step:
        """One ingested mini-batch waiting for its launch."""
        __slots__ = ("xs", "y", "pairs", "executed", "on_host", "bg", "b", "key", "masks", "cached", "stepped", "ready")

        def __init__(self, xs, y, pairs, executed, on_host, bg, cached=None):
            self.xs, self.y, self.pairs, self.executed, self.on_host, self.bg = xs, y, pairs, executed, on_host, bg
            self.b, self.key, self.masks, self.cached, self.stepped = None, None, None, cached, False
            self.ready = None                               # the staging copy's event, when the wait for it was left to the launch

        def key_tuple(self):
            return (self.xs, self.y, self.pairs, self.bg, self.b, self.key)

    def _make_step(self, data, target, encoder_sequence, mode: str, train: bool, defer_wait: bool = False) -> "MultiModN._Step":
        pairs = self.get_encoder_iterable(encoder_sequence, self.shuffle_mode, train=train)
        xs, y, exec_pairs, executed, on_host = self._ingest(data, target, pairs, mode, defer_wait)
        st = MultiModN._Step(xs, y, exec_pairs, executed, on_host, self._global_rows(int(y.shape[0])))
        if defer_wait and on_host and self._stager is not None:
            st.ready, self._stager.last_event = self._stager.last_event, None
        return st

    def _launch_step(self, eng, st: "MultiModN._Step", nxt: Optional["MultiModN._Step"], train: bool, optimizer, mode: str,
                     desc=None):
        """The launches of ONE step (multimodn.py:137-204 without optimizer.step() unless it is fused): returns the
        `executed` list for assign_grads (None = every encoder has a gradient buffer)."""
        dp = self._dp_group is not None
        b = st.b
        alpha, beta = float(self.err_penalty), float(self.state_change_penalty)
        if dp and b.nan_flags and eng._prescanned is not b:
            # data parallel: the skip decision belongs to the GLOBAL batch (multimodn.py:168 looks at the whole batch).
            # Every batch but the first of an epoch was pre-scanned by the step before it and its flags rode in that
            # step's all-reduce; this one is scanned on its own and its flags are summed now.
            eng.nan_scan(b)
            self._dp_all_reduce(eng.flag_tail)
        if train and eng.dropout_encoders:                  # nn.Dropout of the MIMIC encoders is live in train mode only
            st.masks = eng.draw_dropout_masks(b, self.dropout_mask_provider)
        st.stepped = False
        if train:
            # single GPU + multimodn_amd.optim.Adam: optimizer.step() rides in the last launch
            fuse = optimizer if (not dp and hasattr(optimizer, "fused_descriptor")) else None
            # (the next step of this loop draws its dropout multipliers on the device too: this step's last launch does it)
            predraw = bool(nxt is not None and eng.dropout_encoders and self.dropout_mask_provider is None)
            st.stepped = eng.local_step(b, alpha, beta, accumulate=not dp, optimizer=fuse,
                                        next_batch=None if nxt is None else nxt.b, predraw_next=predraw,
                                        **({"desc": desc} if desc is not None else {}))
        else:
            eng.eval_step(b, accumulate=not dp)
        if dp and train and optimizer is not None and self._oneshot_ready(eng) and \
                eng.accumulate_and_step_oneshot(alpha, beta, optimizer, **({"desc": desc} if desc is not None else {})):
            st.stepped = True                               # exchange + rank-ordered sum + Adam + accumulation: ONE launch
        elif dp:
            self._dp_all_reduce(eng.reduce_buf if train else eng.stats)     # THE collective of the step: grads + stats + flags
            if getattr(self, "_dp_nominal", 0):
                eng.dp_rescale(self._dp_nominal, with_grads=train)
            if train and optimizer is not None:
                # epoch accumulation + Adam in one launch when the optimizer is multimodn_amd.optim.Adam
                # (it leaves the parameters of encoders that did not run untouched, like grad None)
                st.stepped = eng.accumulate_and_step(alpha, beta, optimizer, **({"desc": desc} if desc is not None else {}))
            else:
                eng.accumulate(alpha if train else 1.0, beta if train else 0.0)
        if train:
            self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + 1
        if mode == "readback":                              # exact grad-None semantics for a foreign optimizer
            return list(eng.executed_rows()[1:])
        return st.executed

    def _run_step(self, eng, data, target, encoder_sequence, train: bool, batch_global: Optional[int] = None,
                  optimizer=None):
        """One step on its own (no look-ahead): what test() / the parity tests / smoke() drive."""
        mode = self._nan_mode(eng, optimizer, train)
        st = self._make_step(data, target, encoder_sequence, mode, train)
        if batch_global:
            st.bg = int(batch_global)
        st.b = eng.make_batch(st.xs, st.y, st.pairs, batch_global=st.bg, device_nan_flags=st.executed is None)
        executed = self._launch_step(eng, st, None, train, optimizer, mode)
        return executed, (st.xs, st.y, st.masks)

    def _regroup_per_sample(self, eng, data, target, encoder_sequence):
        """Batch -> device -> tiles of one executed sequence each (engine.per_sample_batch), on the CURRENT stream."""
        xs, y, _ = self._to_device(data, target)
        seq = None
        if encoder_sequence is not None:
            seq = encoder_sequence if isinstance(encoder_sequence, Tensor) else torch.as_tensor(np.asarray(encoder_sequence))
            seq = seq.to(self.device, torch.int64)
        b, keep = eng.per_sample_batch(xs, y, seq)
        b.batch_global = self._global_rows(int(y.shape[0]))
        return b, keep, xs, y

    def _run_step_per_sample(self, eng, data, target, encoder_sequence, optimizer=None, train: bool = True, regrouped=None,
                             desc=None):
        """One training step in per-sample mode: rows are regrouped on the device into tiles of one
        executed sequence each (engine.per_sample_batch) and run by the fused kernel.  `regrouped`: the result of
        _regroup_per_sample when the caller already ran it (train_epoch does, one batch ahead, on a side stream)."""
        b, keep, xs, y = regrouped if regrouped is not None else self._regroup_per_sample(eng, data, target, encoder_sequence)
        dp = self._dp_group is not None
        alpha, beta = float(self.err_penalty), float(self.state_change_penalty)
        if not train:                                       # forward-only (test / predict / get_states)
            eng.eval_step(b, accumulate=not dp)
            if dp:
                self._dp_all_reduce(eng.stats)
                if getattr(self, "_dp_nominal", 0):
                    eng.dp_rescale(self._dp_nominal, with_grads=False)
                eng.accumulate(1.0, 0.0)
            return None, (xs, y, keep)
        fuse = optimizer if (not dp and hasattr(optimizer, "fused_descriptor")) else None
        if eng.dropout_encoders:
            # nn.Dropout of the MIMIC encoders (train mode): multipliers per REGROUPED row, drawn on the device for the
            # padded batch; a test's provider speaks in original rows - its masks are carried to where the rows went
            prov = self.dropout_mask_provider
            if prov is not None:
                where, _ = eng.per_sample_positions()
                rows = int(b.batch)

                def regrouped(e, n_rows, width, _prov=prov, _where=where, _n=int(y.shape[0])):
                    mk = _prov(e, _n, width)
                    if mk is None:
                        return None
                    out = torch.ones((n_rows, width), dtype=torch.float32, device=self.device)
                    out[_where] = mk.to(self.device, torch.float32)
                    return out
                prov = regrouped
            keep = (keep, eng.draw_dropout_masks(b, prov))
        eng.local_step(b, alpha, beta, accumulate=not dp, optimizer=fuse, **({"desc": desc} if desc is not None else {}))
        if dp and optimizer is not None and self._oneshot_ready(eng) and eng.accumulate_and_step_oneshot(alpha, beta, optimizer):
            pass
        elif dp:                                            # per-sample masks / sequences are per-row data: shards add up
            self._dp_all_reduce(eng.reduce_buf)
            if getattr(self, "_dp_nominal", 0):
                eng.dp_rescale(self._dp_nominal)
            if optimizer is not None:
                eng.accumulate_and_step(alpha, beta, optimizer)
            else:
                eng.accumulate(alpha, beta)
        return None, (xs, y, keep)

    #: steps captured into one hipGraph when every batch of the group already lives on the device
    REPLAY_GROUP = 8
    #: ... and behind the first group of a call (0 = the same).  A graph-to-graph boundary costs a few us of idle GPU (a
    #: boundary between two kernels of one graph ~1.2) and a long graph's launch is hidden behind the group in front of it.
    #: Round 4, one box, C3, the SAME call repeated (tools/ovh_sweep.sh): (8, 0) 55.05 us/step over 20-step calls / 53.36
    #: over 200-step calls; (8, 12) 54.81 / 53.17; (8, 16) 54.79 / 52.97; (8, 24) 54.81 / 52.94; (4, 16) 54.77 / 53.06 -
    #: but a 20-step call that follows a DIFFERENT call (bench.py: warm-up call, then the timed one) takes 57.7 - 67 us/step
    #: with (8, 16) against 56.1 - 56.3 with (8, 0): a big graph launched for the first time after another one pays more
    #: than the boundaries it saves.  Off by default.
    REPLAY_GROUP_NEXT = 0

    def _train_steps(self, train_loader, optimizer, log_interval=None, logger=None):
        """The batch loop of train_epoch (multimodn.py:117-212).  Batches are ingested one step ahead of their launch
        (a group ahead when they are device-resident), so that (a) a step's last launch pre-scans the NEXT batch for
        NaNs - no scan launch in front of the next chain kernel; under data parallel the flags ride in the step's ONE
        all-reduce - and (b) recurring groups of steps are replayed as one hipGraph (engine.run_group).
        Returns (engine, number of steps run)."""
        import collections
        import itertools
        it = iter(train_loader)
        prefetch = None
        if isinstance(train_loader, DataLoader) and getattr(train_loader, "num_workers", 1) == 0 and self.device.type == "cuda" \
                and (getattr(self, "prefetch_loader", False) or os.environ.get("MMN_PREFETCH", "0") not in ("", "0")):
            it = prefetch = _LoaderPrefetch(it)              # (a loader with workers prefetches by itself)
        # torch's intra-op pool for the loop's host-side copies / NaN tests: capped ONCE per call (`_to_device` used to set and
        # restore it around every batch: two pool resizes per step - on a 128-thread host one of them now and then takes a
        # millisecond, and a 96-step measurement of the host path read 745 us per step instead of 110)
        n_thr, cap = torch.get_num_threads(), int(getattr(self, "stage_threads", 16))
        capped = self.device.type == "cuda" and n_thr > cap
        if capped:
            torch.set_num_threads(cap)
            self.__dict__["_threads_capped"] = True
        try:
            return self._train_steps_loop(train_loader, it, optimizer, log_interval, logger)
        finally:
            if capped:
                self.__dict__["_threads_capped"] = False
                torch.set_num_threads(n_thr)
            if prefetch is not None:
                prefetch.close()

    def _train_steps_loop(self, train_loader, it, optimizer, log_interval=None, logger=None):
        import collections
        import itertools
        n_batches = len(train_loader) if hasattr(train_loader, "__len__") else None
        window: "collections.deque[MultiModN._Step]" = collections.deque()
        cache = self.__dict__.setdefault("_batch_cache", {})
        # Only loaders that hand the SAME batch objects back every epoch are cached (a python list / tuple of batches,
        # which holds them alive anyway; a loader that says so: DeviceResidentLoader without shuffling).  Anything else -
        # a shuffling device loader yields fresh index_select tensors every step - would only pin its batches in HBM.
        stable = _stable_batches(train_loader)
        state = {"eng": None, "mode": None, "done": False, "steps": 0, "grads_assigned": False, "fd": None}
        dp = self._dp_group is not None
        plan_ok = stable and not dp and not log_interval and not self.shuffle_mode and getattr(self, "replay_steps", True) \
            and self.dropout_mask_provider is None and optimizer is not None
        if plan_ok:
            done = self._small_epoch(train_loader, optimizer)
            if done is None:
                done = self._replay_epoch_plan(train_loader, optimizer)
            if done is not None:
                return done
        rec: List[tuple] = []                                # the groups this call replays, in order (-> _epoch_plans)
        seen_batches: List[object] = []

        def pull() -> bool:
            if state["done"]:
                return False
            try:
                batch = next(it)
            except StopIteration:
                state["done"] = True
                return False
            seen_batches.append(batch)
            data, target, encoder_sequence = (list(batch) + [None])[:3]
            if state["eng"] is None:
                eng = state["eng"] = self._get_engine(int(target.shape[0]))
                state["need_reset"] = True                   # (in front of the first launch: inside its graph when it is a replay)
                eng.begin_sequence()
                # (the optimizer's descriptor walks every parameter: once per epoch, not once per step)
                state["mode"] = self._nan_mode(eng, optimizer, True)
                state["sig"] = (state["mode"], self._dp_world, self.shuffle_mode)
                state["fd"] = self._fusion_setup(eng, optimizer, state["mode"])
                state["grads_assigned"] = state["fd"] is not None and state["mode"] == "device"
            # Loaders over device-resident data hand the SAME batch objects back every epoch (DeviceResidentLoader
            # without shuffling, a list of device batches): what was derived from their tensors' addresses last time
            # - the sequence, the filled-in mmn_batch struct, its cache key - is reused; checking a batch costs less host
            # time than a small step's launches.  (Only addresses are kept, never values: the tensors' contents may change.)
            ent = cache.get(id(batch)) if (stable and encoder_sequence is None) else None
            if ent is not None and ent[0] is batch and ent[1] is target and ent[3] == state["sig"] \
                    and len(ent[2]) == len(data) and all(a is c for a, c in zip(ent[2], data)):
                window.append(MultiModN._Step(ent[2], target, ent[4], None, False, ent[5], ent))
                return True
            st = self._make_step(data, target, encoder_sequence, state["mode"], True, defer_wait=True)
            if st.on_host or st.y is not target or len(st.xs) != len(data) or not all(a is c for a, c in zip(st.xs, data)):
                state["ingested"] = True                     # (a copy / conversion per call: never part of a whole-call plan)
            if stable and encoder_sequence is None and st.executed is None and not st.on_host and not self.shuffle_mode \
                    and isinstance(batch, tuple) and st.y is target and all(a is c for a, c in zip(st.xs, data)):
                if len(cache) >= _BATCH_CACHE_MAX:
                    cache.clear()
                st.cached = cache[id(batch)] = [batch, target, st.xs, state["sig"], st.pairs, st.bg, None]
            window.append(st)
            return True

        staged: Dict[int, tuple] = {}                        # device tensors of the staging ring -> their filled-in struct

        def materialise() -> None:
            """hip.Batch structs for everything in the window (in order: the flag sets alternate); a batch larger than
            the plan re-plans first, which invalidates the structs made so far."""
            eng = state["eng"]
            need = max(int(st.y.shape[0]) for st in window)
            if need > eng.max_batch and eng.ensure(need):
                for st in window:
                    st.b = None
                eng.begin_sequence()
                state["fd"] = self._fusion_setup(eng, optimizer, state["mode"])      # (it names the plan's gradient buffer)
                state["grads_assigned"] = state["fd"] is not None and state["mode"] == "device"
            for st in window:
                if st.b is None:
                    ent = st.cached
                    tmpl = None if ent is None else ent[6]
                    if ent is None and st.on_host:
                        # a staged host batch: the staging ring hands the SAME device tensors back every `depth` batches; the
                        # struct filled for them is reused as long as sequence and global batch are the same
                        sl = staged.get(id(st.y))
                        if sl is not None and sl[0] is st.y and len(sl[1]) == len(st.xs) and all(a is c for a, c in zip(sl[1], st.xs)) \
                                and sl[2] == (tuple(st.pairs), st.bg):
                            tmpl = sl[3]
                    st.b, st.key, tk = eng.make_batch_keyed(st.xs, st.y, st.pairs, st.bg, st.executed is None, tmpl)
                    if ent is not None:
                        ent[6] = tk
                    elif st.on_host:
                        if len(staged) > 16:
                            staged.clear()
                        staged[id(st.y)] = (st.y, list(st.xs), (tuple(st.pairs), st.bg), tk)

        while True:
            if not window and not pull():
                break
            eng, mode = state["eng"], state["mode"]
            # how far to look ahead: device-resident batches by a whole group (+ the batch the group's last step
            # pre-scans); batches staged from the host by one (the staging ring is three deep)
            resident = not window[0].on_host
            fused_surface = mode == "device" and state["fd"] is not None
            # data parallel: groups are captured too when the exchange is the one-shot kernel (nothing but this library's
            # launches in the graph), or - MMN_DP_GRAPH=1 - with torch's all-reduce inside the capture
            dp_tail = self._dp_group_tail(eng, optimizer) if (dp and mode == "device" and state["fd"] is not None) else None
            can_replay = ((not dp or dp_tail is not None) and mode == "device" and getattr(self, "replay_steps", True)
                          and not log_interval and self.dropout_mask_provider is None and hasattr(eng, "run_group")
                          and optimizer is not None)
            group = 1
            # (sending the first step of a sequence out on its own, so that the GPU works while the first group is being
            #  ingested, was measured: 77.6 instead of 74.9 us/step over 20 steps - the group's replay then starts late)
            if can_replay and resident:
                group = max(1, int(getattr(self, "REPLAY_GROUP", 8)))
                if state["steps"] > 0 and int(getattr(self, "REPLAY_GROUP_NEXT", 0)) > 0:
                    group = max(1, int(self.REPLAY_GROUP_NEXT))
            # (stop at the first batch that arrives the other way - host-staged behind device-resident or vice versa: the
            #  staging ring is three deep, a longer window of host batches would overwrite buffers of steps not yet launched)
            while len(window) < group + 1 and pull():
                if window[-1].on_host != window[0].on_host:
                    break
            materialise()
            n = min(group, len(window))
            if any(st.on_host != window[0].on_host for st in itertools.islice(window, n)):
                n = 1
            small = int(window[0].y.shape[0]) <= getattr(self, "REPLAY_MAX_ROWS", 256)
            # (a replayed SINGLE step pays where the step is host-bound - measured: 50 -> 39 us/step at 32 rows, break-even
            #  at 512 - a group of 8 pays at any batch size: one host submission for 8 steps)
            # host batches: their copies run on the staging ring's own stream; the step waits for ITS batch here, in front of
            # its launches.  A replayed group's last launch also pre-scans the batch behind the group (its graph holds that
            # batch's buffers): it waits for that copy too - groups of host batches are the small ones (<= REPLAY_MAX_ROWS rows).
            def await_copy(st_):
                if st_.ready is not None:
                    torch.cuda.current_stream(self.device).wait_event(st_.ready)
                    st_.ready = None
            for st in itertools.islice(window, n):
                await_copy(st)
            if can_replay and (n > 1 or small) and len(window) > n:
                await_copy(window[n])
            if can_replay and (n > 1 or small):
                steps = [st.key_tuple() for st in itertools.islice(window, n)]
                nxt = window[n].key_tuple() if len(window) > n else None
                if not state["grads_assigned"]:
                    eng.assign_grads(None)
                    state["grads_assigned"] = True
                reset_now = bool(state.get("need_reset"))
                if dp_tail is not None and window[0].b.nan_flags and eng._prescanned is not window[0].b:
                    # the group's first batch has no predecessor whose exchange carried its NaN flags (first batch of an
                    # epoch): scanned on its own and summed over the ranks now, in front of the group
                    eng.nan_scan(window[0].b)
                    self._dp_all_reduce(eng.flag_tail)
                ent_now = eng.group_entry(steps, nxt, float(self.err_penalty), float(self.state_change_penalty), optimizer,
                                          bool(eng.dropout_encoders), state["fd"], reset_now)[0] if plan_ok else None
                if eng.run_group(steps, nxt, float(self.err_penalty), float(self.state_change_penalty), optimizer,
                                 bool(eng.dropout_encoders), state["fd"], reset_first=reset_now, dp_tail=dp_tail):
                    state["need_reset"] = False
                    rec.append((steps, nxt, ent_now))
                    for _ in range(n):
                        window.popleft()
                    state["steps"] += n
                    self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + n
                    optimizer.fused_step_seen(n)             # what n calls of optimizer.step() would do now
                    continue
            # eagerly: the whole group (first sighting of its buffers: the groups of later epochs then start at the same
            # positions), or the single step
            plan_ok = False                                  # (a step outside a replayed group: no plan from this call)
            if state.get("need_reset"):
                eng.epoch_reset()
                state["need_reset"] = False
            for _ in range(n):
                st = window.popleft()
                await_copy(st)
                nxt = window[0] if window else None
                if nxt is not None and nxt.ready is not None:
                    # the next batch's copy is still on its way (on the copy stream): this step's last launch must not pre-scan
                    # it - the next step then scans its own batch in front of its chain (one small launch), and the copy
                    # overlaps this whole step.  Data parallel keeps the pre-scan (its flags ride in this step's ONE
                    # all-reduce): there this step waits for that copy.
                    if dp:
                        await_copy(nxt)
                    else:
                        nxt = None
                if fused_surface and state["grads_assigned"]:
                    # the engine applies optimizer.step() itself (multimodn_amd.optim.Adam): zero_grad / .grad / step()
                    # would only re-point 31 tensors and cross the Optimizer hooks, host time a small step does not have
                    executed = self._launch_step(eng, st, nxt, True, optimizer, mode, state["fd"])
                    if st.stepped:
                        optimizer.fused_step_seen()
                    else:                                    # the library refused the fusion after all
                        eng.assign_grads(executed)
                        optimizer.step()
                else:
                    optimizer.zero_grad()
                    executed = self._launch_step(eng, st, nxt, True, optimizer, mode, state["fd"])
                    eng.assign_grads(executed)          # what loss.backward() leaves behind (multimodn.py:203)
                    optimizer.step()
                    state["grads_assigned"] = executed is None
                state["steps"] += 1
                batch_idx = state["steps"] - 1
                if log_interval and batch_idx % log_interval == log_interval - 1:
                    v = eng.step_values()
                    logger(f"Batch {batch_idx + 1}/{n_batches}\n"
                           f"\tLoss: {float(v['loss']):.4f}\n"
                           f"\tErr loss: {float(v['global_err']):.4f}\n"
                           f"\tState change: {float(v['global_sc']):.4f}")
        if state["eng"] is not None and state["fd"] is not None and state["mode"] == "device" and not dp:
            # every step of this call applied the optimizer inside the library (the copies the chain kernels read were
            # scattered with each update): the next call need not repack unless somebody writes the parameters in between
            state["eng"].note_parameters_current()
        if plan_ok and not state.get("ingested") and state["eng"] is not None and rec \
                and sum(len(g[0]) for g in rec) == state["steps"] == len(seen_batches):
            # every step of this call ran inside a replayed group: the next call over the same batch objects skips the
            # ingest altogether (_replay_epoch_plan)
            plans = self.__dict__.setdefault("_epoch_plans", {})
            if len(plans) >= 4:
                plans.pop(next(iter(plans)))
            eng = state["eng"]
            plans[(len(seen_batches), id(seen_batches[0]), id(seen_batches[-1]))] = {
                "batches": seen_batches, "groups": rec, "eng": eng, "plan": eng._plan.value, "opt": optimizer,
                "group": (int(getattr(self, "REPLAY_GROUP", 8)), int(getattr(self, "REPLAY_GROUP_NEXT", 0))), "rows": max(int(st_[1].shape[0]) for g in rec for st_ in g[0])}
            # (an entry captured DURING this call was looked up before its capture: take it from the cache now)
            plans[(len(seen_batches), id(seen_batches[0]), id(seen_batches[-1]))]["groups"] = [
                (g[0], g[1], g[2] if (g[2] is not None and g[2][1] is not None) else None) for g in rec]
        return state["eng"], state["steps"]

    def _small_epoch(self, train_loader, optimizer):
        """The whole batch loop as ONE launch (mmn_train_epoch_small, csrc/mmn_epoch_small.inc): small MLPEncoder + ClassDecoder
        models on device-resident batches of at most `mmn_epoch_small_rows` rows each - the reference's Titanic pipeline,
        pipelines/titanic/titanic_mlp_pipeline.py:63-85, is the case it is built for.  Applies under the conditions of a
        replayed plan (stable batch objects, multimodn_amd.optim.Adam over exactly this model, device NaN policy, default
        encoder sequence, no data parallel, no logging, no dropout provider); `model.epoch_kernel = False` or
        MMN_EPOCH_KERNEL=0 keeps the step-by-step path.  Returns (engine, steps) or None.
        Like a replayed plan it records ADDRESSES: a batch is recognised by its objects (tuple, tensors), not re-read."""
        import itertools
        import operator
        if not getattr(self, "epoch_kernel", True) or os.environ.get("MMN_EPOCH_KERNEL", "1") == "0":
            return None
        if getattr(self, "nan_policy", "auto") not in ("auto", "device") or not hasattr(optimizer, "fused_descriptor"):
            return None
        eng0 = self._engine
        if eng0 is not None and eng0.__dict__.get("_eps_rows", (None, 1))[1] == 0 and eng0._eps_rows[0] == eng0._plan.value:
            return None                                      # (this model is outside the kernel's shapes: said once per plan)
        seq = train_loader if isinstance(train_loader, (list, tuple)) else list(train_loader)
        try:
            if not seq or len(seq) > 65536 or int(seq[0][1].shape[0]) > 64:
                return None
        except (TypeError, IndexError, AttributeError, KeyError):     # (a batch format the general loop will have its own words for)
            return None
        try:
            # (host batches never take this path: said HERE, before the engine is planned for `rows` and the first batch is
            #  staged a second time only to find that out - ADVICE r5)
            d0, y0 = seq[0][0], seq[0][1]
            if not (isinstance(y0, Tensor) and y0.device.type == self.device.type and y0.dtype == torch.int64 and y0.is_contiguous()
                    and all(isinstance(t, Tensor) and t.device.type == self.device.type and t.dtype == torch.float32
                            and t.is_contiguous() for t in d0)):
                return None
        except (TypeError, IndexError, KeyError):
            return None
        cache = self.__dict__.setdefault("_small_epochs", {})
        key = (len(seq), id(seq[0]), id(seq[-1]))
        ep = cache.get(key)
        if ep is not None:
            datas = list(map(operator.itemgetter(0), seq))
            if not (ep["opt"] is optimizer and ep["eng"] is self._engine and all(map(operator.is_, seq, ep["batches"]))
                    and all(map(operator.is_, map(operator.itemgetter(1), seq), ep["ys"])) and list(map(len, datas)) == ep["lens"]
                    and all(map(operator.is_, itertools.chain.from_iterable(datas), ep["xs"]))
                    and (max(map(len, seq)) == 2 or all(len(b_) == 2 or b_[2] is None for b_ in seq))
                    # (the descriptors hold ADDRESSES: `x.set_(...)` / `x.data = ...` on a cached batch keeps the object and moves
                    #  its storage - a few microseconds once per call)
                    and [t.data_ptr() for t in ep["xs"]] + [t.data_ptr() for t in ep["ys"]] == ep["ptrs"]):
                cache.pop(key, None)
                ep = None
        rows = ep["rows"] if ep is not None else max(int(b_[1].shape[0]) for b_ in seq)
        if rows > 64:
            return None
        eng = self._get_engine(rows)
        if not hasattr(eng, "epoch_small_rows") or eng.epoch_small_rows() < rows:
            return None
        if ep is not None and (eng._plan is None or eng._plan.value != ep["plan"]):
            cache.pop(key, None)
            ep = None
        eng.begin_sequence()
        if self._nan_mode(eng, optimizer, True) != "device":
            return None
        fd = self._fusion_setup(eng, optimizer, "device")
        if fd is None or fd.seg_skip:
            return None
        if ep is None:
            structs, ys, xs_all, lens = [], [], [], []
            for batch in seq:
                data, target, encoder_sequence = (list(batch) + [None])[:3]
                if encoder_sequence is not None or not isinstance(batch, tuple):
                    return None
                st = self._make_step(data, target, None, "device", True)
                if st.on_host or st.y is not target or len(st.xs) != len(data) or not all(a is c for a, c in zip(st.xs, data)) \
                        or st.bg != int(target.shape[0]) or [tuple(pe) for pe in st.pairs] != [(k, k) for k in range(len(self.encoders))]:
                    return None
                structs.append(eng.make_batch_keyed(st.xs, st.y, st.pairs, st.bg, False)[0])
                ys.append(target); xs_all.extend(data); lens.append(len(data))
            host = (hip.Batch * len(structs))(*structs)
            dev = torch.from_numpy(np.frombuffer(host, dtype=np.uint8).copy()).to(self.device)
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            ep = cache[key] = {"batches": list(seq), "ys": ys, "xs": xs_all, "lens": lens, "host": host, "dev": dev, "rows": rows,
                               "opt": optimizer, "eng": eng, "plan": eng._plan.value,
                               "ptrs": [t.data_ptr() for t in xs_all] + [t.data_ptr() for t in ys]}
        eng.epoch_reset()
        rc = eng.lib.mmn_train_epoch_small(eng._plan, ep["host"], ep["dev"].data_ptr(), len(seq), float(self.err_penalty),
                                           float(self.state_change_penalty), C.byref(fd), eng._stream())
        if rc == hip.ERR_UNSUPPORTED:
            cache.pop(key, None)
            return None
        hip.check(rc, "mmn_train_epoch_small")
        eng._versions_seen = None                            # (the chain kernels' weight copies are stale: the library says so too)
        optimizer.fused_step_seen(len(seq))
        self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + len(seq)
        return eng, len(seq)

    def _replay_epoch_plan(self, train_loader, optimizer):
        """The whole batch loop of a call whose batch OBJECTS, optimizer and engine are those of an earlier call that ran
        entirely as replayed hipGraph groups: no per-batch ingest, just the groups' replays (a per-call fixed cost of ~25
        instead of ~70 us: the driver's 20-step measurement is one such call).  Returns (engine, steps) or None when the
        plan does not apply.  Everything a replay must see fresh lives in device memory or is checked here: parameters
        moved / re-plan (engine.ensure), hyper-parameters and the dropout seed (run_group's graph key: a miss runs that
        group's steps eagerly), the kernels' weight copies (begin_sequence)."""
        import operator
        plans = self.__dict__.get("_epoch_plans")
        if not plans:
            return None
        seq = train_loader if isinstance(train_loader, (list, tuple)) else list(train_loader)
        if not seq:
            return None
        ep = plans.get((len(seq), id(seq[0]), id(seq[-1])))
        if ep is None or ep["opt"] is not optimizer or ep["eng"] is not self._engine or ep["group"] != (int(getattr(self, "REPLAY_GROUP", 8)), int(getattr(self, "REPLAY_GROUP_NEXT", 0))) \
                or len(ep["batches"]) != len(seq) or not all(map(operator.is_, seq, ep["batches"])):
            return None
        # the graphs hold the ADDRESSES recorded last time: every batch must still be made of the very tensor objects its
        # recorded step names (a batch whose inner list was edited: `batch[0][k] = new_x`).  The tensors' addresses are NOT
        # re-read (five data_ptr() calls per step are ~0.6 us per step of a call's latency, all of it in front of the
        # first launch): set_ / resize_-ing a tensor of a replayed plan in place needs `model._epoch_plans.clear()`.
        flat = ep.get("flat")
        if flat is None:                                     # per batch, in order: the targets, the slots' counts, every slot tensor
            steps_all = [st for g in ep["groups"] for st in g[0]]
            flat = ep["flat"] = ([st[1] for st in steps_all], [len(st[0]) for st in steps_all],
                                 [x for st in steps_all for x in st[0]])
        if len(flat[0]) != len(seq):
            return None
        # (the comparisons run inside map / all, not in a Python loop per batch: ~3 instead of ~10 us for 20 batches, all of
        #  it in front of the call's first launch)
        import itertools
        datas = list(map(operator.itemgetter(0), seq))
        if not (all(map(operator.is_, map(operator.itemgetter(1), seq), flat[0])) and list(map(len, datas)) == flat[1]
                and all(map(operator.is_, itertools.chain.from_iterable(datas), flat[2]))
                and (max(map(len, seq)) == 2 or all(len(b_) == 2 or b_[2] is None for b_ in seq))):
            plans.pop((len(seq), id(seq[0]), id(seq[-1])), None)
            return None
        eng = self._get_engine(ep["rows"])                   # (compares every parameter's address with the plan's)
        if eng._plan is None or eng._plan.value != ep["plan"]:
            return None
        if getattr(self, "nan_policy", "auto") not in ("auto", "device") or not hasattr(optimizer, "fused_descriptor"):
            return None
        # the rebuild of the kernels' weight copies (one small launch) goes out NOW: the GPU does it under the rest of this
        # function's checks instead of in front of the first graph (whatever path the call then takes needs it anyway)
        eng.begin_sequence(sig_checked=True)
        eng.refresh_weights()
        fd = optimizer.fused_descriptor(eng)
        if fd is None or not eng.adam_fusable(optimizer, fd):
            return None
        mode = "device"
        if eng.params[0].grad is not eng.grad_views[0] or eng.params[-1].grad is not eng.grad_views[-1]:
            eng.assign_grads(None)
        alpha, beta = float(self.err_penalty), float(self.state_change_penalty)
        draw = bool(eng.dropout_encoders)
        hp = eng.group_hp_key(alpha, beta, optimizer, fd, eng._dropout_seed() if draw else 0)
        if draw and eng._dropout_seed() != eng._drop_seed:    # a new torch seed restarts the draw index: the slow path does that
            return None
        total = 0
        need_reset = True                                    # the epoch accumulators: zeroed by the first group's graph
        for gi, (steps, nxt, ent) in enumerate(ep["groups"]):
            ok = eng.replay_known(ent, steps, nxt, hp, optimizer, draw, reset_first=need_reset)
            if not ok:
                ok = eng.run_group(steps, nxt, alpha, beta, optimizer, draw, fd, reset_first=need_reset)
                if ok:                                       # (the entry it replayed from: the shortcut next time)
                    ep["groups"][gi] = (steps, nxt, getattr(eng, "_last_group_entry", None))
            if need_reset and not ok:
                eng.epoch_reset()
            need_reset = False
            if ok:
                optimizer.fused_step_seen(len(steps))
            else:                                            # a key miss (LR schedule, new dropout seed, ...): this group eagerly
                for i, (xs, y, pairs, bg, b, key) in enumerate(steps):
                    st = MultiModN._Step(xs, y, pairs, None, False, bg)
                    st.b, st.key = b, key
                    nx = steps[i + 1] if i + 1 < len(steps) else nxt
                    nst = None
                    if nx is not None:
                        nst = MultiModN._Step(nx[0], nx[1], nx[2], None, False, nx[3])
                        nst.b, nst.key = nx[4], nx[5]
                    executed = self._launch_step(eng, st, nst, True, optimizer, mode, fd)
                    if st.stepped:
                        optimizer.fused_step_seen()
                    else:
                        eng.assign_grads(executed)
                        optimizer.step()
            total += len(steps)
            self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + len(steps)
        eng.note_parameters_current()
        return eng, total

    def _train_steps_per_sample(self, train_loader, optimizer):
        """The batch loop in per-sample mode.  With device-resident batches the regrouping of batch t+1 (four launches
        that depend on the data alone) runs on a side stream while step t runs on the main one: k_fb8 fills every CU's
        LDS with one workgroup but only a quarter of its wave slots, the regrouping kernels use no LDS to speak of.
        Two persistent buffer sets take the regrouped batches in turn; events order the two streams."""
        import itertools
        eng = None
        it = iter(train_loader)
        main = torch.cuda.current_stream() if self.device.type == "cuda" else None
        side = None
        done = [None, None]                                  # per buffer set: event behind the step that last read it
        state = {"i": 0, "fd": None, "fused": False, "plan": None}
        cache = self.__dict__.setdefault("_batch_cache", {})
        stable = _stable_batches(train_loader)

        def engine_for(target):
            """The engine, sized for this batch's regrouped rows; (re)binds the fused optimizer when the plan is new."""
            nonlocal eng
            if eng is None:
                eng = self._get_engine(int(target.shape[0]))
                eng.epoch_reset()
                eng.begin_sequence()
            rows = int(eng.lib.mmn_regroup_rows(int(target.shape[0]), eng.E)) if hasattr(eng, "lib") else 0
            replanned = eng.ensure(max(rows, 1))
            if state["plan"] is None or replanned:                          # first batch, or the plan had to grow
                state["fd"] = self._fusion_setup(eng, optimizer, "device") if self._dp_group is None else None
                state["fused"] = state["fd"] is not None
                state["plan"] = True

        def prepare(batch):
            nonlocal side
            data, target, encoder_sequence = (list(batch) + [None])[:3]
            engine_for(target)
            slot = state["i"] & 1
            state["i"] += 1
            if main is not None and hasattr(eng, "per_sample_batch_async") and isinstance(target, Tensor) and target.is_cuda:
                if side is None:
                    side = self.__dict__.get("_ps_stream")
                    if side is None:
                        side = self.__dict__["_ps_stream"] = torch.cuda.Stream(device=self.device)
                # EVERY batch: the loader may have produced this batch on the main stream inside next(it) (a shuffling
                # DeviceResidentLoader's index_select, a user's .to(device, non_blocking=True)); the regrouping kernels read
                # it on the side stream.  Step t is not enqueued yet at this point, so the wait costs no overlap.
                side.wait_stream(main)
                if done[slot] is not None:
                    side.wait_event(done[slot])             # the step that read this buffer set two batches ago
                y = target if target.dim() == 2 else target.view(-1, 1)
                ent = cache.get(id(batch)) if stable else None
                tmpl = ent[1] if (ent is not None and ent[0] is batch) else None
                r = eng.per_sample_batch_async(list(data), y, encoder_sequence if isinstance(encoder_sequence, Tensor) else None
                                               if encoder_sequence is None else torch.as_tensor(np.asarray(encoder_sequence)).to(self.device),
                                               slot, side, tmpl)
                if r is not None:
                    b, keep, ev, tmpl = r
                    if stable and isinstance(batch, tuple):
                        if len(cache) >= _BATCH_CACHE_MAX:
                            cache.clear()
                        cache[id(batch)] = (batch, tmpl)
                    b.batch_global = self._global_rows(int(y.shape[0]))
                    return (b, keep, keep[5], keep[6]), ev, slot
            return self._regroup_per_sample(eng, data, target, encoder_sequence), None, slot

        def run_eager(it):
            try:
                nxt = prepare(next(it))
            except StopIteration:
                return
            while nxt is not None:
                cur, ev, slot = nxt
                try:
                    nxt = prepare(next(it))                 # regrouping of batch t+1: enqueued before step t's kernels
                except StopIteration:
                    nxt = None
                if ev is not None:
                    main.wait_event(ev)
                if state["fused"]:
                    self._run_step_per_sample(eng, None, None, None, optimizer, regrouped=cur, desc=state["fd"])
                    optimizer.step()                         # (a no-op after a fused step; the real one if the library refused)
                else:
                    optimizer.zero_grad()
                    executed, keep = self._run_step_per_sample(eng, None, None, None, optimizer, regrouped=cur)
                    eng.assign_grads(executed)
                    optimizer.step()
                if ev is not None:
                    if done[slot] is None:
                        done[slot] = torch.cuda.Event()
                    done[slot].record(main)
                self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + 1

        def run_group(chunk) -> bool:
            """REPLAY_GROUP device-resident batches that come back every epoch as ONE hipGraph (engine.run_group_per_sample:
            regrouping launches, dropout draw and step of every batch); False: nothing launched, run them eagerly."""
            items = []
            for batch in chunk:
                data, target, seq = (list(batch) + [None])[:3]
                if not (isinstance(target, Tensor) and target.is_cuda and all(isinstance(x, Tensor) and x.is_cuda for x in data)):
                    return False
                if seq is not None and not (isinstance(seq, Tensor) and seq.is_cuda):
                    return False
                y = target if target.dim() == 2 else target.view(-1, 1)
                items.append((list(data), y, seq, self._global_rows(int(y.shape[0]))))
            for _, y, _, _ in items:
                engine_for(y)
            if not state["fused"] or not hasattr(eng, "run_group_per_sample"):
                return False
            if not eng.run_group_per_sample(items, float(self.err_penalty), float(self.state_change_penalty), optimizer,
                                            state["fd"], bool(eng.dropout_encoders)):
                return False
            n = len(items)
            self.__dict__["train_steps_launched"] = self.__dict__.get("train_steps_launched", 0) + n
            optimizer.fused_step_seen(n)
            return True

        G = int(self.REPLAY_GROUP)
        if (G > 1 and stable and main is not None and self._dp_group is None and getattr(self, "replay_steps", True)
                and self.dropout_mask_provider is None and hasattr(optimizer, "fused_descriptor")):
            # groups of G steps at fixed positions of the epoch: a group's graph is captured when its batches are seen
            # for the second time (the eager run of the first epoch is the warm-up) and replayed from then on
            while True:
                chunk = list(itertools.islice(it, G))
                if not chunk:
                    break
                if len(chunk) == G and run_group(chunk):
                    continue
                run_eager(iter(chunk))
            return eng
        run_eager(it)
        return eng

    def train_epoch(
            self,
            train_loader: DataLoader,
            optimizer: Optimizer,
            criterion: Union[nn.Module, Callable],
            history: Optional[MultiModNHistory] = None,
            log_interval: Optional[int] = None,
            logger: Optional[Callable] = None,
            last_epoch: Optional[bool] = False,
    ) -> None:
        """One pass over train_loader (multimodn.py:89-252)."""
        check_criterion(criterion)
        if log_interval and not logger:
            logger = print
        self.train()
        n_batches = len(train_loader)
        if self.per_sample:
            eng = self._train_steps_per_sample(train_loader, optimizer)
        else:
            eng, _ = self._train_steps(train_loader, optimizer, log_interval, logger)
        if eng is None:
            return None
        if history is not None:
            lists = [history.state_change_loss] + [getattr(history, m)["train"] for m in
                                                   ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy")]
            keys = ("state_change", "loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy")
            pending = None
            if all(isinstance(lst, HistoryList) for lst in lists) and hasattr(eng, "epoch_read_async"):
                # no synchronisation at the end of an epoch: the sums travel to pinned memory behind the last launch and
                # the six arrays are formed when somebody reads the History (the next epoch can be submitted meanwhile)
                rd = eng.epoch_read_async()
                if rd is not None:
                    wait, fetch = rd
                    pending = PendingEpoch(wait, lambda: self._epoch_arrays(eng, n_batches, fetch()))
            if pending is not None:
                for lst, key in zip(lists, keys):
                    lst.append_pending(pending, key)
            else:
                arrays = self._epoch_arrays(eng, n_batches)
                for lst, key in zip(lists, keys):
                    lst.append(arrays[key])
        if last_epoch:
            return self.test(train_loader, criterion, history=None)
        return None

    def _epoch_arrays(self, eng, n_batches: int, ep=None):
        """multimodn.py:222-242 from the device-side epoch accumulators."""
        if ep is None:
            ep = eng.epoch_read()
        n_samples = np.ones((len(self.encoders) + 1, 1)) + ep["rows"].reshape(-1, 1)   # starts at ONE (:105)
        loss = ep["err_sum"] / n_batches
        sc = ep["sc_sum"] / n_batches
        acc = ep["n_correct"] / n_samples
        tp, tn = ep["tp"].astype(np.float32), ep["tn"].astype(np.float32)
        fp, fn = ep["fp"].astype(np.float32), ep["fn"].astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            sd = tp + fn
            sens = np.where(sd == 0, np.float32(0), tp / sd).astype(np.float32)
            pd_ = tn + fp
            spec = np.where(pd_ == 0, np.float32(0), tn / pd_).astype(np.float32)
        return {"loss": loss, "state_change": sc, "accuracy": acc, "sensitivity": sens,
                "specificity": spec, "balanced_accuracy": (sens + spec) / 2}

    def test(
            self,
            test_loader: DataLoader,
            criterion: Union[nn.Module, Callable],
            history: Optional[MultiModNHistory] = None,
            tag: str = 'test',
            log_results: bool = False,
            logger: Optional[Callable] = None,
    ):
        """Forward-only epoch (multimodn.py:255-419) on the HIP forward kernels.  Appends the
        `tag` arrays to `history` and returns, per decoder, the reference's 15-value report
        (get_performance_metrics, multimodn.py:22-49) on the decoder fed with the state after the
        last encoder, computed natively (multimodn_amd/metrics.py) instead of via torchmetrics."""
        check_criterion(criterion)
        if log_results and not logger:
            logger = print
        self.eval()
        n_batches = len(test_loader)
        eng = None
        keep = None
        # Nothing in the batch loop waits for the GPU: targets, last-row outputs and (device NaN policy) the "row
        # exists" flags of every step stay on the device as small clones and are sorted out once after the loop
        # (a per-step .to("cpu") / flag readback made this loop ~3 ms per batch against ~45 us of kernels).
        outputs_epoch: List[Tensor] = []
        targets_epoch: List[Tensor] = []
        last_ran: List = []                                  # per step: bool (host policy) or a device flag tensor
        fast = self._test_steps_collected(test_loader) if (not self.per_sample and getattr(self, "collect_in_step", True)) else None
        if fast is not None:
            eng, outputs_epoch, targets_epoch, last_ran = fast
            test_loader = ()                                 # (the loop below has nothing left to do)
        for batch in test_loader:
            data, target, encoder_sequence = (list(batch) + [None])[:3]
            if eng is None:
                eng = self._get_engine(int(target.shape[0]))
                eng.epoch_reset()
                eng.begin_sequence()
            else:
                eng.ensure(int(target.shape[0]))
            last = len(self.encoders) - 1
            if self.per_sample:
                _, keep = self._run_step_per_sample(eng, data, target, encoder_sequence, train=False)
                where, codes = eng.per_sample_positions()
                has_last = torch.zeros_like(codes, dtype=torch.bool)
                for j in range(len(self.encoders)):
                    has_last |= ((codes >> (4 * j)) & 15) == last + 1
                rows_l = torch.nonzero(has_last).flatten()       # samples whose LAST encoder ran (multimodn.py:354-357)
                targets_epoch.append(keep[1].detach()[rows_l])
                bp = int(keep[2][1].shape[0])
                outputs_epoch.append(eng.decoder_outputs(last + 1, bp)[where[rows_l]].clone())
                continue
            executed, keep = self._run_step(eng, data, target, encoder_sequence, train=False)
            targets_epoch.append(keep[1].detach().clone())      # (a staged host batch's device copy is recycled)
            outputs_epoch.append(eng.decoder_outputs(last + 1, int(keep[1].shape[0])).clone())
            # multimodn.py:354-357: only batches whose LAST encoder ran contribute outputs; under the device NaN
            # policy that is known on the GPU only: keep this step's flag word for later
            last_ran.append(executed[last] if executed is not None else eng.executed_flags()[last + 1:last + 2].clone())
        if eng is None:
            return None
        arrays = self._epoch_arrays(eng, n_batches)
        del keep
        if log_results:
            logger(f"{tag.capitalize()} results\n"
                   f"\tAverage loss: {np.mean(arrays['loss']):.4f}\n"
                   f"\tAccuracy: {np.mean(arrays['accuracy']):.4f}\n"
                   f"\tBalanced accuracy: {np.mean(arrays['balanced_accuracy']):.4f}")
        if history is not None:
            for name in ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy"):
                getattr(history, name).setdefault(tag, []).append(arrays[name])
        # per-decoder report on the state after the last encoder (multimodn.py:410-419)
        results: List = [[]] * len(self.decoders)
        if last_ran and not self.per_sample:                   # one readback for all the steps' flags
            dev_flags = [f for f in last_ran if isinstance(f, Tensor)]
            if dev_flags:
                host_flags = iter(torch.cat(dev_flags).cpu().tolist())
                last_ran = [bool(next(host_flags)) if isinstance(f, Tensor) else f for f in last_ran]
            outputs_epoch = [o for o, ran in zip(outputs_epoch, last_ran) if ran]
            # (the reference sets these outputs against the targets of ALL batches and its report raises on the two
            # lengths, multimodn.py:410-418; here the report covers the batches whose last encoder ran - the case of
            # pipelines/titanic/titanic_missingness_pipeline.py, whose last feature is the one most often missing)
            targets_epoch = [t for t, ran in zip(targets_epoch, last_ran) if ran]
        if outputs_epoch:
            out = torch.cat(outputs_epoch, dim=0)                      # [N, 2D], on the model's device
            tgt = torch.cat(targets_epoch, dim=0).to(out.device)
            for d in range(len(self.decoders)):
                o = out[:, 2 * d:2 * d + 2]
                o = torch.div(o, torch.sum(o, dim=1).reshape(-1, 1))   # class probabilities sum to 1 (:415)
                _, pred = torch.max(o, dim=1)
                # the report (sort + cumulative sums over N scores) runs where the scores are; like the reference's
                # torchmetrics values the results are handed back as CPU tensors
                res = get_performance_metrics(tgt[:, d], pred, o[:, 1])
                results[d] = tuple(v.cpu() if isinstance(v, Tensor) else v for v in res)
        return results

    def _test_steps_collected(self, test_loader):
        """test()'s batch loop for a loader whose batches already live on this model's device, under the device NaN
        policy: ONE library call per step (scan, forward kernels, statistics, and the step's last-row outputs / "row exists"
        flag copied into epoch-sized buffers by that call: mmn_eval_step_ex), batch structs reused across epochs for
        loaders that hand the same batch objects back, no clone of the targets (the loader's own tensors are read once,
        after the loop).  The general loop costs ~95 us of host time per step against ~40 us of kernels at batch 4096.
        Returns (engine, [outputs of every step], [targets of every step], [device flag per step]) or None when the loader
        does not qualify (host batches, explicit sequences, mixed batch sizes are left to the general loop)."""
        if self.device.type != "cuda" or self._dp_group is not None or not isinstance(test_loader, (list, tuple)) \
                and not getattr(test_loader, "stable_batches", False):
            return None
        if getattr(self, "nan_policy", "auto") not in ("auto", "device"):
            return None
        batches = test_loader if isinstance(test_loader, (list, tuple)) else list(test_loader)
        if not batches:
            return None
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def here(t):                                        # (a device without an index means the current one)
            return isinstance(t, Tensor) and t.is_cuda and t.device.index == idx

        for batch in batches:
            if not isinstance(batch, (list, tuple)) or len(batch) < 2 or (len(batch) > 2 and batch[2] is not None):
                return None
            data, target = batch[0], batch[1]
            if not (here(target) and target.dtype == torch.int64 and target.dim() == 2 and target.is_contiguous()):
                return None
            if len(data) != len(self.encoders) or not all(here(x) and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
                                                          for x in data):
                return None
        rows = [int(b[1].shape[0]) for b in batches]
        eng = self._get_engine(max(rows))
        if not hasattr(eng, "eval_step_collect") or self._nan_mode(eng, None, False) != "device":
            return None
        eng.epoch_reset()
        eng.begin_sequence()
        n, D2, last = len(batches), 2 * len(self.decoders), len(self.encoders) - 1
        bufs = eng.__dict__.setdefault("_eval_bufs", {})
        key = (n, max(rows), D2)
        if key not in bufs:
            bufs.clear()
            bufs[key] = (torch.empty((n, max(rows), D2), dtype=torch.float32, device=self.device),
                         torch.zeros(n, dtype=torch.int32, device=self.device))
        out_buf, flag_buf = bufs[key]
        pairs = self.get_encoder_iterable(None, self.shuffle_mode, train=False)
        cache = self.__dict__.setdefault("_eval_batch_cache", {})   # (its own: train_epoch's entries carry another signature)
        sig = ("eval", self.shuffle_mode)
        for i, batch in enumerate(batches):
            data, target = batch[0], batch[1]
            ent = cache.get(id(batch)) if isinstance(batch, tuple) else None
            tmpl = None
            if ent is not None and ent[0] is batch and ent[1] is target and ent[3] == sig and len(ent[2]) == len(data) \
                    and all(a is c for a, c in zip(ent[2], data)):
                tmpl = ent[6]
            b, _, tmpl = eng.make_batch_keyed(list(data), target, pairs, rows[i], True, tmpl)
            if isinstance(batch, tuple) and not self.shuffle_mode:
                if len(cache) >= min(_BATCH_CACHE_MAX, 2 * n + 8):   # (never more than two loaders' worth of batches kept alive)
                    cache.clear()
                cache[id(batch)] = [batch, target, list(data), sig, pairs, rows[i], tmpl]
            eng.eval_step_collect(b, last + 1, out_buf[i], flag_buf[i:i + 1], accumulate=True)
        outputs = [out_buf[i, :rows[i]] for i in range(n)]
        targets = [b[1] for b in batches]
        flags = [flag_buf[i:i + 1] for i in range(n)]
        return eng, outputs, targets, flags

    def predict(self, x: List[Tensor], encoder_sequence: Optional[np.ndarray] = None) -> np.ndarray:
        """Predicted class of every decoder on every state: ndarray [(E+1), D, N] (multimodn.py:422-458).
        As in the reference there is NO NaN skip on this path, and rows of encoders that are not in
        the sequence stay 0."""
        self.eval()
        n_samples = int(x[0].shape[0])
        full = np.zeros((len(self.encoders) + 1, len(self.decoders), n_samples))
        if n_samples == 0:
            return full
        if self.per_sample:                                    # every sample its own order / missing modalities (skipped)
            eng = self._get_engine(n_samples)
            eng.begin_sequence()
            dummy = torch.zeros((n_samples, len(self.decoders)), dtype=torch.int64)
            _, keep = self._run_step_per_sample(eng, x, dummy, encoder_sequence, train=False)
            where, codes = eng.per_sample_positions()
            bp = int(keep[2][1].shape[0])
            for row in range(len(self.encoders) + 1):
                o = eng.decoder_outputs(row, bp)[where]
                pred = (o[:, 1::2] > o[:, 0::2]).to(torch.float64)
                if row > 0:
                    has = torch.zeros_like(codes, dtype=torch.bool)
                    for j in range(len(self.encoders)):
                        has |= ((codes >> (4 * j)) & 15) == row
                    pred = pred * has.unsqueeze(1)
                full[row] = pred.t().cpu().numpy()
            return full
        seq = None if encoder_sequence is None else np.asarray(encoder_sequence)
        pairs = self.get_encoder_iterable(seq, self.shuffle_mode, train=False)
        eng = self._get_engine(n_samples)
        eng.begin_sequence()
        xs = [t.to(self.device, dtype=torch.float32).contiguous() for t in x]
        y = torch.zeros((n_samples, len(self.decoders)), dtype=torch.int64, device=self.device)
        b = eng.make_batch(xs, y, pairs, batch_global=n_samples, device_nan_flags=False)
        eng.eval_step(b, accumulate=False)
        D = len(self.decoders)
        for row in [0] + [e + 1 for _, e in pairs]:
            o = eng.decoder_outputs(row, n_samples)
            pred = (o[:, 1::2] > o[:, 0::2])                    # torch.max: first index wins ties
            full[row] = pred.to(torch.float64).t().cpu().numpy().reshape(D, n_samples)
        del xs, y
        return full

    def display_arch(self, input: np.ndarray):
        """Prints every encoder and decoder layer by layer - output shape and parameter count for ONE sample, `input[i]` the
        features of encoder i (multimodn.py:494-507, which hands each module to torchsummary; that package is not a
        dependency here: the table comes from forward hooks on a CPU copy of the module, nothing touches the GPU)."""
        import copy
        S = int(self.init_state.state_size)

        def table(module, args):
            module = copy.deepcopy(module).to("cpu").eval()
            rows, hooks = [], []
            for name, sub in module.named_modules():
                if name and not list(sub.children()):
                    hooks.append(sub.register_forward_hook(
                        lambda m, i, o, name=name: rows.append((f"{type(m).__name__} ({name})", tuple(o.shape),
                                                                sum(p.numel() for p in m.parameters(recurse=False))))))
            with torch.no_grad():
                module(*args)
            for h in hooks:
                h.remove()
            print(f"{'Layer (name)':<34}{'Output shape':<22}{'Param #':>10}")
            for nm, shape, n in rows:
                print(f"{nm:<34}{str(list(shape)):<22}{n:>10,}")
            total = sum(p.numel() for p in module.parameters())
            print(f"Total params: {total:,}  (trainable: {sum(p.numel() for p in module.parameters() if p.requires_grad):,})")

        for i, enc in enumerate(self.encoders):
            print('Encoder {}:'.format(i))
            x = torch.as_tensor(np.asarray(input[i], dtype=np.float32)).reshape(1, -1)
            table(enc, (torch.zeros(1, S), x))
            print()
        for i, dec in enumerate(self.decoders):
            print('Decoder {}:'.format(i))
            table(dec, (torch.zeros(1, S),))
            print()

    def get_states(self, data_loader: DataLoader) -> List[Tensor]:
        """The state after the last executed encoder, one [S] tensor per sample (multimodn.py:460-492)."""
        self.eval()
        batch_states: List[Tensor] = []
        for batch in data_loader:
            data, _, encoder_sequence = (list(batch) + [None])[:3]
            n = int(data[0].shape[0])
            eng = self._get_engine(n)
            if not batch_states:
                eng.begin_sequence()
            dummy = torch.zeros((n, len(self.decoders)), dtype=torch.int64)
            if self.per_sample:
                _, keep = self._run_step_per_sample(eng, data, dummy, encoder_sequence, train=False)
                where, codes = eng.per_sample_positions()
                bp = int(keep[2][1].shape[0])
                state = self.init_state(n).detach().to(self.device)
                last_e = torch.full_like(codes, -1)
                for j in range(len(self.encoders)):                # the last non-zero nibble = last executed encoder
                    nib = (codes >> (4 * j)) & 15
                    last_e = torch.where(nib > 0, nib - 1, last_e)
                for e in range(len(self.encoders)):
                    sel = torch.nonzero(last_e == e).flatten()
                    if sel.numel():
                        state[sel] = eng.state_rows(e, bp)[where[sel]]
                batch_states.append(state)
                continue
            pairs = self.get_encoder_iterable(encoder_sequence, self.shuffle_mode, train=False)
            xs, y, exec_pairs, executed, _ = self._ingest(data, dummy, pairs, self._nan_mode(eng, None, False))
            b = eng.make_batch(xs, y, exec_pairs, batch_global=n, device_nan_flags=executed is None)
            eng.eval_step(b, accumulate=False)
            if executed is None:
                # device NaN policy: which encoders ran is known on the GPU only; pick the last executed one's state
                # with device-side selects instead of reading the flags back every batch
                flags = eng.executed_flags()
                state = self.init_state(n).detach().to(self.device)
                for _, e in pairs:
                    state = torch.where(flags[e + 1] != 0, eng.state_rows(e, n), state)
                batch_states.append(state.clone())
                del xs, y
                continue
            last = None
            for _, e in pairs:
                if executed[e]:
                    last = e
            if last is None:
                batch_states.append(self.init_state(n).detach().to(self.device))
            else:
                batch_states.append(eng.state_rows(last, n).clone())
            del xs, y
        return list(torch.cat(batch_states, dim=0)) if batch_states else []
