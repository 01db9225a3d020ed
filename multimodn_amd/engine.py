"""HipChainEngine: owns the flat parameter / gradient buffers, the device workspace and the
launch plan of libmmn_hip.so for one MultiModN model, and runs the training step
(reference: multimodn/multimodn.py:137-203) as HIP launches on torch's current stream.

PyTorch is plumbing here: device memory (tensors), streams, torch.distributed.  All arithmetic
of the step happens in the HIP kernels; there is no torch-op or CPU fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from .decoders import ClassDecoder, MLPDecoder
from .encoders import MIMIC_MLPEncoder, MLPEncoder, _identity
from .state import TrainableInitState


class UnsupportedModelError(NotImplementedError):
    pass


def activation_code(fn) -> int:
    if fn in (F.relu, torch.relu):
        return hip.ACT_RELU
    if fn in (torch.sigmoid, F.sigmoid):
        return hip.ACT_SIGMOID
    if fn is _identity:
        return hip.ACT_IDENTITY
    raise UnsupportedModelError(
        f"encoder activation {fn!r} is not on the HIP path (supported: relu, sigmoid, identity via "
        f"LinearEncoder)")


def check_supported(model) -> None:
    """The HIP path covers the reference's tabular hot path: TrainableInitState +
    MLPEncoder family + two-class sigmoid ClassDecoder, and the MIMIC pipelines' modules (8f #1):
    MIMIC_MLPEncoder + two-class sigmoid MLPDecoder."""
    if not isinstance(model.init_state, TrainableInitState):
        raise UnsupportedModelError("init_state must be TrainableInitState")
    if len(model.encoders) > hip.MAX_ENCODERS or len(model.decoders) > hip.MAX_DECODERS:
        raise UnsupportedModelError("too many encoders / decoders for libmmn_hip")
    for enc in model.encoders:
        if isinstance(enc, MIMIC_MLPEncoder):
            if len(enc.linears) > hip.MAX_LAYERS:
                raise UnsupportedModelError("too many encoder layers")
            activation_code(enc.activation)
            continue
        if not isinstance(enc, MLPEncoder):
            raise UnsupportedModelError(f"encoder {type(enc).__name__} is not on the HIP path "
                                        f"(MLPEncoder family and MIMIC_MLPEncoder only)")
        if len(enc.layers) > hip.MAX_LAYERS:
            raise UnsupportedModelError("too many encoder layers")
        if len(enc.layers) > 1:
            activation_code(enc.activation)
    for dec in model.decoders:
        if isinstance(dec, MLPDecoder):
            if dec.n_classes != 2 or dec.output_activation not in (torch.sigmoid, F.sigmoid):
                raise UnsupportedModelError("MLPDecoder must have n_classes=2 and a sigmoid output")
            if len(dec.layers) - 1 > hip.MAX_DEC_HIDDEN:
                raise UnsupportedModelError(f"MLPDecoder: at most {hip.MAX_DEC_HIDDEN} hidden layers")
            if len(dec.layers) > 1:
                activation_code(dec.hidden_activation)
            continue
        if not isinstance(dec, ClassDecoder) or dec.n_classes != 2 or dec.activation not in (torch.sigmoid, F.sigmoid):
            raise UnsupportedModelError("decoders must be ClassDecoder(n_classes=2, sigmoid) / LogisticDecoder / MLPDecoder")


def check_criterion(criterion) -> None:
    """The decoder-grid kernel implements nn.CrossEntropyLoss() exactly as every reference
    pipeline constructs it (titanic_mlp_pipeline.py:76): mean reduction, no weights/smoothing."""
    ok = (isinstance(criterion, torch.nn.CrossEntropyLoss) and criterion.reduction == "mean"
          and criterion.weight is None and getattr(criterion, "label_smoothing", 0.0) == 0.0)
    if not ok:
        raise UnsupportedModelError(
            "criterion must be torch.nn.CrossEntropyLoss() (mean reduction, no class weights, no label smoothing)")


@contextlib.contextmanager
def _capture(graph, stream):
    """torch.cuda.graph(graph, stream) with Python's cyclic garbage collector held back for the length of the capture: a
    collection that runs INSIDE a capture may finalise an older torch.cuda.CUDAGraph (another model's or an evicted group's,
    kept alive by a reference cycle until then) - destroying a hipGraph is not permitted while a stream of the thread
    captures, torch's destructor throws and the process terminates (seen under the GPU electric fence, round 6;
    torch.cuda.graph collects once BEFORE it begins, which does not cover garbage that becomes collectable during the
    capture).  Reference-counted frees are not affected; the collector runs again right behind the capture."""
    was = gc.isenabled()
    cm = torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local")
    cm.__enter__()
    gc.disable()                                                # (behind torch.cuda.graph.__enter__'s own collection)
    try:
        try:
            yield
        except BaseException:
            if not cm.__exit__(*sys.exc_info()):                # (ends the capture; the collector stays off until it has)
                raise
        else:
            cm.__exit__(None, None, None)
    finally:
        if was:
            gc.enable()


PER_SAMPLE_SCOPE = ("per-sample mode (per-sample missing modalities / encoder order, BASELINE configs[4]) runs models of at most 4 "
                    "encoders with per-sample encoder order (MLPEncoder family with n_features <= 64 and hidden widths <= 32: the fused "
                    "chain kernel's tiled form; every other MLPEncoder / MIMIC_MLPEncoder / MLPDecoder shape: the generic tier's tiled "
                    "forms) and models of 5 to 8 encoders in the default encoder order (no encoder_sequence: per-sample missing "
                    "modalities only); this model / batch is outside that set")


def _check_regroup(rc: int, what: str) -> None:
    if rc == hip.ERR_UNSUPPORTED:
        raise UnsupportedModelError(PER_SAMPLE_SCOPE)
    hip.check(rc, what)


class HipChainEngine:
    def __init__(self, model, max_batch: int):
        check_supported(model)
        self.lib = hip.load()
        self.model = model
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise hip.MmnError(
                f"MultiModN parameters live on {dev}; the training hot path only runs on an AMD GPU "
                f"(torch device 'cuda' on ROCm). There is no CPU fallback.")
        self.device = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.params: List[torch.nn.Parameter] = list(model.parameters())
        self.names: List[str] = [n for n, _ in model.named_parameters()]
        for p in self.params:
            if p.dtype != torch.float32:
                raise UnsupportedModelError("parameters must be float32")
        self._flatten_params()
        self.n_params = self.flat_params.numel()
        self._torch_regroup = False          # tests: force the torch-op regrouping of per-sample mode
        self._generic_tier = False           # plan the generic tier's kernels (per-sample mode of MLPEncoder shapes outside k_fb9's)
        self._build(max_batch)

    # ------------------------------------------------------------------ buffers and plan
    def _flatten_params(self) -> None:
        """Make every Parameter a view into one flat buffer (same Parameter objects, so an
        optimizer built earlier keeps working; values preserved)."""
        total = sum(p.numel() for p in self.params)
        flat = torch.empty(total, dtype=torch.float32, device=self.device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self.flat_params = flat

    def _build(self, max_batch: int) -> None:
        lib, model = self.lib, self.model
        m = hip.Model()
        m.state_size = model.init_state.state_size
        m.n_encoders = len(model.encoders)
        m.n_decoders = len(model.decoders)
        m.flags = hip.MODEL_GENERIC_TIER if self._generic_tier else 0
        n_stats_probe = None
        # reduce buffer = [flat grads | stats] so that data-parallel needs ONE all-reduce
        # (stats size needs the model dims only)
        m_probe = hip.Model()
        m_probe.state_size, m_probe.n_encoders, m_probe.n_decoders = m.state_size, m.n_encoders, m.n_decoders
        self.n_stats = int(lib.mmn_stats_floats(C.byref(m_probe)))
        self.reduce_buf = torch.zeros(self.n_params + self.n_stats, dtype=torch.float32, device=self.device)
        self.flat_grads = self.reduce_buf[:self.n_params]
        self.stats = self.reduce_buf[self.n_params:]
        self.grad_views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.grad_views.append(self.flat_grads[off:off + n].view(p.shape))
            off += n
        gv: Dict[int, torch.Tensor] = {id(p): g for p, g in zip(self.params, self.grad_views)}

        def ptrs(p):
            return p.data_ptr(), gv[id(p)].data_ptr()

        m.init_state, m.g_init_state = ptrs(model.init_state.state_value)
        self.enc_param_ids: List[List[int]] = []
        for e, enc in enumerate(model.encoders):
            me = m.enc[e]
            me.n_features = enc.n_features
            mimic = isinstance(enc, MIMIC_MLPEncoder)
            linears = enc.linears if mimic else list(enc.layers)
            me.kind = hip.ENC_MIMIC if mimic else hip.ENC_MLP
            me.n_layers = len(linears)
            me.activation = activation_code(enc.activation) if (mimic or len(linears) > 1) else hip.ACT_IDENTITY
            ids = []
            for l, lin in enumerate(linears):
                ml = me.layer[l]
                ml.w, ml.gw = ptrs(lin.weight)
                ml.b, ml.gb = ptrs(lin.bias)
                ml.out_dim, ml.in_dim = lin.out_features, lin.in_features
                ids += [id(lin.weight), id(lin.bias)]
            self.enc_param_ids.append(ids)
        for d, dec in enumerate(model.decoders):
            md = m.dec[d]
            if isinstance(dec, MLPDecoder):
                lins = list(dec.layers)
                md.n_hidden = len(lins) - 1
                md.hidden_activation = activation_code(dec.hidden_activation) if md.n_hidden else hip.ACT_IDENTITY
                for l, lin in enumerate(lins[:-1]):
                    mh = md.hidden[l]
                    mh.w, mh.gw = ptrs(lin.weight)
                    mh.b, mh.gb = ptrs(lin.bias)
                    mh.out_dim, mh.in_dim = lin.out_features, lin.in_features
                out = lins[-1]
            else:
                out = dec.fc
            md.w, md.gw = ptrs(out.weight)
            md.b, md.gb = ptrs(out.bias)
        self._m = m
        self.max_batch = int(max_batch)
        ws_bytes = int(lib.mmn_workspace_bytes(C.byref(m), self.max_batch))
        if ws_bytes == 0:
            raise UnsupportedModelError("model dimensions are outside libmmn_hip's limits "
                                        f"(state/hidden width <= {hip.MAX_DIM})")
        self.workspace = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        self._ws_ptr = (base + 255) // 256 * 256
        plan = C.c_void_p()
        hip.check(lib.mmn_plan_create(C.byref(m), self.max_batch, self._ws_ptr, ws_bytes,
                                      self.stats.data_ptr(), C.byref(plan)), "mmn_plan_create")
        self._plan = plan
        # plan-owned NaN flags: TWO sets at the end of the stats block.  Batches that carry device flags take them in
        # turn, so that a step can consume one set while its last launch pre-scans the next batch into the other.
        self._flag_sets = [lib.mmn_nan_flags_set(plan, 0), lib.mmn_nan_flags_set(plan, 1)]
        self._nan_flags_ptr = self._flag_sets[0]
        self.flag_tail = self.stats[self.n_stats - 2 * hip.MAX_ENCODERS:]     # both sets, as the floats the all-reduce sums
        self._flag_turn = 0
        self._prescanned: Optional[hip.Batch] = None       # the batch whose flags the last step's pre-scan left behind
        self.n_epoch = int(lib.mmn_epoch_doubles(C.byref(m)))
        self._sig = tuple(p.data_ptr() for p in self.params)
        self.E, self.D, self.S = m.n_encoders, m.n_decoders, m.state_size
        # (encoder id, mask width F + S, p) of the MIMIC encoders whose nn.Dropout is active in training
        self.dropout_encoders = [(e, enc.n_features + self.S, float(enc.dropout))
                                 for e, enc in enumerate(model.encoders) if isinstance(enc, MIMIC_MLPEncoder) and enc.dropout > 0]
        self._drop_p = (C.c_float * hip.MAX_ENCODERS)(*[float(getattr(enc, "dropout", 0.0)) if isinstance(enc, MIMIC_MLPEncoder)
                                                          else 0.0 for enc in model.encoders])
        self._drop_buf = None
        self._drop_seed = None
        self._predrawn = None                               # the batch whose multipliers the previous step's last launch drew
        self._step_graphs: Dict[tuple, list] = {}             # run_group: key -> [sightings, graph, keep-alive]
        self._graph_hits = 0

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self.lib.mmn_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def set_per_sample(self, on: bool) -> bool:
        """Per-sample mode (MultiModN.per_sample, BASELINE configs[4]) needs a plan whose chain kernels take regrouped tiles:
        the fused kernel's tiled form (MLPEncoder models with n_features <= 64 and hidden widths <= 32), the MIMIC modules'
        kernels - or, for every other MLPEncoder shape, the generic tier's sequential tiled form, which a plan only has when
        it is asked for (mmn_model.flags, MMN_MODEL_GENERIC_TIER): the reference runs ANY shape at batch size 1
        (multimodn/multimodn.py:509-531, pipelines/titanic/titanic_missingness_pipeline.py), so per-sample mode must too.
        Re-plans when the answer changes (the running epoch's sums are carried over).  Returns True if it re-planned."""
        want = False
        if on and not self.lib.mmn_per_sample_supported(self._plan):
            want = True
        elif on and self._generic_tier:
            want = True                                      # (stays: the default plan of this model did not take per-sample batches)
        if want == self._generic_tier:
            return False
        self._generic_tier = want
        self._replan(self.max_batch)
        if on and not self.lib.mmn_per_sample_supported(self._plan):
            self._generic_tier = False                      # (no tier takes it - more than 8 encoders, LDS: say so; ordinary batches keep their plan)
            self._replan(self.max_batch)
            raise UnsupportedModelError(PER_SAMPLE_SCOPE)
        return True

    def _replan(self, max_batch: int) -> None:
        # a new plan starts with empty epoch accumulators: carry the running epoch's sums over (a batch larger than
        # every earlier one may arrive in the middle of an epoch: variable batch samplers, per-sample mode)
        carried = np.zeros(self.n_epoch, np.float64)
        hip.check(self.lib.mmn_epoch_read(self._plan, carried.ctypes.data_as(C.POINTER(C.c_double)), self._stream()),
                  "mmn_epoch_read")
        self.lib.mmn_plan_destroy(self._plan)
        self._plan = None
        self._build(max_batch)
        hip.check(self.lib.mmn_epoch_write(self._plan, carried.ctypes.data_as(C.POINTER(C.c_double)), self._stream()),
                  "mmn_epoch_write")

    def ensure(self, batch: int) -> bool:
        """Re-plan if the batch outgrew the workspace or the parameters moved (model.to(), ...).  Returns True if it
        did: hip.Batch structs made before hold pointers into the old plan."""
        moved = tuple(p.data_ptr() for p in self.params) != self._sig
        if moved:
            self._flatten_params()
        if moved or batch > self.max_batch:
            self._replan(max(batch, self.max_batch))
            return True
        return False

    # ------------------------------------------------------------------ per-step
    def _stream(self) -> int:
        """torch's current stream on this engine's device as a raw handle.  (torch.cuda.current_stream() builds a Stream object
        through four Python layers - ~2 us a call, several calls in front of a batch-loop call's first launch; the raw
        getter is one C call.)"""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index)
        except AttributeError:                              # (a torch build without the raw getter)
            return torch.cuda.current_stream(self.device).cuda_stream

    def make_batch(self, xs: Sequence[torch.Tensor], y: torch.Tensor, pairs: Sequence[Tuple[int, int]],
                   batch_global: Optional[int] = None, device_nan_flags: bool = False) -> hip.Batch:
        """xs / y must be device tensors (float32 / int64).  The returned struct holds raw
        pointers: the caller keeps xs / y alive until the step's launches have run."""
        return self.make_batch_keyed(xs, y, pairs, batch_global, device_nan_flags)[0]

    def make_batch_keyed(self, xs, y, pairs, batch_global=None, device_nan_flags: bool = False, template=None):
        """make_batch plus what recurring batches are recognised by: returns (struct, key, template).  `key` names the
        buffers, the sequence and the flag set (run_group's cache key is made of these); `template` = (filled-in struct
        without flags, key without flags) can be handed back in for the same tensors next time: filling the ctypes
        arrays field by field costs more host time than a small step's launches."""
        if template is None:
            B = int(y.shape[0])
            bg = int(batch_global) if batch_global else B
            tmpl = hip.Batch()
            for k, x in enumerate(xs):
                if x.dtype != torch.float32 or x.device != self.device or x.dim() != 2 or x.stride(1) != 1:
                    raise ValueError(f"data slot {k}: expected a float32 [B, F] tensor on {self.device} with unit inner stride")
                tmpl.x[k] = x.data_ptr()
                tmpl.ldx[k] = x.stride(0)
            if y.dtype != torch.int64 or not y.is_contiguous() or y.device != self.device:
                raise ValueError("targets must be a contiguous int64 [B, D] tensor on the model's device")
            tmpl.y = y.data_ptr()
            tmpl.batch = B
            tmpl.batch_global = bg
            tmpl.n_seq = len(pairs)
            for t, (k, e) in enumerate(pairs):
                tmpl.seq_data[t] = k
                tmpl.seq_enc[t] = e
            base = (tuple(tmpl.x[k] for k in range(len(xs))), tuple(tmpl.ldx[k] for k in range(len(xs))), tmpl.y, B,
                    tuple(pairs), bg)
            template = (tmpl, base)
        b = hip.Batch.from_buffer_copy(template[0])
        if device_nan_flags:                                # the two flag sets alternate in the order batches are made
            b.nan_flags = self._flag_sets[self._flag_turn]
            self._flag_turn ^= 1
        return b, template[1] + (b.nan_flags,), template

    def draw_dropout_masks(self, b: hip.Batch, provider=None) -> List[torch.Tensor]:
        """Training-mode nn.Dropout of the MIMIC encoders (mlp_encoder.py:34,41): one [batch, F_e + S]
        multiplier tensor (0 or 1/(1-p)) per encoder that runs this step, drawn on the device by ONE
        HIP launch (k_dropout, seeded with torch.initial_seed(): torch.manual_seed governs it as it governs
        the reference's modules) and handed to the kernels through mmn_batch.drop_mask.  `provider(e, batch, width)` (tests:
        the masks the reference drew) replaces the draw.  Returns the tensors: keep them alive until
        the step's launches have run."""
        running = {b.seq_enc[t] for t in range(b.n_seq)}
        todo = [(e, w, p) for e, w, p in self.dropout_encoders if e in running]
        if not todo:
            return []
        B = int(b.batch)
        if provider is None:
            # one k_dropout launch (Philox4x32-10 keyed by torch's seed; include/mmn_hip.h, mmn_draw_dropout): the draw
            # index advances on the device, so a captured step draws fresh multipliers at every graph replay
            need = int(self.lib.mmn_dropout_floats(self._plan, self.max_batch))
            if self._drop_buf is None or self._drop_buf.numel() < need:
                self._drop_buf = torch.empty(need, dtype=torch.float32, device=self.device)
            seed = self._dropout_seed()
            predrawn, self._predrawn = self._predrawn, None
            if predrawn is b and seed == self._drop_seed:
                # the previous step's last launch drew them (mmn_step_opts.next_drop_p): no launch, same draw index
                hip.check(self.lib.mmn_dropout_adopt(self._plan, C.byref(b), self._drop_p, self._drop_buf.data_ptr(),
                                                     self._drop_buf.numel(), self._stream()), "mmn_dropout_adopt")
            else:
                if seed != self._drop_seed:                    # torch.manual_seed(...) restarts the sequence
                    self.reset_dropout()
                    self._drop_seed = seed
                hip.check(self.lib.mmn_draw_dropout(self._plan, C.byref(b), self._drop_p, seed, self._drop_buf.data_ptr(),
                                                    self._drop_buf.numel(), self._stream()), "mmn_draw_dropout")
            views, off = [], 0
            for e, enc in enumerate(self.model.encoders):      # the library's layout: MIMIC encoders in id order
                if not isinstance(enc, MIMIC_MLPEncoder):
                    continue
                w = enc.n_features + self.S
                n = (B * w + 3) // 4 * 4
                if enc.dropout > 0 and e in running:
                    views.append(self._drop_buf[off:off + B * w].view(B, w))
                    off += n
            return views
        keep = []
        for e, width, p in todo:
            mk = provider(e, B, width)
            if mk is None:
                continue
            mk = mk.to(self.device, torch.float32).contiguous()
            if tuple(mk.shape) != (B, width):
                raise ValueError(f"dropout mask of encoder {e}: expected {(B, width)}, got {tuple(mk.shape)}")
            b.drop_mask[e] = mk.data_ptr()
            keep.append(mk)
        return keep

    def _dropout_seed(self) -> int:
        return (int(torch.initial_seed()) ^ (int(getattr(self, "dropout_salt", 0)) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF

    def reset_dropout(self) -> None:
        """Restart the dropout generator's draw index (same seed -> same multipliers again)."""
        hip.check(self.lib.mmn_dropout_reset(self._plan, self._stream()), "mmn_dropout_reset")

    def _param_versions(self):
        return (hip.PARAM_WRITES[0],) + tuple(p._version for p in self.params)

    def note_parameters_current(self) -> None:
        """The chain kernels' weight copies are current NOW (a fused step has just scattered them / a repack has run) and
        nothing but this engine has written the parameters since: remember torch's per-parameter version counters, which
        every in-place operation on a Parameter - optimizer steps, load_state_dict's copy_, nn.init.* - advances (the
        library's own updates go through raw pointers and leave them alone)."""
        self._versions_seen = self._param_versions()

    def begin_sequence(self, sig_checked: bool = False) -> None:
        """Start of an epoch / an entry point: the flag sets are handed out from set 0 again (so that a replayed group of
        steps meets the sets it was captured with), no pre-scan is carried over.  The chain kernels' copies of the weights
        are REBUILT by the first step (one small launch per call): the reference re-reads its parameters on every call
        (multimodn.py:139-191 run the modules themselves), so anything may have written them in between - also writers
        torch's version counters never see (`p.data.mul_(...)`, `p.data.copy_(ema)`, raw pointers).
        `model.trust_param_versions = True` (opt-in) skips the rebuild while nothing has written the parameters since this
        engine last left the copies current, as far as torch's per-parameter `_version` counters, the library's own write
        counter and the storage addresses can tell; writes that bypass those then need `invalidate_weights()`."""
        self._flag_turn = 0
        self._prescanned = None
        self._predrawn = None
        seen = getattr(self, "_versions_seen", None)
        trust = bool(getattr(self.model, "trust_param_versions", False))
        if not trust or seen is None or seen != self._param_versions() or \
                (not sig_checked and tuple(p.data_ptr() for p in self.params) != self._sig):   # (ensure() has just compared them)
            self.lib.mmn_pack_invalidate(self._plan)
        self._versions_seen = None                          # whoever runs steps next says when the copies are current again

    def refresh_weights(self) -> None:
        """Rebuild the kernels' weight copies now if they are stale (one small launch on the current stream; what the first
        step of a sequence would do in front of its own launches)."""
        hip.check(self.lib.mmn_pack_refresh(self._plan, self._stream()), "mmn_pack_refresh")

    def invalidate_weights(self) -> None:
        """Something wrote the parameters behind torch's back: the next step rebuilds the kernels' weight copies."""
        self._versions_seen = None
        self.lib.mmn_pack_invalidate(self._plan)

    def local_step(self, b: hip.Batch, err_penalty: float, sc_penalty_x001: float, accumulate: bool = False,
                   optimizer=None, next_batch: Optional[hip.Batch] = None, desc=None, predraw_next: bool = False) -> bool:
        """[prepare: NaN scan when the batch carries device flags nobody pre-scanned; repack of the weights when the
        copies are stale] + fwd + bwd + wgrad + reduce: afterwards reduce_buf = [grads | stats] holds this rank's sums
        (already divided by batch_global).  accumulate=True also folds the loss combination / epoch accumulation into
        the last launch (single-GPU).

        optimizer: a multimodn_amd.optim.Adam over exactly this model's parameters; its step is
        then fused behind the gradient sum in the last launch (single GPU only: with data parallel
        the all-reduce has to come first).  Returns True if the optimizer step was applied by this
        call (the optimizer's next .step() is then a no-op), False if the caller still has to step.

        next_batch: the batch the NEXT call will run (made right after `b`, so that it holds the other flag set): its NaN
        scan rides in this step's last launch, and the next call finds its flags ready.
        desc: optimizer.fused_descriptor(self), when the caller already has it (it walks every parameter).
        predraw_next: the next call is a training step that draws its dropout multipliers on the device: they are drawn
        in this step's last launch (same generator, same draw index), and draw_dropout_masks(next_batch) adopts them."""
        if self._launch_step(b, err_penalty, sc_penalty_x001, accumulate, optimizer, next_batch, desc, predraw_next):
            optimizer.mark_fused_step()
            return True
        return False

    def _launch_step(self, b, err_penalty, sc_penalty_x001, accumulate, optimizer, next_batch, desc=None,
                     predraw_next: bool = False) -> bool:
        b.flags_ready = 1 if (b.nan_flags and self._prescanned is b) else 0
        o = hip.StepOpts()
        o.accumulate_epoch = 1 if accumulate else 0
        if next_batch is not None:
            o.next = C.pointer(next_batch)
        drew = None
        if (predraw_next and next_batch is not None and self._drop_buf is not None and self._drop_seed is not None
                and self._drop_seed == self._dropout_seed() and not next_batch.tile_seq):
            o.next_drop_p = self._drop_p
            o.next_drop_buf = self._drop_buf.data_ptr()
            o.next_drop_seed = self._drop_seed
            o.next_drop_floats = self._drop_buf.numel()
            drew = next_batch
        pre = next_batch if (next_batch is not None and next_batch.nan_flags and b.nan_flags
                             and next_batch.nan_flags != b.nan_flags and not next_batch.tile_seq) else None
        if optimizer is not None:
            d = desc if desc is not None else optimizer.fused_descriptor(self)
            if d is not None:
                o.adam = C.pointer(d)
                rc = self.lib.mmn_train_step_ex(self._plan, C.byref(b), err_penalty, sc_penalty_x001, C.byref(o), self._stream())
                if rc == 0:
                    self._prescanned = pre
                    self._predrawn = drew
                    return True
                if rc != hip.ERR_UNSUPPORTED:
                    hip.check(rc, "mmn_train_step_ex")
                optimizer.fusion_refused(self)
                o.adam = None
        hip.check(self.lib.mmn_train_step_ex(self._plan, C.byref(b), err_penalty, sc_penalty_x001, C.byref(o), self._stream()),
                  "mmn_train_step_ex")
        self._prescanned = pre
        self._predrawn = drew
        return False

    # ------------------------------------------------------------------ whole steps as replayable hipGraphs
    MAX_STEP_GRAPHS = 64

    def run_group(self, steps, nxt, err_penalty: float, sc_penalty_x001: float, optimizer, draw_dropout: bool,
                  desc=None, reset_first: bool = False, dp_tail=None) -> bool:
        """`steps`: consecutive training steps (device NaN policy, Adam fused) as (xs, y, pairs, batch_global, hip.Batch,
        key) tuples (make_batch_keyed) whose device buffers this engine may have seen before; `nxt`: the step after them (or None) - the last
        step of the group pre-scans its batch.  The group - per step k_dropout, [k_prepare scan], chain, k_wgrad,
        k_reduce + Adam - is captured into ONE hipGraph on its second sighting and replayed from then on: one host
        submission for len(steps) steps.  Loaders hand the same device buffers back every epoch (DeviceResidentLoader
        views, the staging ring of host batches), so this pays from the second epoch on; at the reference pipelines'
        batch sizes of 16-32 rows the step is host-bound otherwise.  Everything a replay must see fresh lives in device
        memory (Adam step counters, epoch accumulators, NaN flags, the dropout draw index); what is baked into the
        graph is in the key: buffers, flag sets, sequences, hyper-parameters, dropout seed, the optimizer's buffers,
        whether the first batch arrives pre-scanned.  `reset_first`: the group is the first of an epoch and its graph
        starts with the reset of the epoch accumulators (a memset node: epoch_reset() as a call of its own costs ~10 us of
        host time in front of the call's first launch).  Returns False when the group cannot be replayed - nothing has
        been launched, not even the reset: the caller runs it eagerly (which is also the warm-up before the capture).
        `dp_tail` (data parallel): a callable (key, fn); every step is then [local sums: chain, k_wgrad, k_reduce without
        Adam] + fn(desc) - the exchange with the other ranks and the Adam / accumulation launch behind it (the one-shot
        exchange kernel, or an all-reduce torch's RCCL binding captures) - and `key` joins the graph key."""
        d = desc if desc is not None else (optimizer.fused_descriptor(self) if hasattr(optimizer, "fused_descriptor") else None)
        if d is None or torch.cuda.is_current_stream_capturing() or self._step_graphs is None:
            return False
        seed = ((int(torch.initial_seed()) ^ (int(getattr(self, "dropout_salt", 0)) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF) \
            if draw_dropout else 0
        if draw_dropout and seed != self._drop_seed:            # a new seed restarts the draw index: do that eagerly
            return False
        first = steps[0][4]
        entry_ready = bool(first.nan_flags and self._prescanned is first)
        entry_drawn = bool(draw_dropout and self._predrawn is first)
        hp = self.group_hp_key(err_penalty, sc_penalty_x001, optimizer, d, seed)
        key = (tuple(st[5] for st in steps), None if nxt is None else nxt[5], entry_ready, entry_drawn, bool(reset_first),
               None if dp_tail is None else dp_tail[0]) + hp
        ent = self._step_graphs.get(key)
        if ent is None:
            if len(self._step_graphs) >= self.MAX_STEP_GRAPHS:
                if self._graph_hits == 0:                       # buffers that never come back (shuffled device batches, ...):
                    self._step_graphs = None                    # stop looking
                    return False
                self._step_graphs.clear()                       # e.g. an LR scheduler changed the baked hyper-parameters:
                self._graph_hits = 0                            # start over with the current ones
            self._step_graphs[key] = [1, None, None, None]      # first sighting: the caller's eager steps are the warm-up
            return False
        if ent[1] is None:
            if ent[0] < 0:                                      # capture failed earlier for this key
                return False
            try:
                hip.check(self.lib.mmn_pack_refresh(self._plan, self._stream()), "mmn_pack_refresh")
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph()
                keep = []
                saved, saved_drawn = self._prescanned, self._predrawn
                with torch.cuda.stream(side):
                    with _capture(graph, side):
                        if reset_first:
                            self.epoch_reset()
                        for i, (xs, y, pairs, bg, b, _k) in enumerate(steps):
                            nb = steps[i + 1][4] if i + 1 < len(steps) else (None if nxt is None else nxt[4])
                            if draw_dropout:
                                keep.append(self.draw_dropout_masks(b))
                            if dp_tail is not None:         # local sums, then the exchange + Adam / accumulation
                                self._launch_step(b, err_penalty, sc_penalty_x001, False, None, nb, None,
                                                  predraw_next=draw_dropout and nb is not None)
                                if not dp_tail[1](d):
                                    raise RuntimeError("data-parallel tail refused during capture")
                            elif not self._launch_step(b, err_penalty, sc_penalty_x001, True, optimizer, nb, d,
                                                       predraw_next=draw_dropout and nb is not None):
                                raise RuntimeError("fusion refused during capture")
                torch.cuda.current_stream().wait_stream(side)
                self._prescanned, self._predrawn = saved, saved_drawn   # nothing has run yet
                ent[1], ent[2] = graph, (steps, nxt, keep, side, optimizer)
            except Exception:
                ent[0] = -1
                self._prescanned = None
                self._predrawn = None
                if dp_tail is None:
                    return False
            if dp_tail is not None and not self._dp_capture_agreed(dp_tail, ent):
                return False
        ent[3] = (hp, entry_ready, entry_drawn, bool(reset_first))
        self.replay_entry(ent, steps, nxt, optimizer, draw_dropout)
        self._last_group_entry = ent                            # (the whole-call plan's shortcut for next time: replay_known)
        return True

    def _dp_capture_agreed(self, dp_tail, ent) -> bool:
        """Data parallel: replay-or-eager is decided by ALL ranks together.  A capture that failed on one rank only would
        leave that rank launching eager collectives against its peers' replayed ones - a different number and order of
        collectives per rank, i.e. a hang.  Every rank contributes its outcome to one MIN all-reduce (outside any capture,
        once per group key); unless all succeeded, all drop the graph and run the group eagerly from here on."""
        import torch.distributed as dist
        group = dp_tail[2] if len(dp_tail) > 2 else None
        ok_here = ent[1] is not None and ent[0] >= 0
        flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32,
                            device=self.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            return True
        ent[0], ent[1], ent[2] = -1, None, None
        self._prescanned = None
        self._predrawn = None
        return False

    @staticmethod
    def group_hp_key(err_penalty, sc_penalty_x001, optimizer, d, seed) -> tuple:
        """What a captured group has baked in besides its buffers: penalties, dropout seed, the optimizer's buffers and
        hyper-parameters."""
        return (float(err_penalty), float(sc_penalty_x001), seed, id(optimizer), d.params, d.grads, d.exp_avg, d.exp_avg_sq,
                d.steps, d.seg_start, d.lr, d.beta1, d.beta2, d.eps, d.weight_decay, d.maximize)

    def replay_entry(self, ent, steps, nxt, optimizer, draw_dropout: bool) -> None:
        """Replay a captured group (run_group's cache entry) and leave the look-ahead bookkeeping as its last step did."""
        hip.check(self.lib.mmn_pack_refresh(self._plan, self._stream()), "mmn_pack_refresh")
        ent[1].replay()
        self._graph_hits += 1
        last = steps[-1][4]
        self._prescanned = nxt[4] if (nxt is not None and nxt[4].nan_flags and last.nan_flags
                                      and nxt[4].nan_flags != last.nan_flags) else None
        self._predrawn = nxt[4] if (draw_dropout and nxt is not None and not nxt[4].tile_seq) else None
        optimizer.mark_fused_step()

    def replay_known(self, ent, steps, nxt, hp, optimizer, draw_dropout: bool, reset_first: bool = False) -> bool:
        """The whole-call replay's shortcut (MultiModN._replay_epoch_plan): `ent` is the cache entry this very group was
        replayed from last time; it is replayed again without rebuilding its key if what the key stands for is unchanged -
        hyper-parameters `hp`, and whether the first batch arrives pre-scanned / pre-drawn."""
        if ent is None or ent[1] is None or ent[3] is None or torch.cuda.is_current_stream_capturing():
            return False
        first = steps[0][4]
        if ent[3] != (hp, bool(first.nan_flags and self._prescanned is first), bool(draw_dropout and self._predrawn is first),
                      bool(reset_first)):
            return False
        self.replay_entry(ent, steps, nxt, optimizer, draw_dropout)
        return True

    def group_entry(self, steps, nxt, err_penalty, sc_penalty_x001, optimizer, draw_dropout: bool, desc, reset_first: bool = False):
        """The cache entry run_group would use for this group right now (None if there is none)."""
        if self._step_graphs is None or desc is None:
            return None, None
        seed = self._dropout_seed() if draw_dropout else 0
        first = steps[0][4]
        hp = self.group_hp_key(err_penalty, sc_penalty_x001, optimizer, desc, seed)
        key = (tuple(st[5] for st in steps), None if nxt is None else nxt[5], bool(first.nan_flags and self._prescanned is first),
               bool(draw_dropout and self._predrawn is first), bool(reset_first), None) + hp      # (None: no data-parallel tail - run_group's key)
        return self._step_graphs.get(key), hp

    # ------------------------------------------------------------------ per-sample mode (BASELINE configs[4])
    def per_sample_batch(self, xs: Sequence[torch.Tensor], y: torch.Tensor, seq: Optional[torch.Tensor]):
        """Group the rows of one mini-batch into 16-row tiles of identical EXECUTED sequence (the
        ordered list of encoders whose modality is present for that sample) - what the fused kernel's
        per-sample mode consumes (include/mmn_hip.h, mmn_batch.tile_rows / tile_seq).

        xs[k]: [B, F] device features of data slot k (NaN anywhere in a row = that sample's modality is
        missing); seq: None (slot k feeds encoder k) or [B, E] int64, sample b feeds slot k to encoder
        seq[b, k].  Everything is torch ops on the device with fixed shapes: no host sync.  Returns
        (hip.Batch, keep-alive tuple); the batch has 16 * n_tiles rows, batch_global = B."""
        E, B, dev = self.E, int(y.shape[0]), self.device
        if E > 8 or len(xs) != E:
            raise UnsupportedModelError("per-sample mode needs one data slot per encoder and E <= 8")
        if E > 4 and seq is not None:
            raise UnsupportedModelError(PER_SAMPLE_SCOPE)
        feats = [int(enc.n_features) for enc in self.model.encoders]
        if seq is not None and len(set(feats)) != 1:
            raise UnsupportedModelError("per-sample encoder order needs modalities of equal width "
                                        "(slot k must be able to feed any encoder)")
        rows = int(self.lib.mmn_regroup_rows(B, E))
        if rows > 0 and not self._torch_regroup:
            # four HIP launches (k_ps_code / k_ps_hist / k_ps_layout / k_ps_gather) instead of ~40 torch ops, any batch size
            self.ensure(rows)
            xs_p = [torch.empty((rows, f), dtype=torch.float32, device=dev) for f in feats]
            y_p = torch.empty((rows, y.shape[1]), dtype=torch.int64, device=dev)
            tile_rows = torch.empty(rows // 16, dtype=torch.int32, device=dev)
            tile_seq = torch.empty(rows // 16, dtype=torch.int32, device=dev)
            bin_ = self.make_batch(xs, y, [(k, k) for k in range(E)], batch_global=B)
            bout = self.make_batch(xs_p, y_p, [(k, k) for k in range(E)], batch_global=B)
            bout.tile_rows, bout.tile_seq = tile_rows.data_ptr(), tile_seq.data_ptr()
            sq = None if seq is None else seq.to(dev, torch.int64).contiguous()
            # the kernels' scratch (codes, presence masks, source row of every position) is this call's own: nothing of
            # the plan's workspace is touched, so the NEXT batch can be regrouped on another stream while a step runs
            scratch = torch.empty(2 * B + rows, dtype=torch.int32, device=dev)
            _check_regroup(self.lib.mmn_regroup_ex(self._plan, C.byref(bin_), None if sq is None else sq.data_ptr(),
                                                   C.byref(bout), scratch.data_ptr(), self._stream()), "mmn_regroup_ex")
            self._ps_layout = ("hip", scratch[2 * B:], B, tile_seq)
            return bout, (xs_p, y_p, tile_rows, tile_seq, sq, xs, y, scratch)
        if E > 4:
            raise UnsupportedModelError("the torch-op regrouping (a test switch) covers at most 4 encoders")
        present = torch.stack([~torch.isnan(x).any(dim=1) for x in xs], dim=1)           # [B, E] slot present
        enc_of = seq.to(dev, torch.int64) if seq is not None else torch.arange(E, device=dev).expand(B, E)
        pi = present.to(torch.int64)
        pos = torch.cumsum(pi, dim=1) - 1
        code = (pi * ((enc_of + 1) << (4 * pos.clamp(min=0)))).sum(dim=1)                 # packed executed sequence
        n_codes = 16 ** E
        counts = torch.bincount(code, minlength=n_codes)
        padded = (counts + 15) // 16 * 16
        base = torch.cumsum(padded, 0) - padded
        first = torch.cumsum(counts, 0) - counts
        order = torch.argsort(code, stable=True)
        scode = code[order]
        where = base[scode] + (torch.arange(B, device=dev) - first[scode])                # padded position of each row
        n_patterns = sum(int(np.prod(range(E - k + 1, E + 1))) for k in range(E + 1))     # ordered subsets of E encoders
        bp = (B + 15) // 16 * 16 + 16 * min(n_patterns, B)
        self.ensure(bp)
        xs_p = []
        for e in range(E):
            src = torch.zeros((B, feats[e]), dtype=torch.float32, device=dev)
            for k in range(E):                                                            # the slot that feeds encoder e
                m = (present[:, k] & (enc_of[:, k] == e)).unsqueeze(1)
                if xs[k].shape[1] == feats[e]:
                    src = torch.where(m, torch.nan_to_num(xs[k]), src)
            xp = torch.zeros((bp, feats[e]), dtype=torch.float32, device=dev)
            xp.index_copy_(0, where, src[order])
            xs_p.append(xp)
        y_p = torch.zeros((bp, y.shape[1]), dtype=torch.int64, device=dev)
        y_p.index_copy_(0, where, y[order])
        tiles = bp // 16
        tile_rows = torch.zeros(tiles, dtype=torch.int32, device=dev)
        tile_rows.index_add_(0, where // 16, torch.ones(B, dtype=torch.int32, device=dev))
        tile_seq = torch.zeros(tiles, dtype=torch.int32, device=dev)
        tile_seq.index_copy_(0, where // 16, scode.to(torch.int32))
        b = self.make_batch(xs_p, y_p, [(k, k) for k in range(E)], batch_global=B)
        b.tile_rows, b.tile_seq = tile_rows.data_ptr(), tile_seq.data_ptr()
        self._ps_layout = ("torch", where, B, tile_seq)
        return b, (xs_p, y_p, tile_rows, tile_seq)

    def per_sample_batch_async(self, xs, y, seq, slot: int, side: "torch.cuda.Stream", template=None):
        """per_sample_batch for the training loop's look-ahead: device-resident inputs, the three regrouping launches on
        stream `side`, outputs in one of two PERSISTENT buffer sets (`slot` 0 / 1; nothing is allocated per step, so no
        allocator bookkeeping across streams).  The caller orders the streams: `side` must have waited for the step that
        last read this slot, the main stream must wait for the returned event before the step.  Returns
        (hip.Batch, keep-alive, event, template) or None when this path does not apply (host tensors, torch regrouping)."""
        E, B, dev = self.E, int(y.shape[0]), self.device
        rows = int(self.lib.mmn_regroup_rows(B, E)) if len(xs) == E and E <= 8 else 0
        feats = [int(enc.n_features) for enc in self.model.encoders]
        if rows <= 0 or self._torch_regroup or (seq is not None and len(set(feats)) != 1):
            return None
        if not (y.is_cuda and y.dtype == torch.int64 and y.is_contiguous() and all(x.is_cuda and x.dtype == torch.float32 for x in xs)
                and (seq is None or (seq.is_cuda and seq.dtype == torch.int64 and seq.is_contiguous()))):
            return None
        self.ensure(rows)
        bufs = self.__dict__.setdefault("_ps_bufs", {})
        key = (slot, B, rows, int(y.shape[1]), self._plan.value)
        ent = bufs.get(key)
        if ent is None:
            for k in [k for k in bufs if k[0] == slot]:
                del bufs[k]
            xs_p = [torch.empty((rows, f), dtype=torch.float32, device=dev) for f in feats]
            y_p = torch.empty((rows, y.shape[1]), dtype=torch.int64, device=dev)
            tile_rows = torch.empty(rows // 16, dtype=torch.int32, device=dev)
            tile_seq = torch.empty(rows // 16, dtype=torch.int32, device=dev)
            scratch = torch.empty(2 * B + rows, dtype=torch.int32, device=dev)
            bout = self.make_batch(xs_p, y_p, [(k, k) for k in range(E)], batch_global=B)
            bout.tile_rows, bout.tile_seq = tile_rows.data_ptr(), tile_seq.data_ptr()
            torch.cuda.current_stream().synchronize()      # (first use: the buffers exist before another stream writes them)
            ent = bufs[key] = (xs_p, y_p, tile_rows, tile_seq, scratch, bout, torch.cuda.Event())
        xs_p, y_p, tile_rows, tile_seq, scratch, bout, ev = ent
        bin_, _, template = self.make_batch_keyed(xs, y, [(k, k) for k in range(E)], B, False, template)
        b = hip.Batch.from_buffer_copy(bout)
        _check_regroup(self.lib.mmn_regroup_ex(self._plan, C.byref(bin_), None if seq is None else seq.data_ptr(),
                                               C.byref(b), scratch.data_ptr(), side.cuda_stream), "mmn_regroup_ex")
        ev.record(side)
        return b, (xs_p, y_p, tile_rows, tile_seq, seq, xs, y, scratch), ev, template

    def run_group_per_sample(self, items, err_penalty: float, sc_penalty_x001: float, optimizer, desc, draw_dropout: bool) -> bool:
        """run_group for per-sample mode.  `items`: consecutive device-resident mini-batches (xs, y, seq or None,
        batch_global) whose buffers come back every epoch.  On its second sighting the group is captured into ONE hipGraph:
        ONE set of four regrouping launches for all its batches (mmn_regroup_multi, every batch into a buffer set of its
        own), then per batch the dropout draw and the step (chain, k_wgrad, k_reduce + Adam).  (`MMN_PS_GRAPH_FORK=1`: the
        regrouping of every batch as launches of its own on a graph branch beside the steps - measured: the cross-queue
        hand-off of a two-branch graph costs ~120 us per group, c5 88 instead of 77 us per step.)  Eager per-sample steps are host-bound (the loop's
        Python and ~10 launches per step); a replayed group is one submission for len(items) steps.  Returns False when the
        group cannot be replayed - nothing has been launched: the caller runs the steps eagerly."""
        import os
        if desc is None or self._step_graphs is None or torch.cuda.is_current_stream_capturing() or self._torch_regroup:
            return False
        E, dev = self.E, self.device
        feats = [int(enc.n_features) for enc in self.model.encoders]
        sizes = []
        for xs, y, seq, bg in items:
            B = int(y.shape[0])
            rows = int(self.lib.mmn_regroup_rows(B, E)) if len(xs) == E and E <= 8 else 0
            if rows <= 0 or (seq is not None and len(set(feats)) != 1):
                return False
            if not (y.is_cuda and y.dtype == torch.int64 and y.is_contiguous() and y.dim() == 2
                    and all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 for x in xs)
                    and (seq is None or (seq.is_cuda and seq.dtype == torch.int64 and seq.is_contiguous()))):
                return False
            sizes.append((B, rows))
        seed = self._dropout_seed() if draw_dropout else 0
        if draw_dropout and seed != self._drop_seed:            # a new seed restarts the draw index: do that eagerly
            return False
        self.ensure(max(r for _, r in sizes))
        hp = self.group_hp_key(err_penalty, sc_penalty_x001, optimizer, desc, seed)
        fork = os.environ.get("MMN_PS_GRAPH_FORK", "0") == "1"
        key = ("per_sample", self._plan.value, fork) + tuple(
            (tuple(x.data_ptr() for x in xs), tuple(x.stride(0) for x in xs), y.data_ptr(), int(y.shape[0]), int(y.shape[1]),
             None if seq is None else seq.data_ptr(), int(bg)) for xs, y, seq, bg in items) + hp
        ent = self._step_graphs.get(key)
        if ent is None:
            if len(self._step_graphs) >= self.MAX_STEP_GRAPHS:
                if self._graph_hits == 0:
                    self._step_graphs = None
                    return False
                self._step_graphs.clear()
                self._graph_hits = 0
            self._step_graphs[key] = [1, None, None, None]      # first sighting: the caller's eager steps are the warm-up
            return False
        if ent[1] is None:
            if ent[0] < 0:
                return False
            try:
                sets = []
                for (xs, y, seq, bg), (B, rows) in zip(items, sizes):
                    xs_p = [torch.empty((rows, f), dtype=torch.float32, device=dev) for f in feats]
                    y_p = torch.empty((rows, y.shape[1]), dtype=torch.int64, device=dev)
                    tile_rows = torch.empty(rows // 16, dtype=torch.int32, device=dev)
                    tile_seq = torch.empty(rows // 16, dtype=torch.int32, device=dev)
                    scratch = torch.empty(2 * B + rows, dtype=torch.int32, device=dev)
                    bin_ = self.make_batch(xs, y, [(k, k) for k in range(E)], batch_global=B)
                    bout = self.make_batch(xs_p, y_p, [(k, k) for k in range(E)], batch_global=int(bg))
                    bout.tile_rows, bout.tile_seq = tile_rows.data_ptr(), tile_seq.data_ptr()
                    sets.append((bin_, bout, scratch, xs_p, y_p, tile_rows, tile_seq))
                hip.check(self.lib.mmn_pack_refresh(self._plan, self._stream()), "mmn_pack_refresh")
                torch.cuda.current_stream().synchronize()      # (the buffers exist before another stream writes them)
                side = torch.cuda.Stream(device=dev)
                branch = torch.cuda.Stream(device=dev) if fork else None
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph()
                keep = []
                saved, saved_drawn = self._prescanned, self._predrawn

                def regroup_all(stream):                    # one set of four launches for the whole group (mmn_regroup_multi)
                    n = len(items)
                    ins = (C.POINTER(hip.Batch) * n)(*[C.pointer(st[0]) for st in sets])
                    outs = (C.POINTER(hip.Batch) * n)(*[C.pointer(st[1]) for st in sets])
                    seqs = (C.c_void_p * n)(*[None if it[2] is None else it[2].data_ptr() for it in items])
                    scr = (C.c_void_p * n)(*[st[2].data_ptr() for st in sets])
                    hip.check(self.lib.mmn_regroup_multi(self._plan, n, ins, seqs, outs, scr, stream.cuda_stream), "mmn_regroup_multi")

                def regroup(k, stream):
                    bin_, bout, scratch = sets[k][:3]
                    seq = items[k][2]
                    hip.check(self.lib.mmn_regroup_ex(self._plan, C.byref(bin_), None if seq is None else seq.data_ptr(),
                                                      C.byref(bout), scratch.data_ptr(), stream.cuda_stream), "mmn_regroup_ex")
                with torch.cuda.stream(side):
                    with _capture(graph, side):
                        evs = []
                        if fork:                            # the regrouping of every batch of the group: one branch beside the steps
                            branch.wait_stream(side)
                            for k in range(len(items)):
                                regroup(k, branch)
                                ev = torch.cuda.Event()
                                ev.record(branch)
                                evs.append(ev)
                        if not fork:
                            regroup_all(side)
                        for k in range(len(items)):
                            if fork:
                                side.wait_event(evs[k])
                            b = sets[k][1]
                            if draw_dropout:
                                keep.append(self.draw_dropout_masks(b))
                            if not self._launch_step(b, err_penalty, sc_penalty_x001, True, optimizer, None, desc):
                                raise RuntimeError("fusion refused during capture")
                        if fork:
                            side.wait_stream(branch)
                torch.cuda.current_stream().wait_stream(side)
                self._prescanned, self._predrawn = saved, saved_drawn   # nothing has run yet
                ent[1], ent[2] = graph, (items, sets, keep, side, branch, optimizer)
            except Exception:
                ent[0] = -1
                self._prescanned = None
                self._predrawn = None
                return False
        hip.check(self.lib.mmn_pack_refresh(self._plan, self._stream()), "mmn_pack_refresh")
        ent[1].replay()
        self._graph_hits += 1
        self._prescanned = None
        self._predrawn = None
        last = ent[2][1][-1]
        self._ps_layout = ("hip", last[2][2 * int(items[-1][1].shape[0]):], int(items[-1][1].shape[0]), last[6])
        optimizer.mark_fused_step()
        return True

    def per_sample_positions(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """For the last per_sample_batch: (position of every original row in the regrouped layout
        [B] int64, packed executed sequence of every original row [B] int64)."""
        kind, info, B, tile_seq = self._ps_layout
        if kind == "torch":
            where = info
        else:                                               # k_ps_layout's source-row table (the call's scratch)
            src_of = info.to(torch.int64)
            pos = torch.nonzero(src_of >= 0).flatten()
            where = torch.empty(B, dtype=torch.int64, device=self.device)
            where[src_of[pos]] = pos
        return where, tile_seq.to(torch.int64)[where // 16]

    def eval_step_collect(self, b: hip.Batch, row: int, out_dst: torch.Tensor, flag_dst: Optional[torch.Tensor],
                          accumulate: bool = True) -> None:
        """eval_step plus, in the same call, the decoders' outputs on grid row `row` -> out_dst ([batch, 2D] float32,
        contiguous) and "the row exists" -> flag_dst (int32 scalar view or None): what test() keeps of a step, without a
        torch copy per kept thing (include/mmn_hip.h mmn_eval_step_ex)."""
        hip.check(self.lib.mmn_eval_step_ex(self._plan, C.byref(b), 1 if accumulate else 0, int(row), out_dst.data_ptr(),
                                            None if flag_dst is None else flag_dst.data_ptr(), self._stream()),
                  "mmn_eval_step_ex")

    def eval_step(self, b: hip.Batch, accumulate: bool = False) -> None:
        hip.check(self.lib.mmn_eval_step(self._plan, C.byref(b), 1 if accumulate else 0, self._stream()), "mmn_eval_step")

    def nan_scan(self, b: hip.Batch) -> None:
        """The NaN scan of a batch that carries device flags, on its own launch (data parallel, first step of an epoch:
        the flags are summed across the ranks - `flag_tail` - before the step; every later batch is pre-scanned by the
        step before it and its flags ride in that step's all-reduce).  The step that follows finds the flags ready."""
        hip.check(self.lib.mmn_nan_scan(self._plan, C.byref(b), self._stream()), "mmn_nan_scan")
        self._prescanned = b

    def epoch_small_rows(self) -> int:
        """Largest batch (rows) the one-launch epoch kernel takes for this model, 0 if it does not apply (mmn_epoch_small_rows)."""
        c = self.__dict__.get("_eps_rows")
        if c is None or c[0] != self._plan.value:
            c = self.__dict__["_eps_rows"] = (self._plan.value, int(self.lib.mmn_epoch_small_rows(self._plan)))
        return c[1]

    def adam_fusable(self, optimizer, desc=None) -> bool:
        """True if `optimizer`'s step can ride in this engine's launches: a multimodn_amd.optim.Adam over exactly this
        model's parameters, one group, one contiguous run whose layout the library accepts.  Only then does the fused
        step leave a skipped encoder's parameters, moments and step count untouched (torch's grad-None behaviour)."""
        d = desc if desc is not None else (optimizer.fused_descriptor(self) if hasattr(optimizer, "fused_descriptor") else None)
        return d is not None and self.lib.mmn_adam_fusable(self._plan, C.byref(d)) == 0

    def accumulate_and_step(self, err_penalty: float, sc_penalty_x001: float, optimizer, desc=None) -> bool:
        """Data-parallel tail after the all-reduce: epoch accumulation + the optimizer's Adam step in ONE
        launch when `optimizer` is a multimodn_amd.optim.Adam over this model (returns True: its next
        .step() is a no-op); otherwise only the accumulation (returns False)."""
        d = desc if desc is not None else (optimizer.fused_descriptor(self) if hasattr(optimizer, "fused_descriptor") else None)
        if d is None:
            self.accumulate(err_penalty, sc_penalty_x001)
            return False
        hip.check(self.lib.mmn_adam_step_accumulate(self._plan, C.byref(d), err_penalty, sc_penalty_x001, self._stream()),
                  "mmn_adam_step_accumulate")
        optimizer.mark_fused_step()
        return True

    # ---- one-shot data-parallel exchange (opt-in; include/mmn_hip.h mmn_dp_oneshot_attach)
    def attach_oneshot(self, group, world: int, rank: int, spin_ms: int = 5000) -> bool:
        """Collective over `group`: every rank allocates its exchange buffer, the 64-byte hipIpc handles travel through
        torch.distributed's object all-gather, every rank maps the peers' buffers and hands all of them to the plan.
        To be called again after a re-plan (the buffers are sized by the plan).
        Failure-atomic across the ranks: every rank reports whether its own allocation / mapping / attach worked, and
        unless ALL did, every rank closes what it opened and returns False (the caller then keeps the all-reduce path) -
        no rank is left waiting at a barrier its peer never reaches.  Ranks on different hosts cannot map each other's
        device memory: refused the same way.  Ranks on DIFFERENT devices of the node ask for fine-grained exchange buffers
        (MMN_DP_XBUF_FINE, unless the environment says otherwise): the kernel's system-scope accesses do not depend on it,
        but two processes on one GPU - all the tests here can build - share an L2 and could not tell."""
        import os
        import socket
        import torch.distributed as dist
        self.detach_oneshot()
        nbytes = int(self.lib.mmn_dp_xbuf_bytes(self._plan))
        who = [None] * world
        dist.all_gather_object(who, (socket.gethostname(), int(self.device.index if self.device.index is not None else torch.cuda.current_device())),
                               group=group)
        same_host = all(h[0] == who[0][0] for h in who)
        if len({h[1] for h in who}) > 1 and "MMN_DP_XBUF_FINE" not in os.environ:
            os.environ["MMN_DP_XBUF_FINE"] = "1"
        own = C.c_void_p()
        handle = C.create_string_buffer(64)
        ok = nbytes > 0 and same_host
        if ok:
            ok = self.lib.mmn_dp_xbuf_alloc(nbytes, C.byref(own), handle) == 0
        handles = [None] * world
        dist.all_gather_object(handles, (nbytes if ok else -1, bytes(handle.raw)), group=group)
        opened = []
        if ok and all(h[0] == nbytes for h in handles):
            ptrs = (C.c_void_p * world)()
            for r in range(world):
                if r == rank:
                    ptrs[r] = own.value
                    continue
                peer = C.c_void_p()
                if self.lib.mmn_dp_xbuf_open(handles[r][1], C.byref(peer)) != 0:
                    ok = False
                    break
                opened.append(peer.value)
                ptrs[r] = peer.value
            if ok:
                ok = self.lib.mmn_dp_oneshot_attach(self._plan, world, rank, ptrs, int(spin_ms)) == 0
        else:
            ok = False
        verdict = [None] * world
        dist.all_gather_object(verdict, bool(ok), group=group)     # (doubles as the barrier: nobody steps before everybody has mapped everything)
        if not all(verdict):
            self.lib.mmn_dp_oneshot_detach(self._plan)          # (this rank's attach may have succeeded: the plan must not keep
            for ptr in opened:                                  #  the addresses of buffers that are closed below)
                self.lib.mmn_dp_xbuf_close(ptr, 0)
            if own.value:
                self.lib.mmn_dp_xbuf_close(own.value, 1)
            self._oneshot = None
            self._oneshot_refused = self._plan.value            # (do not try again for this plan)
            return False
        self._oneshot = {"ptrs": [ptrs[r] for r in range(world)], "rank": rank, "plan": self._plan.value}
        return True

    def detach_oneshot(self) -> None:
        st = getattr(self, "_oneshot", None)
        self._oneshot = None
        if st is None:
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self._plan is not None and st["plan"] == self._plan.value:
            self.lib.mmn_dp_oneshot_detach(self._plan)
        for r, ptr in enumerate(st["ptrs"]):
            self.lib.mmn_dp_xbuf_close(ptr, 1 if r == st["rank"] else 0)

    def oneshot_attached(self) -> bool:
        st = getattr(self, "_oneshot", None)
        return st is not None and self._plan is not None and st["plan"] == self._plan.value

    def oneshot_check(self) -> None:
        """Raises once a wait of the exchange ran out (a peer died or fell minutes behind): no synchronisation, one read
        of a host-mapped word."""
        if getattr(self, "_oneshot", None) is not None:
            rc = self.lib.mmn_dp_oneshot_error(self._plan)
            if rc != 0:
                d = (C.c_uint32 * 8)()
                self.lib.mmn_dp_oneshot_diag(self._plan, d)
                hip.check(rc, f"one-shot data-parallel exchange [rank {d[5]} of {d[6]}: chunk {d[1]} waited for step {d[2]} of peer "
                              f"{int(d[0]) - 1}, last saw flag {d[3]} (other parity {d[4]}); own counter of that chunk {d[7]}]")

    def accumulate_and_step_oneshot(self, err_penalty: float, sc_penalty_x001: float, optimizer, desc=None) -> bool:
        """Data-parallel tail WITHOUT a collective in front of it: exchange of [grads | stats] with the peers' mapped
        buffers, rank-ordered sum, Adam and epoch accumulation in one launch.  Returns False (nothing launched) when
        the optimizer cannot be fused: the caller then takes the all-reduce path."""
        d = desc if desc is not None else (optimizer.fused_descriptor(self) if hasattr(optimizer, "fused_descriptor") else None)
        if d is None or self.lib.mmn_adam_fusable(self._plan, C.byref(d)) != 0:
            return False
        rc = self.lib.mmn_adam_step_accumulate_oneshot(self._plan, C.byref(d), err_penalty, sc_penalty_x001, self._stream())
        if rc == hip.ERR_PEER:
            self.oneshot_check()                            # (raises with what the wait that ran out was looking at)
        hip.check(rc, "mmn_adam_step_accumulate_oneshot")
        optimizer.mark_fused_step()
        return True

    def dp_rescale(self, nominal_batch: int, with_grads: bool = True) -> None:
        """Data parallel with uneven shards, after the all-reduce and in front of accumulate*(): the step ran with
        batch_global = nominal_batch on every rank; gradients, loss cells and state changes become means over the
        true global batch (the summed row count of grid row 0), include/mmn_hip.h mmn_dp_rescale."""
        hip.check(self.lib.mmn_dp_rescale(self._plan, self.flat_grads.data_ptr() if with_grads else None,
                                          self.n_params if with_grads else 0, int(nominal_batch), self._stream()),
                  "mmn_dp_rescale")

    def accumulate(self, err_penalty: float, sc_penalty_x001: float) -> None:
        hip.check(self.lib.mmn_epoch_accumulate(self._plan, err_penalty, sc_penalty_x001, self._stream()),
                  "mmn_epoch_accumulate")

    def assign_grads(self, executed: Optional[Sequence[bool]] = None) -> None:
        """Point every Parameter's .grad at its slice of the flat gradient buffer; encoders that
        were skipped (NaN batch) get grad None, as autograd would leave them (multimodn.py:168)."""
        skipped = set()
        if executed is not None:
            for e, ran in enumerate(executed):
                if not ran:
                    skipped.update(self.enc_param_ids[e])
        for p, g in zip(self.params, self.grad_views):
            p.grad = None if id(p) in skipped else g

    # ------------------------------------------------------------------ epoch accumulators
    def epoch_reset(self) -> None:
        hip.check(self.lib.mmn_epoch_reset(self._plan, self._stream()), "mmn_epoch_reset")

    def epoch_read(self) -> Dict[str, np.ndarray]:
        out = np.zeros(self.n_epoch, np.float64)
        hip.check(self.lib.mmn_epoch_read(self._plan, out.ctypes.data_as(C.POINTER(C.c_double)), self._stream()),
                  "mmn_epoch_read")
        return split_epoch(out, self.E, self.D)

    def epoch_read_async(self):
        """The epoch accumulators on their way to pinned host memory, WITHOUT synchronising: returns (wait, fetch) - wait()
        blocks until the copy has landed, fetch() then gives the same dict as epoch_read().  The copy is enqueued on the
        current stream, i.e. behind the epoch's last launch and in front of the next epoch's reset.  A small ring of pinned
        buffers; when every buffer is in flight the oldest is waited for first."""
        ring = self.__dict__.setdefault("_ep_ring", [])
        slot = None
        for ent in ring:
            if ent["free"]:
                slot = ent
                break
        if slot is None:
            if len(ring) < 8:
                slot = {"buf": torch.empty(self.n_epoch, dtype=torch.float64, pin_memory=True), "ev": torch.cuda.Event(), "free": True}
                ring.append(slot)
            else:
                return None                                 # (the caller falls back to the synchronous read)
        ptr = self.lib.mmn_debug_buffer(self._plan, 8, 0)
        off = ptr - self.workspace.data_ptr()
        view = self.workspace[off:off + 8 * self.n_epoch].view(torch.float64)
        slot["free"] = False
        slot["buf"].copy_(view, non_blocking=True)
        slot["ev"].record()
        E, D = self.E, self.D

        def wait():
            slot["ev"].synchronize()

        def fetch():
            out = slot["buf"].numpy().copy()
            slot["free"] = True
            return split_epoch(out, E, D)
        return wait, fetch

    def step_values(self) -> Dict[str, np.ndarray]:
        """Last step's stats block (synchronises)."""
        return split_stats(self.stats.detach().cpu().numpy(), self.E, self.D)

    # ------------------------------------------------------------------ what a forward-only step leaves behind
    def state_rows(self, e: int, batch: int) -> torch.Tensor:
        """[batch, S] view of the state after encoder e of the last step (any step kind)."""
        return self.debug_tensor(0, e + 1, self.max_batch, self.S)[:batch]

    def decoder_outputs(self, row: int, batch: int) -> torch.Tensor:
        """[batch, 2D] view: sigmoid outputs of every decoder on grid row `row` of the last
        eval_step (decoder d at columns 2d, 2d+1).  After a TRAINING step this buffer holds dz."""
        return self.debug_tensor(1, row, self.max_batch, 2 * self.D)[:batch]

    def executed_flags(self) -> torch.Tensor:
        """int32 [E+1] device view: 1 where the last step produced the state row (row 0 always 1).  No sync."""
        return self.debug_tensor(4, 0, 1, self.E + 1).view(torch.int32).flatten()

    def executed_rows(self) -> List[bool]:
        """Which state rows the last step produced (synchronises; the device NaN policy needs it)."""
        return [bool(v) for v in self.debug_tensor(4, 0, 1, self.E + 1).view(torch.int32).flatten().tolist()]

    def debug_tensor(self, kind: int, index: int, rows: int, cols: int) -> torch.Tensor:
        ptr = self.lib.mmn_debug_buffer(self._plan, kind, index)
        if not ptr:
            raise IndexError((kind, index))
        off = (ptr - self._ws_ptr)
        raw = self.workspace[(self._ws_ptr - self.workspace.data_ptr()) + off:]
        return raw[:rows * cols * 4].view(torch.float32).view(rows, cols)


def split_stats(st: np.ndarray, E: int, D: int) -> Dict[str, np.ndarray]:
    R = E + 1
    RD = R * D
    o = RD + E
    names = ("n_correct", "tp", "tn", "fp", "fn")
    out = {"err_loss": st[:RD].reshape(R, D), "state_change": st[RD:RD + E]}
    for i, n in enumerate(names):
        out[n] = st[o + i * RD:o + (i + 1) * RD].reshape(R, D)
    out["rows"] = st[o + 5 * RD:o + 5 * RD + R]
    tail = st[o + 5 * RD + R:]
    out["loss"], out["global_err"], out["global_sc"] = tail[0], tail[1], tail[2]
    return out


def split_epoch(ep: np.ndarray, E: int, D: int) -> Dict[str, np.ndarray]:
    R = E + 1
    RD = R * D
    o = RD + E
    names = ("n_correct", "tp", "tn", "fp", "fn")
    out = {"err_sum": ep[:RD].reshape(R, D).copy(), "sc_sum": ep[RD:RD + E].copy()}
    for i, n in enumerate(names):
        out[n] = ep[o + i * RD:o + (i + 1) * RD].reshape(R, D).copy()
    out["rows"] = ep[o + 5 * RD:o + 5 * RD + R].copy()
    out["n_steps"] = ep[o + 5 * RD + R]
    return out
