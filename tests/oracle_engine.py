"""TEST-ONLY step backend: the same engine interface as multimodn_amd.engine.HipChainEngine, with
the arithmetic done by the numpy oracle on the CPU.  It exists so that the HOST logic of the
package (batch ingest, NaN policy, sequence handling, data-parallel protocol, epoch aggregation,
History) can be tested here without a GPU.  It lives under tests/ and is never imported by the
product; MultiModN's default engine factory is the HIP engine, which refuses to run off-GPU."""
import numpy as np
import torch

from multimodn_amd.engine import activation_code, check_supported, split_epoch, split_stats
from oracle import multimodn_oracle as O


class _Batch:
    pass


class OracleEngine:
    def __init__(self, model, max_batch):
        check_supported(model)
        self.model = model
        self.params = list(model.parameters())
        self.names = [n for n, _ in model.named_parameters()]
        self.E, self.D, self.S = len(model.encoders), len(model.decoders), model.init_state.state_size
        from multimodn_amd.encoders import MIMIC_MLPEncoder
        from multimodn_amd.decoders import MLPDecoder
        encs = []
        for enc in model.encoders:
            if isinstance(enc, MIMIC_MLPEncoder):
                encs.append(O.EncoderSpec(enc.n_features, tuple(enc.hidden_layers), activation_code(enc.activation),
                                          kind="mimic", dropout=float(enc.dropout)))
                continue
            act = activation_code(enc.activation) if len(enc.layers) > 1 else O.ACT_IDENTITY
            encs.append(O.EncoderSpec(enc.n_features, tuple(enc.hidden_layers), act))
        decs = [O.DecoderSpec("mlp", tuple(l.out_features for l in list(dec.layers)[:-1]),
                              activation_code(dec.hidden_activation) if len(dec.layers) > 1 else O.ACT_IDENTITY)
                if isinstance(dec, MLPDecoder) else O.DecoderSpec() for dec in model.decoders]
        self.spec = O.ModelSpec(self.S, encs, self.D, float(model.err_penalty), float(model.state_change_penalty) / 0.01,
                                decoders=decs)
        self.dropout_encoders = [(e, enc.n_features + self.S, float(enc.dropout)) for e, enc in enumerate(model.encoders)
                                 if isinstance(enc, MIMIC_MLPEncoder) and enc.dropout > 0]
        self.n_params = sum(p.numel() for p in self.params)
        R = self.E + 1
        self.NF = 16                                        # hip.MAX_ENCODERS: words per NaN-flag set
        self.n_stats = R * self.D + self.E + 5 * R * self.D + R + 4 + 2 * self.NF     # ... + the two flag sets
        self.reduce_buf = torch.zeros(self.n_params + self.n_stats)
        self.flat_grads = self.reduce_buf[:self.n_params]
        self.stats = self.reduce_buf[self.n_params:]
        self.grad_views, off = [], 0
        for p in self.params:
            self.grad_views.append(self.flat_grads[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.enc_param_ids = [[id(p) for p in enc.parameters()] for enc in model.encoders]
        self.epoch = np.zeros(R * self.D + self.E + 5 * R * self.D + R + 1)
        self.max_batch = max_batch
        # the flag protocol of HipChainEngine (engine.py: make_batch / nan_scan / local_step): two sets at the end of
        # the stats block, handed out in turn; a step consumes (and re-zeroes) its batch's set and pre-scans the next
        # batch into the other one, so that under data parallel the flags ride in the step's one all-reduce
        self.flag_tail = self.stats[self.n_stats - 2 * self.NF:]
        self._flag_turn = 0
        self._prescanned = None

    def ensure(self, batch):
        self.max_batch = max(self.max_batch, batch)
        return False

    def begin_sequence(self):
        self._flag_turn = 0
        self._prescanned = None

    def adam_fusable(self, optimizer, desc=None):
        return False

    def _set_of(self, b):
        return self.flag_tail[(b.nan_flags - 1) * self.NF:b.nan_flags * self.NF]

    def _scan(self, b):
        """k_prepare's / k_reduce's scan blocks: raise the word of every data slot of b's sequence that holds a NaN."""
        fl = self._set_of(b)
        for k, _ in b.pairs:
            if np.isnan(b.xs[k]).any():
                fl[k] = 1.0

    def nan_scan(self, b):
        self._scan(b)
        self._prescanned = b

    def epoch_reset(self):
        self.epoch[:] = 0

    def make_batch(self, xs, y, pairs, batch_global=None, device_nan_flags=False):
        b = _Batch()
        b.xs = [x.numpy() for x in xs]
        b.y = y.numpy()
        b.pairs = list(pairs)
        b.batch_global = batch_global or len(b.y)
        b.device_nan = device_nan_flags
        b.nan_flags = None
        if device_nan_flags:
            b.nan_flags = 1 + self._flag_turn                # (truthy like the device pointer it stands for)
            self._flag_turn ^= 1
        b.tile_seq = None
        b.masks = None
        return b

    def make_batch_keyed(self, xs, y, pairs, batch_global=None, device_nan_flags=False, template=None):
        return self.make_batch(xs, y, pairs, batch_global, device_nan_flags), None, None

    def draw_dropout_masks(self, b, provider=None):
        """HipChainEngine.draw_dropout_masks on the host (torch's CPU generator)."""
        b.masks = {}
        running = {e for _, e in b.pairs}
        for e, width, p in self.dropout_encoders:
            if e not in running:
                continue
            mk = provider(e, len(b.y), width) if provider is not None else \
                torch.empty((len(b.y), width)).bernoulli_(1.0 - p).div_(1.0 - p)
            if mk is not None:
                b.masks[e] = mk.numpy()
        return list(b.masks.values())

    def reset_dropout(self):
        pass

    def per_sample_batch(self, xs, y, seq):
        b = _Batch()
        b.xs = [x.numpy() for x in xs]
        b.y = y.numpy()
        b.seq = None if seq is None else seq.numpy()
        b.batch_global = len(b.y)
        b.per_sample = True
        b.nan_flags = None
        b.tile_seq = True
        return b, None

    def _run(self, b, want_grads):
        params = {n: p.detach().numpy() for n, p in zip(self.names, self.params)}
        if getattr(b, "per_sample", False):
            r = O.per_sample_step(params, self.spec, b.xs, b.y, b.seq, batch_global=b.batch_global)   # shard: global divisors
            return self._publish(r, want_grads, rows=r.row_counts.astype(np.float32))
        n_slots = max(k for k, _ in b.pairs) + 1 if b.pairs else 0
        seq = None
        xs = b.xs
        if b.pairs:
            # oracle takes a [B, n_seq] sequence whose t-th entry feeds data slot t
            xs = [b.xs[k] for k, _ in b.pairs]
            seq = np.tile(np.array([e for _, e in b.pairs], np.int64), (len(b.y), 1))
        override = [True] * len(b.pairs)
        if b.device_nan:                                    # the flag words decide (summed over the ranks under DP)
            fl = self._set_of(b)
            override = [float(fl[k]) == 0.0 for k, _ in b.pairs]
        spec = self.spec
        if not b.pairs:
            xs, seq, override = [], np.zeros((len(b.y), 0), np.int64), []
        r = O.forward_backward(params, spec, xs, b.y, seq, batch_global=b.batch_global,
                               present_override=override, want_grads=want_grads, keep_states=True,
                               drop_masks=getattr(b, "masks", None))
        self._last = (params, r)
        rows = np.zeros(self.E + 1, np.float32)
        rows[0] = len(b.y)
        rows[1:][r.executed] = len(b.y)
        return self._publish(r, want_grads, rows)

    def _publish(self, r, want_grads, rows):
        R, D, E = self.E + 1, self.D, self.E
        st = np.zeros(self.n_stats, np.float32)
        RD = R * D
        st[:RD] = r.err_loss.reshape(-1)
        st[RD:RD + E] = r.state_change
        for i, k in enumerate(("n_correct", "tp", "tn", "fp", "fn")):
            st[RD + E + i * RD:RD + E + (i + 1) * RD] = getattr(r, k).reshape(-1)
        st[RD + E + 5 * RD:RD + E + 5 * RD + R] = rows
        nv = self.n_stats - 2 * self.NF                      # (the flag sets behind the step values are not touched)
        self.stats[:nv].copy_(torch.from_numpy(st[:nv]))
        if want_grads:
            flat = np.concatenate([np.zeros(p.numel(), np.float32) if r.grads[n] is None
                                   else np.asarray(r.grads[n], np.float32).reshape(-1)
                                   for n, p in zip(self.names, self.params)])
            self.flat_grads.copy_(torch.from_numpy(flat))

    def _consume_flags(self, b, next_batch=None):
        """Before the step: scan unless an earlier step pre-scanned this batch.  After it (k_reduce): the consumed set
        is zero again and the next batch's flags stand in the other set."""
        if b.nan_flags is not None and self._prescanned is not b:
            self._set_of(b)[:] = 0
            self._scan(b)

    def _after_step(self, b, next_batch=None):
        self._prescanned = None
        if b.nan_flags is not None:
            self._set_of(b)[:] = 0
            if next_batch is not None and next_batch.nan_flags is not None and next_batch.nan_flags != b.nan_flags:
                self._scan(next_batch)
                self._prescanned = next_batch

    def local_step(self, b, alpha, beta, accumulate=False, optimizer=None, next_batch=None, predraw_next=False, desc=None):
        if not getattr(b, "per_sample", False):
            self._consume_flags(b)
        self._run(b, True)
        if not getattr(b, "per_sample", False):
            self._after_step(b, next_batch)
        if accumulate:
            self.accumulate(alpha, beta)
        return False                     # never applies the optimizer step itself

    def eval_step(self, b, accumulate=False):
        spec = self.spec
        import dataclasses
        self.spec = dataclasses.replace(spec, err_penalty=1.0, state_change_penalty=0.0)
        try:
            if not getattr(b, "per_sample", False):
                self._consume_flags(b)
            self._run(b, False)
            if not getattr(b, "per_sample", False):
                self._after_step(b)
        finally:
            self.spec = spec
        if accumulate:
            self.accumulate(1.0, 0.0)

    # what a step leaves behind (HipChainEngine.state_rows / decoder_outputs / executed_rows)
    def _state(self, row, batch):
        """Row of a skipped encoder: the HIP engine's buffer holds stale data there (callers mask it); zeros here."""
        st = self._last[1].states
        return st[row][:batch] if row in st else np.zeros((batch, self.S), np.float32)

    def state_rows(self, e, batch):
        return torch.from_numpy(np.ascontiguousarray(self._state(e + 1, batch)))

    def decoder_outputs(self, row, batch):
        params, r = self._last
        o = O.decoder_outputs(params, self.spec, self._state(row, batch))
        return torch.from_numpy(np.ascontiguousarray(o.reshape(o.shape[0], -1)))

    def executed_flags(self):
        return torch.tensor([1] + [int(v) for v in self._last[1].executed], dtype=torch.int32)

    def executed_rows(self):
        return [True] + [bool(v) for v in self._last[1].executed]

    def accumulate_and_step(self, alpha, beta, optimizer, desc=None):
        self.accumulate(alpha, beta)
        return False                     # the caller's optimizer.step() still has to run

    def dp_rescale(self, nominal_batch, with_grads=True):
        """numpy twin of k_dp_rescale."""
        st = self.stats.numpy()
        R, D, E = self.E + 1, self.D, self.E
        RD = R * D
        f = np.float32(nominal_batch) / st[RD + E + 5 * RD]
        st[:RD + E] *= f
        if with_grads:
            self.flat_grads.mul_(float(f))

    def accumulate(self, alpha, beta):
        """numpy twin of k_epoch_accumulate."""
        st = self.stats.numpy()
        R, D, E = self.E + 1, self.D, self.E
        RD = R * D
        ge = np.float32(st[:RD].sum(dtype=np.float32) / np.float32(D * R))
        gs = np.float32(st[RD:RD + E].sum(dtype=np.float32) / np.float32(E))
        tail = RD + E + 5 * RD + R
        st[tail] = ge * np.float32(alpha) + gs * np.float32(beta)
        st[tail + 1], st[tail + 2] = ge, gs
        ep = self.epoch
        ep[:RD] += st[:RD].astype(np.float64)
        ep[RD:RD + E] += st[RD:RD + E].astype(np.float64)
        o = RD + E
        ep[o:o + RD] += st[o:o + RD]
        for k in range(1, 5):
            sl = slice(o + k * RD, o + (k + 1) * RD)
            ep[sl] = (ep[sl].astype(np.float32) + st[sl]).astype(np.float64)
        ep[o + 5 * RD:o + 5 * RD + R] += st[o + 5 * RD:o + 5 * RD + R]
        ep[-1] += 1

    def assign_grads(self, executed=None):
        skipped = set()
        if executed is not None:
            for e, ran in enumerate(executed):
                if not ran:
                    skipped.update(self.enc_param_ids[e])
        for p, g in zip(self.params, self.grad_views):
            p.grad = None if id(p) in skipped else g

    def epoch_read(self):
        return split_epoch(self.epoch.copy(), self.E, self.D)

    def step_values(self):
        return split_stats(self.stats.numpy().copy(), self.E, self.D)
