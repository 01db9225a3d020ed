"""CPU ORACLE for the MultiModN sequential-fusion training step.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm
(EPFLiGHT/MultiModN, `multimodn/multimodn.py:89-252` train_epoch and the plugin
modules it drives).  It is the *checker* for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
Nothing under `multimodn_amd/` imports or calls it; the product path raises when
the HIP library is missing instead of falling back here.

Pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported in the build container with three import shims
(`tests/golden/make_golden.py`, which also commits the resulting vectors under
`tests/golden/*.npz`).  `tests/test_oracle_golden.py` re-checks the oracle
against those vectors on every run.

Arithmetic: the reference computes in fp32 through ATen (third-party: PyTorch,
pinned `torch==1.13.1` in requirements-cpu.txt; golden vectors were generated
with torch 2.10.0 whose Linear/relu/sigmoid/log_softmax/nll_loss/Adam semantics
are unchanged).  Here every op is restated from its published definition in
numpy at a caller-chosen dtype (float32 to mimic, float64 as truth).

Parameter naming follows the reference `state_dict()` keys:
    init_state.state_value                 [1, S]        (state.py:25-27)
    encoders.{e}.layers.{l}.weight / .bias               (mlp_encoder.py:64-72)
    decoders.{d}.fc.weight / .bias         [2, S] / [2]  (decoders.py:16)
MIMIC family: MIMIC_MLPEncoder (mlp_encoder.py:9-47) keeps an nn.Dropout at
layers.0, so its Linear l is `encoders.{e}.layers.{l+1}`; MLPDecoder (decoders.py:22-46) names
its Linears `decoders.{d}.layers.{l}`.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ACT_IDENTITY, ACT_RELU, ACT_SIGMOID = 0, 1, 2
_ACT_NAMES = {"identity": ACT_IDENTITY, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID}


@dataclass
class EncoderSpec:
    """MLPEncoder shape (mlp_encoder.py:51-72).  hidden=() gives SLP/Linear/LogisticEncoder
    (slp_encoders.py:5-34) whose activation is never applied."""
    n_features: int
    hidden: Tuple[int, ...] = ()
    activation: int = ACT_RELU
    # "mlp": MLPEncoder.  "mimic": MIMIC_MLPEncoder (mlp_encoder.py:9-47): Dropout(p) on
    # cat([x, state]) feeds the FIRST Linear, the activation follows EVERY Linear incl. the last.
    kind: str = "mlp"
    dropout: float = 0.0

    def layer_key(self, l: int) -> int:
        """Index of Linear l inside the module's `layers` list (state_dict key)."""
        return l + 1 if self.kind == "mimic" else l

    def layer_shapes(self, state_size: int) -> List[Tuple[int, int]]:
        if self.kind == "mimic":                # mlp_encoder.py:24-29
            dims = [self.n_features + state_size] + list(self.hidden) + [state_size]
            return [(dout, din) for din, dout in zip(dims, dims[1:])]
        dims = [self.n_features] + list(self.hidden) + [state_size]
        shapes = []
        for i, (din, dout) in enumerate(zip(dims, dims[1:])):
            if i == len(dims) - 2:          # state concatenated into the LAST layer only
                shapes.append((dout, din + state_size))
            else:
                shapes.append((dout, din))
        return shapes


@dataclass
class DecoderSpec:
    """kind "class": ClassDecoder(n_classes=2, sigmoid) = LogisticDecoder (decoders.py:9-20,49-53).
    kind "mlp": MLPDecoder (decoders.py:22-46): hidden Linear + hidden_activation layers, then
    sigmoid(Linear(-> 2))."""
    kind: str = "class"
    hidden: Tuple[int, ...] = ()
    hidden_activation: int = ACT_RELU

    def names(self, d: int) -> List[str]:
        if self.kind == "class":
            return [f"decoders.{d}.fc"]
        return [f"decoders.{d}.layers.{l}" for l in range(len(self.hidden) + 1)]

    def layer_shapes(self, state_size: int) -> List[Tuple[int, int]]:
        dims = [state_size] + (list(self.hidden) if self.kind == "mlp" else []) + [2]
        return [(dout, din) for din, dout in zip(dims, dims[1:])]


@dataclass
class ModelSpec:
    state_size: int
    encoders: List[EncoderSpec]
    n_decoders: int
    err_penalty: float = 1.0
    state_change_penalty: float = 0.0       # the *user* value; x0.01 applied here (multimodn.py:86)
    decoders: Optional[List[DecoderSpec]] = None      # None: n_decoders LogisticDecoders

    def dec(self, d: int) -> DecoderSpec:
        return self.decoders[d] if self.decoders is not None else DecoderSpec()

    @property
    def E(self) -> int:
        return len(self.encoders)

    @property
    def D(self) -> int:
        return self.n_decoders

    def param_names(self) -> List[str]:
        """named_parameters() order: init_state, encoders.*, decoders.* (SURVEY 9.11)."""
        names = ["init_state.state_value"]
        for e, enc in enumerate(self.encoders):
            for l in range(len(enc.hidden) + 1):
                k = enc.layer_key(l)
                names += [f"encoders.{e}.layers.{k}.weight", f"encoders.{e}.layers.{k}.bias"]
        for d in range(self.n_decoders):
            for n in self.dec(d).names(d):
                names += [n + ".weight", n + ".bias"]
        return names

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        S = self.state_size
        shapes: Dict[str, Tuple[int, ...]] = {"init_state.state_value": (1, S)}
        for e, enc in enumerate(self.encoders):
            for l, (o, i) in enumerate(enc.layer_shapes(S)):
                k = enc.layer_key(l)
                shapes[f"encoders.{e}.layers.{k}.weight"] = (o, i)
                shapes[f"encoders.{e}.layers.{k}.bias"] = (o,)
        for d in range(self.n_decoders):
            dec = self.dec(d)
            for n, (o, i) in zip(dec.names(d), dec.layer_shapes(S)):
                shapes[n + ".weight"] = (o, i)
                shapes[n + ".bias"] = (o,)
        return shapes


def init_params(spec: ModelSpec, seed: int, dtype=np.float32) -> Dict[str, np.ndarray]:
    """Synthetic parameters with nn.Linear-like scale (uniform +-1/sqrt(fan_in)) and a randn
    init state.  NOT the torch RNG stream; parity tests load captured weights instead."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in spec.param_shapes().items():
        if name == "init_state.state_value":
            params[name] = rng.standard_normal(shape).astype(dtype)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:   # bias: fan_in of its weight
                fan_in = spec.param_shapes()[name[:-4] + "weight"][1]
            bound = 1.0 / np.sqrt(fan_in)
            params[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
    return params


def _act(a: np.ndarray, kind: int) -> np.ndarray:
    if kind == ACT_RELU:
        return np.maximum(a, 0)
    if kind == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-a))
    return a


def _act_grad_from_output(h: np.ndarray, kind: int) -> np.ndarray:
    """d act / d pre expressed through the activation OUTPUT h."""
    if kind == ACT_RELU:
        return (h > 0).astype(h.dtype)
    if kind == ACT_SIGMOID:
        return h * (1 - h)
    return np.ones_like(h)


#: relative distance to zero under which a relu pre-activation counts as "on the kink" (fp32 implementations place a
#: pre-activation within ~1e-6 of the exact value, relative to the layer's scale)
KINK_MARGIN = 4e-6


def _note_kinks(kinks, key, pre, activation) -> None:
    if kinks is None or activation != ACT_RELU:
        return
    scale = float(np.abs(pre).max())
    if scale > 0:
        kinks[key] = kinks.get(key, 0) + int((np.abs(pre) < KINK_MARGIN * scale).sum())


def tied_parameters(spec: "ModelSpec", kinks: Dict[tuple, int]) -> set:
    """Names of the parameters whose gradient is discontinuous at the relu kinks `kinks` found (forward_backward(...,
    kinks=...)): on a sample whose pre-activation sits on the kink, act'(.) is 0 or 1 depending on the last rounding, and
    that sample's whole contribution to the gradients UPSTREAM of the unit comes or goes - two correct fp32
    implementations may then differ by one sample's share (~1 / sqrt(batch) of a gradient element), far above rounding
    noise.  MLPEncoder family: the hidden MLP sees x only, so a kink in layer l of encoder e reaches layers 0..l of that
    encoder and nothing else.  MIMIC_MLPEncoder: the state enters the first layer, so a kink reaches every encoder and
    the init state.  MLPDecoder hidden layer l of decoder d: that decoder's layers 0..l, every encoder, the init state."""
    tied = set()
    enc_all = False
    for key, n in kinks.items():
        if not n:
            continue
        if key[0] == "enc":
            _, e, l = key
            enc = spec.encoders[e]
            if enc.kind == "mimic":
                enc_all = True
            else:
                for j in range(l + 1):
                    tied.update({f"encoders.{e}.layers.{j}.weight", f"encoders.{e}.layers.{j}.bias"})
        else:
            _, d, l = key
            names = spec.dec(d).names(d)
            for j in range(l + 1):
                tied.update({names[j] + ".weight", names[j] + ".bias"})
            enc_all = True
    if enc_all:
        tied.update(n for n in spec.param_names() if n.startswith("encoders.") or n.startswith("init_state."))
    return tied


def default_sequence(E: int) -> List[Tuple[int, int]]:
    """multimodn.py:515-516: (data_idx, enc_idx) = enumerate(range(E))."""
    return [(i, i) for i in range(E)]


def encoder_iterable(E: int, encoder_sequence: Optional[np.ndarray]) -> List[Tuple[int, int]]:
    """multimodn.py:509-531 without shuffle: all rows of the batch's sequence must agree."""
    if encoder_sequence is None:
        return default_sequence(E)
    seq = np.asarray(encoder_sequence)
    first = seq[0]
    if not (seq == first).all():
        raise ValueError("Encoder sequence has different values across the batch. "
                         "Hint: set batch size to 1 to avoid this error.")
    return [(k, int(e)) for k, e in enumerate(first)]


@dataclass
class StepResult:
    loss: float
    err_loss: np.ndarray            # [(E+1), D]  (multimodn.py:123,146,181)
    state_change: np.ndarray        # [E]         (multimodn.py:124,174)
    n_correct: np.ndarray           # [(E+1), D]  (multimodn.py:147,183)
    tp: np.ndarray
    tn: np.ndarray
    fp: np.ndarray
    fn: np.ndarray
    executed: np.ndarray            # [E] bool: encoder ran (no NaN skip) (multimodn.py:168-171)
    grads: Dict[str, Optional[np.ndarray]] = field(default_factory=dict)
    states: Dict[int, np.ndarray] = field(default_factory=dict)   # row -> [B,S] (row 0 = init)
    hidden: Dict[tuple, np.ndarray] = field(default_factory=dict)  # (encoder, layer) -> hidden activations [B, H] (keep_states)
    row_counts: Optional[np.ndarray] = None   # per-sample mode only: samples that own each grid row, [(E+1)]


def forward_backward(params: Dict[str, np.ndarray], spec: ModelSpec,
                     xs: Sequence[np.ndarray], y: np.ndarray,
                     encoder_sequence: Optional[np.ndarray] = None,
                     batch_global: Optional[int] = None,
                     present_override: Optional[Sequence[bool]] = None,
                     dtype=np.float32, want_grads: bool = True,
                     keep_states: bool = False,
                     drop_masks: Optional[Dict[int, np.ndarray]] = None,
                     kinks: Optional[Dict[tuple, int]] = None) -> StepResult:
    """One mini-batch of train_epoch's body (multimodn.py:119-203) with a hand-derived backward.

    xs[k]: [B, F_k] features of data slot k; y: [B, D] int targets in {0,1}.
    batch_global: divisor used for every batch mean (defaults to B).  A data-parallel shard
    passes the GLOBAL batch size so that summing shard results reproduces the full batch.
    present_override[k]: NaN-skip decision per data slot (the reference decides it on the whole
    batch, multimodn.py:168; a shard must use the global decision).
    Skipped encoders get grad None (autograd never touches them), everything else an array.
    drop_masks[e]: MIMIC_MLPEncoder e's dropout multipliers for this step, [B, F_e + S] with entries
    0 or 1/(1-p) (what nn.Dropout applies to cat([x, state]) in training mode, mlp_encoder.py:34,41);
    absent = no dropout (eval mode or p = 0).
    kinks: if a dict, it receives for every relu hidden layer the number of (sample, unit) pre-activations within
    KINK_MARGIN (relative to the layer's largest) of zero: keys ("enc", e, layer) / ("dec", d, layer).  The loss is not
    differentiable there, so two correct implementations may disagree on act'(.) for that sample (tied_parameters()).
    """
    S, E, D = spec.state_size, spec.E, spec.D
    dt = np.dtype(dtype)
    P = {k: np.asarray(v, dtype=dt) for k, v in params.items()}
    xs = [np.asarray(x, dtype=dt) for x in xs]
    y = np.asarray(y).astype(np.int64)
    B = y.shape[0]
    Bg = B if batch_global is None else int(batch_global)
    alpha = dt.type(spec.err_penalty)
    beta = dt.type(0.01 * spec.state_change_penalty)       # multimodn.py:86
    seq = encoder_iterable(E, encoder_sequence)
    enc_ids = [e for _, e in seq]
    if len(set(enc_ids)) != len(enc_ids):
        raise ValueError("repeated encoder ids in encoder_sequence are not supported")

    err_loss = np.zeros((E + 1, D), dt)
    state_change = np.zeros(E, dt)
    n_correct = np.zeros((E + 1, D), np.int64)
    tp = np.zeros((E + 1, D), np.int64); tn = np.zeros_like(tp)
    fp = np.zeros_like(tp); fn = np.zeros_like(tp)
    executed = np.zeros(E, bool)

    dec_names = [spec.dec(d).names(d) for d in range(D)]
    dz_rows: Dict[int, np.ndarray] = {}
    dec_tape: Dict[int, list] = {}      # row -> per decoder: activations [state, h_1, ..., h_last]

    def decode(row: int, s: np.ndarray):
        """decoders.py:19-20 (ClassDecoder) / :41-46 (MLPDecoder: hidden_activation(Linear) layers,
        then sigmoid(Linear)) + multimodn.py:144-157: argmax (ties -> 0), CrossEntropyLoss =
        mean_b(-log_softmax(o)[y]) over the sigmoid OUTPUTS, confusion matrix."""
        dz = np.zeros((B, D, 2), dt)
        acts_row = []
        for d in range(D):
            ds = spec.dec(d)
            hs = [s]
            for l, n in enumerate(dec_names[d][:-1]):
                pre = hs[-1] @ P[n + ".weight"].T + P[n + ".bias"]
                _note_kinks(kinks, ("dec", d, l), pre, ds.hidden_activation)
                hs.append(_act(pre, ds.hidden_activation))
            acts_row.append(hs)
            z = hs[-1] @ P[dec_names[d][-1] + ".weight"].T + P[dec_names[d][-1] + ".bias"]
            o = 1.0 / (1.0 + np.exp(-z))
            m = o.max(axis=1, keepdims=True)
            lse = m[:, 0] + np.log(np.exp(o - m).sum(axis=1))
            t = y[:, d]
            o_t = o[np.arange(B), t]
            err_loss[row, d] = (lse - o_t).sum(dtype=dt) / dt.type(Bg)
            pred = (o[:, 1] > o[:, 0]).astype(np.int64)     # torch.max: first index on ties
            n_correct[row, d] += int((pred == t).sum())
            tp[row, d] += int(((pred == 1) & (t == 1)).sum())
            tn[row, d] += int(((pred == 0) & (t == 0)).sum())
            fp[row, d] += int(((pred == 1) & (t == 0)).sum())   # cm[true=0][pred=1] (multimodn.py:57)
            fn[row, d] += int(((pred == 0) & (t == 1)).sum())   # cm[true=1][pred=0] (multimodn.py:58)
            if want_grads:
                p = np.exp(o - lse[:, None])
                g = p.copy()
                g[np.arange(B), t] -= 1
                dz[:, d, :] = g * o * (1 - o)
        dz_rows[row] = dz
        dec_tape[row] = acts_row

    state = np.tile(P["init_state.state_value"], (B, 1))      # state.py:29-32
    states = {0: state}
    decode(0, state)

    tape = []    # (enc_idx, s_in_row, hs) for executed encoders, in execution order
    prev_row = 0
    for k, e in seq:
        x = xs[k]
        if present_override is not None:
            present = bool(present_override[k])
        else:
            present = not np.isnan(x).any()                   # multimodn.py:168
        if not present:
            continue
        executed[e] = True
        enc = spec.encoders[e]
        L = len(enc.hidden)
        if enc.kind == "mimic":                               # mlp_encoder.py:40-47
            h = np.concatenate([x, state], axis=1)
            if drop_masks is not None and e in drop_masks:
                h = h * np.asarray(drop_masks[e], dt)         # nn.Dropout in training mode
            hs = [h]
            for l in range(L + 1):
                k = enc.layer_key(l)
                pre = h @ P[f"encoders.{e}.layers.{k}.weight"].T + P[f"encoders.{e}.layers.{k}.bias"]
                _note_kinks(kinks, ("enc", e, l), pre, enc.activation)
                h = _act(pre, enc.activation)
                hs.append(h)
            new_state = h
        else:
            hs = [x]
            h = x
            for l in range(L):                                    # mlp_encoder.py:75-76
                pre = h @ P[f"encoders.{e}.layers.{l}.weight"].T + P[f"encoders.{e}.layers.{l}.bias"]
                _note_kinks(kinks, ("enc", e, l), pre, enc.activation)
                h = _act(pre, enc.activation)
                hs.append(h)
            cat = np.concatenate([h, state], axis=1)              # mlp_encoder.py:78 (h first, state last)
            new_state = cat @ P[f"encoders.{e}.layers.{L}.weight"].T + P[f"encoders.{e}.layers.{L}.bias"]
        diff = new_state - state
        state_change[e] = (diff * diff).sum(dtype=dt) / dt.type(Bg * S)   # multimodn.py:174
        tape.append((e, prev_row, hs))
        state = new_state
        states[e + 1] = state
        prev_row = e + 1
        decode(e + 1, state)

    global_err = err_loss.sum(dtype=dt) / dt.type(D * (E + 1))            # multimodn.py:194
    global_sc = state_change.sum(dtype=dt) / dt.type(E)                   # multimodn.py:196
    loss = global_err * alpha + global_sc * beta                          # multimodn.py:199-202

    res = StepResult(float(loss), err_loss, state_change, n_correct, tp, tn, fp, fn, executed)
    if keep_states:
        res.states = states
        for e, _, hs in tape:
            for l, h in enumerate(hs[1:]):
                res.hidden[(e, l)] = h
    if not want_grads:
        return res

    grads: Dict[str, Optional[np.ndarray]] = {n: None for n in spec.param_names()}
    cL = alpha / dt.type(D * (E + 1) * Bg)
    cS = beta * dt.type(2.0) / dt.type(E * Bg * S)
    dec_grads: Dict[str, np.ndarray] = {}
    for d in range(D):
        for n in dec_names[d]:
            dec_grads[n + ".weight"] = np.zeros_like(P[n + ".weight"])
            dec_grads[n + ".bias"] = np.zeros_like(P[n + ".bias"])

    def decoder_back(row: int) -> np.ndarray:
        dz = dz_rows[row] * cL
        gs = np.zeros_like(states[row])
        for d in range(D):
            hs = dec_tape[row][d]
            names = dec_names[d]
            g = dz[:, d, :]                                    # grad wrt the last Linear's output
            for l in range(len(names) - 1, -1, -1):
                n = names[l]
                dec_grads[n + ".weight"] += g.T @ hs[l]
                dec_grads[n + ".bias"] += g.sum(axis=0)
                g = g @ P[n + ".weight"]
                if l > 0:
                    g = g * _act_grad_from_output(hs[l], spec.dec(d).hidden_activation)
            gs += g
        return gs

    G = np.zeros((B, S), dt)
    for e, in_row, hs in reversed(tape):
        enc = spec.encoders[e]
        L = len(enc.hidden)
        s_out, s_in = states[e + 1], states[in_row]
        diff = s_out - s_in
        G = G + decoder_back(e + 1) + cS * diff
        if enc.kind == "mimic":
            F_e = enc.n_features
            g = G
            for l in range(L, -1, -1):
                k = enc.layer_key(l)
                dpre = g * _act_grad_from_output(hs[l + 1], enc.activation)
                grads[f"encoders.{e}.layers.{k}.weight"] = dpre.T @ hs[l]
                grads[f"encoders.{e}.layers.{k}.bias"] = dpre.sum(axis=0)
                g = dpre @ P[f"encoders.{e}.layers.{k}.weight"]
            if drop_masks is not None and e in drop_masks:
                g = g * np.asarray(drop_masks[e], dt)
            G = g[:, F_e:] - cS * diff                         # no grad flows to x
            continue
        Wl = P[f"encoders.{e}.layers.{L}.weight"]
        HL = hs[-1].shape[1]
        cat = np.concatenate([hs[-1], s_in], axis=1)
        grads[f"encoders.{e}.layers.{L}.weight"] = G.T @ cat
        grads[f"encoders.{e}.layers.{L}.bias"] = G.sum(axis=0)
        dcat = G @ Wl
        dh = dcat[:, :HL]
        carry = dcat[:, HL:] - cS * diff
        for l in range(L - 1, -1, -1):
            dpre = dh * _act_grad_from_output(hs[l + 1], enc.activation)
            grads[f"encoders.{e}.layers.{l}.weight"] = dpre.T @ hs[l]
            grads[f"encoders.{e}.layers.{l}.bias"] = dpre.sum(axis=0)
            if l > 0:
                dh = dpre @ P[f"encoders.{e}.layers.{l}.weight"]
        G = carry
    G = G + decoder_back(0)
    grads["init_state.state_value"] = G.sum(axis=0, keepdims=True)        # grad of tile = sum_b
    grads.update(dec_grads)
    res.grads = grads
    return res


class Adam:
    """torch.optim.Adam (lr, betas=(0.9,0.999), eps=1e-8, no weight decay, no amsgrad), the
    optimiser every reference pipeline builds (titanic_mlp_pipeline.py:74).  Parameters whose
    grad is None are skipped entirely: no moment decay, no step increment."""

    def __init__(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state: Dict[str, dict] = {}

    def step(self, params: Dict[str, np.ndarray], grads: Dict[str, Optional[np.ndarray]]) -> None:
        b1, b2 = self.betas
        for name, g in grads.items():
            if g is None:
                continue
            p = params[name]
            dt = p.dtype
            st = self.state.setdefault(name, {"step": 0, "m": np.zeros_like(p), "v": np.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            g = g.astype(dt)
            st["m"] = (st["m"] * dt.type(b1) + g * dt.type(1 - b1)).astype(dt)
            st["v"] = (st["v"] * dt.type(b2) + (g * g) * dt.type(1 - b2)).astype(dt)
            bc1 = 1 - b1 ** t
            bc2_sqrt = (1 - b2 ** t) ** 0.5
            step_size = self.lr / bc1
            denom = (np.sqrt(st["v"]) / dt.type(bc2_sqrt) + dt.type(self.eps)).astype(dt)
            params[name] = (p - dt.type(step_size) * (st["m"] / denom)).astype(dt)


@dataclass
class EpochResult:
    """The six arrays train_epoch appends to MultiModNHistory (multimodn.py:244-250), with the
    reference's dtypes: loss/accuracy/state_change float64, sens/spec/bal-acc float32."""
    loss: np.ndarray
    accuracy: np.ndarray
    state_change: np.ndarray
    sensitivity: np.ndarray
    specificity: np.ndarray
    balanced_accuracy: np.ndarray
    step_losses: List[float]


def aggregate_epoch(E: int, D: int, step_results: Sequence[StepResult],
                    batch_sizes: Sequence[int]) -> EpochResult:
    """multimodn.py:104-115, 206-212, 222-242 given per-step results."""
    n_batches = len(step_results)
    n_samples = np.ones((E + 1, 1))                              # starts at ONE (multimodn.py:105)
    err = np.zeros((E + 1, D)); sc = np.zeros(E); ncor = np.zeros((E + 1, D))
    tp = np.zeros((E + 1, D), np.float32); tn = tp.copy(); fp = tp.copy(); fn = tp.copy()
    for r, bs in zip(step_results, batch_sizes):
        if r.row_counts is not None:                             # per-sample mode: rows += samples present
            n_samples[:, 0] += r.row_counts
            continue_counts = True
        else:
            continue_counts = False
            n_samples[0] += bs
        for e in range(E):
            if r.executed[e] and not continue_counts:
                n_samples[e + 1] += bs
        err += r.err_loss.astype(np.float32)                     # f32 step values summed in f64
        sc += r.state_change.astype(np.float32)
        ncor += r.n_correct
        tp += r.tp.astype(np.float32); tn += r.tn.astype(np.float32)
        fp += r.fp.astype(np.float32); fn += r.fn.astype(np.float32)
    err /= n_batches
    sc /= n_batches
    acc = ncor / n_samples
    with np.errstate(divide="ignore", invalid="ignore"):
        sd = tp + fn
        sens = np.where(sd == 0, np.float32(0), tp / sd).astype(np.float32)
        pd_ = tn + fp
        spec_ = np.where(pd_ == 0, np.float32(0), tn / pd_).astype(np.float32)
    bal = (sens + spec_) / 2
    return EpochResult(err, acc, sc, sens, spec_, bal, [r.loss for r in step_results])


def train_epoch(params: Dict[str, np.ndarray], spec: ModelSpec, batches, optimizer: Adam,
                dtype=np.float32, drop_masks: Optional[Sequence[Dict[int, np.ndarray]]] = None) -> EpochResult:
    """One epoch over `batches` = iterable of (xs, y) or (xs, y, encoder_sequence); updates
    `params` in place (dict entries replaced).  drop_masks[i]: batch i's dropout multipliers."""
    results, sizes = [], []
    for bi, batch in enumerate(batches):
        xs, y, seq = (list(batch) + [None])[:3]
        r = forward_backward(params, spec, xs, y, seq, dtype=dtype,
                             drop_masks=None if drop_masks is None else drop_masks[bi])
        optimizer.step(params, r.grads)
        results.append(r)
        sizes.append(np.asarray(y).shape[0])
    return aggregate_epoch(spec.E, spec.D, results, sizes)


# ------------------------------------------------------------------------------------------------
# forward-only entry points: test() / predict() / get_states() (multimodn.py:255-492)
# ------------------------------------------------------------------------------------------------
def decoder_outputs(params: Dict[str, np.ndarray], spec: ModelSpec, state: np.ndarray, dtype=np.float32) -> np.ndarray:
    """sigmoid(Linear(...)) of every decoder (decoders.py:19-20, :41-46): [B, D, 2]."""
    dt = np.dtype(dtype)
    s = np.asarray(state, dt)
    out = np.zeros((s.shape[0], spec.D, 2), dt)
    for d in range(spec.D):
        ds = spec.dec(d)
        names = ds.names(d)
        h = s
        for n in names[:-1]:
            h = _act(h @ np.asarray(params[n + ".weight"], dt).T + np.asarray(params[n + ".bias"], dt), ds.hidden_activation)
        z = h @ np.asarray(params[names[-1] + ".weight"], dt).T + np.asarray(params[names[-1] + ".bias"], dt)
        out[:, d, :] = 1.0 / (1.0 + np.exp(-z))
    return out


def test_epoch(params, spec: ModelSpec, batches, dtype=np.float32):
    """multimodn.py:255-419 without the optimiser: returns (EpochResult with the 'test' History
    arrays, outputs) where outputs[d] = (y_true [N], y_pred [N], y_prob [N]) for the decoder on the
    state after the LAST encoder (enc_idx == E-1, :354-357), probabilities renormalised to sum to 1
    (:415), prediction = argmax of the renormalised pair (first index wins ties).
    Batches whose last encoder was skipped (a NaN in its features) contribute no outputs (:354-357) while the
    reference keeps the targets of ALL batches (:283-286) and its report then raises on the two lengths (:418; recorded
    in tests/golden/titanic_missingness.npz as eval/test_report_raised).  For that case - undefined in the reference -
    outputs pairs the scores with the targets of the batches that produced them."""
    eval_spec = replace(spec, err_penalty=1.0, state_change_penalty=0.0)
    results, sizes = [], []
    outs, tgts = [], []
    for batch in batches:
        xs, y, seq = (list(batch) + [None])[:3]
        r = forward_backward(params, eval_spec, xs, y, seq, dtype=dtype, want_grads=False, keep_states=True)
        results.append(r)
        sizes.append(np.asarray(y).shape[0])
        if r.executed[spec.E - 1]:
            tgts.append(np.asarray(y))
            outs.append(decoder_outputs(params, spec, r.states[spec.E], dtype))
    ep = aggregate_epoch(spec.E, spec.D, results, sizes)
    outputs = []
    if outs:
        o = np.concatenate(outs, axis=0)
        t = np.concatenate(tgts, axis=0)
        for d in range(spec.D):
            p = o[:, d, :] / o[:, d, :].sum(axis=1, keepdims=True)
            pred = (p[:, 1] > p[:, 0]).astype(np.int64)
            outputs.append((t[:, d].astype(np.int64), pred, p[:, 1]))
    return ep, outputs


def predict(params, spec: ModelSpec, xs, encoder_sequence=None, dtype=np.float32, margins: Optional[list] = None) -> np.ndarray:
    """multimodn.py:422-458: argmax class of every decoder on every state, [(E+1), D, N] float64;
    rows of encoders outside the sequence stay 0; NO NaN skip on this path.
    margins: if a list, it receives |o_1 - o_0| of every prediction, [(E+1), D, N] (inf where the row does not exist): a
    prediction whose margin is of rounding size may legitimately differ between two correct implementations."""
    N = np.asarray(xs[0]).shape[0]
    dummy_y = np.zeros((N, spec.D), np.int64)
    r = forward_backward(params, spec, xs, dummy_y, encoder_sequence, dtype=dtype, want_grads=False,
                         keep_states=True, present_override=[True] * len(encoder_iterable(spec.E, encoder_sequence)))
    full = np.zeros((spec.E + 1, spec.D, N))
    marg = np.full((spec.E + 1, spec.D, N), np.inf)
    for row, st in r.states.items():
        o = decoder_outputs(params, spec, st, dtype)
        full[row] = (o[:, :, 1] > o[:, :, 0]).T.astype(np.float64)
        marg[row] = np.abs(o[:, :, 1] - o[:, :, 0]).T
    if margins is not None:
        margins.append(marg)
    return full


def get_states(params, spec: ModelSpec, batches, dtype=np.float32) -> np.ndarray:
    """multimodn.py:460-492: the state after the last EXECUTED encoder of every sample, [N, S]."""
    out = []
    for batch in batches:
        xs, y, seq = (list(batch) + [None])[:3]
        N = np.asarray(xs[0]).shape[0]
        r = forward_backward(params, spec, xs, np.zeros((N, spec.D), np.int64), seq, dtype=dtype,
                             want_grads=False, keep_states=True)
        last = 0
        for _, e in encoder_iterable(spec.E, seq):
            if r.executed[e]:
                last = e + 1
        out.append(r.states[last])
    return np.concatenate(out, axis=0)


def performance_metrics(y_true: np.ndarray, y_pred: np.ndarray, y_prob: np.ndarray) -> Dict[str, object]:
    """get_performance_metrics (multimodn.py:22-49) with the torchmetrics binary metrics restated
    (torchmetrics is a third-party dependency of the reference, unpinned in requirements-cpu.txt and
    absent from this image; algorithms as published for torchmetrics >= 1.0, thresholds=None):
      * f1       : F1Score(task="binary") on the PROBABILITIES -> thresholded at 0.5 (prob > 0.5),
                   2tp / (2tp + fp + fn), 0 when the denominator is 0
      * auc      : AUROC(task="binary") = trapezoidal area under the exact ROC curve below
      * accuracy : Accuracy(task="binary") on the predicted labels
      * sensitivity / specificity from ConfusionMatrix(y_pred, y_true) (0 when undefined, :38-45)
      * fpr, tpr, thr_roc : ROC(task="binary"): samples sorted by descending score, one point per
                   DISTINCT score (cumulative fp / tp), a leading (0, 0) point with threshold 1.0,
                   then divided by the totals
      * precision, recall, thr_pr : PrecisionRecallCurve(task="binary"): the same distinct-score
                   points, precision = tp/(tp+fp), recall = tp/tp_total, reversed so that recall
                   decreases, with a final (precision 1, recall 0) point; thresholds ascending."""
    y_true = np.asarray(y_true).astype(np.int64)
    y_pred = np.asarray(y_pred).astype(np.int64)
    y_prob = np.asarray(y_prob, np.float32)
    tp = int(((y_pred == 1) & (y_true == 1)).sum()); tn = int(((y_pred == 0) & (y_true == 0)).sum())
    fp = int(((y_pred == 1) & (y_true == 0)).sum()); fn = int(((y_pred == 0) & (y_true == 1)).sum())
    sens = tp / (tp + fn) if (tp + fn) != 0 else 0
    spec_ = tn / (tn + fp) if (tn + fp) != 0 else 0
    hard = (y_prob > 0.5).astype(np.int64)
    tp5 = int(((hard == 1) & (y_true == 1)).sum()); fp5 = int(((hard == 1) & (y_true == 0)).sum())
    fn5 = int(((hard == 0) & (y_true == 1)).sum())
    f1 = 2 * tp5 / (2 * tp5 + fp5 + fn5) if (2 * tp5 + fp5 + fn5) != 0 else 0.0
    order = np.argsort(-y_prob, kind="stable")
    ps, ts = y_prob[order], y_true[order]
    distinct = np.nonzero(np.diff(ps))[0]
    idx = np.concatenate([distinct, [len(ps) - 1]]) if len(ps) else np.array([], np.int64)
    tps = np.cumsum(ts)[idx].astype(np.float64)
    fps = (1 + idx - tps).astype(np.float64)
    thr = ps[idx]
    tps0 = np.concatenate([[0.0], tps]); fps0 = np.concatenate([[0.0], fps])
    with np.errstate(divide="ignore", invalid="ignore"):
        fpr = fps0 / fps0[-1] if fps0[-1] > 0 else np.zeros_like(fps0)
        tpr = tps0 / tps0[-1] if tps0[-1] > 0 else np.zeros_like(tps0)
        precision = tps / (tps + fps)
        recall = tps / tps[-1] if len(tps) and tps[-1] > 0 else np.zeros_like(tps)
    auc = float(np.trapezoid(tpr, fpr))
    return {"f1": f1, "auc": auc, "accuracy": (tp + tn) / max(len(y_true), 1), "sensitivity": sens,
            "specificity": spec_, "fpr": fpr, "tpr": tpr,
            "precision": np.concatenate([precision[::-1], [1.0]]), "recall": np.concatenate([recall[::-1], [0.0]]),
            "tn": tn, "fp": fp, "fn": fn, "tp": tp,
            "thr_roc": np.concatenate([[1.0], thr]), "thr_pr": thr[::-1].copy()}


PERFORMANCE_METRICS = ["f1", "auc", "accuracy", "sensitivity", "specificity", "fpr", "tpr", "precision", "recall",
                       "tn", "fp", "fn", "tp", "thr_roc", "thr_pr"]       # multimodn.py:18-19 (tuple order of :47-49)


def per_sample_step(params, spec: ModelSpec, xs, y, sequences, dtype=np.float32,
                    batch_global: Optional[int] = None, drop_masks: Optional[Dict[int, np.ndarray]] = None) -> StepResult:
    """Build-defined extension for per-sample missingness / per-sample encoder order (SURVEY 9.6,
    BASELINE config 5): the reference only defines this at batch size 1 (multimodn.py:518-523 raises
    otherwise, :168 skips per batch).  The batch result is the mean over samples of the reference's
    batch-size-1 result: cells sum_present CE / B, state change sum_present mean_j(ds^2) / B,
    counters over present samples only; grads are the mean of per-sample grads (None -> 0).
    batch_global: the divisor (defaults to B); a data-parallel shard passes the global sample count, so that the
    shards' results add up to the whole batch's.
    drop_masks[e]: MIMIC_MLPEncoder e's dropout multipliers, [B, F_e + S], row b = sample b's (as in forward_backward)."""
    B = np.asarray(y).shape[0]
    Bg = int(batch_global) if batch_global else B
    E, D = spec.E, spec.D
    acc: Optional[StepResult] = None
    gsum: Dict[str, np.ndarray] = {}
    counts = np.zeros(E + 1, np.int64)
    counts[0] = B
    for b in range(B):
        xb = [np.asarray(x)[b:b + 1] for x in xs]
        sb = None if sequences is None else np.asarray(sequences)[b:b + 1]
        mb = None if drop_masks is None else {e: np.asarray(m)[b:b + 1] for e, m in drop_masks.items()}
        r = forward_backward(params, spec, xb, np.asarray(y)[b:b + 1], sb, batch_global=Bg, dtype=dtype, drop_masks=mb)
        counts[1:] += r.executed.astype(np.int64)
        for n, g in r.grads.items():
            if g is not None:
                gsum[n] = g if n not in gsum else gsum[n] + g
        if acc is None:
            acc = r
        else:
            acc.err_loss = acc.err_loss + r.err_loss
            acc.state_change = acc.state_change + r.state_change
            acc.n_correct += r.n_correct; acc.tp += r.tp; acc.tn += r.tn; acc.fp += r.fp; acc.fn += r.fn
            acc.executed |= r.executed
            acc.loss += r.loss
    acc.grads = {n: gsum.get(n) for n in spec.param_names()}
    acc.row_counts = counts
    return acc


def per_sample_eval(params, spec: ModelSpec, xs, y, sequences, dtype=np.float32, margins: Optional[list] = None):
    """Forward-only counterpart of per_sample_step: every sample is the reference's batch-size-1
    forward (test() / predict() / get_states(), multimodn.py:255-492, with the NaN skip of :327).
    Returns (StepResult summed over samples with row_counts, predictions [(E+1), D, N] (0 where the row
    does not exist for the sample), final states [N, S], last-row outputs [(N_last, D, 2)], index of the
    samples whose LAST encoder ran).  margins: as in predict()."""
    B = np.asarray(y).shape[0]
    E, D = spec.E, spec.D
    eval_spec = replace(spec, err_penalty=1.0, state_change_penalty=0.0)
    acc = None
    counts = np.zeros(E + 1, np.int64); counts[0] = B
    preds = np.zeros((E + 1, D, B))
    marg = np.full((E + 1, D, B), np.inf)
    states = np.zeros((B, spec.state_size), np.dtype(dtype))
    last_out, last_idx = [], []
    for b in range(B):
        xb = [np.asarray(x)[b:b + 1] for x in xs]
        sb = None if sequences is None else np.asarray(sequences)[b:b + 1]
        r = forward_backward(params, eval_spec, xb, np.asarray(y)[b:b + 1], sb, batch_global=B, dtype=dtype,
                             want_grads=False, keep_states=True)
        counts[1:] += r.executed.astype(np.int64)
        last = 0
        for _, e in encoder_iterable(E, sb):
            if r.executed[e]:
                last = e + 1
        states[b] = r.states[last][0]
        for row, st in r.states.items():
            o = decoder_outputs(params, spec, st, dtype)
            preds[row, :, b] = (o[0, :, 1] > o[0, :, 0])
            marg[row, :, b] = np.abs(o[0, :, 1] - o[0, :, 0])
            if row == E:
                last_out.append(o[0]); last_idx.append(b)
        if acc is None:
            acc = r
        else:
            acc.err_loss = acc.err_loss + r.err_loss
            acc.n_correct += r.n_correct; acc.tp += r.tp; acc.tn += r.tn; acc.fp += r.fp; acc.fn += r.fn
            acc.executed |= r.executed
    acc.row_counts = counts
    if margins is not None:
        margins.append(marg)
    return acc, preds, states, (np.stack(last_out) if last_out else np.zeros((0, D, 2))), np.array(last_idx, np.int64)


def synthetic_batches(spec: ModelSpec, n_rows: int, batch_size: int, seed: int,
                      learnable: bool = True):
    """SURVEY 8d generator: standard-normal float32 features, binary int64 targets (either
    independent coin flips or y_d = 1[x.w_d + 0.5 eps > 0])."""
    rng = np.random.default_rng(seed)
    Fs = [e.n_features for e in spec.encoders]
    X = rng.standard_normal((n_rows, sum(Fs))).astype(np.float32)
    if learnable:
        w = rng.standard_normal((sum(Fs), spec.D)).astype(np.float32)
        y = ((X @ w + 0.5 * rng.standard_normal((n_rows, spec.D)).astype(np.float32)) > 0).astype(np.int64)
    else:
        y = rng.integers(0, 2, size=(n_rows, spec.D)).astype(np.int64)
    offs = np.cumsum([0] + Fs)
    batches = []
    for s in range(0, n_rows, batch_size):
        xs = [X[s:s + batch_size, offs[k]:offs[k + 1]].copy() for k in range(len(Fs))]
        batches.append((xs, y[s:s + batch_size].copy()))
    return batches
