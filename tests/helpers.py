"""Shared test helpers: golden-vector loading, oracle <-> torch model conversion."""
import json
import os

import numpy as np

from oracle import multimodn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_ID = {"relu": O.ACT_RELU, "sigmoid": O.ACT_SIGMOID, "identity": O.ACT_IDENTITY}
GOLDEN_NAMES = ["c1_titanic", "c1_curve20", "c2_split", "c3_small", "nan_skip", "seq_perm",
                "slp_sigmoid", "mlp_sigmoid", "mlp_identity"]
# MIMIC family: MIMIC_MLPEncoder + MLPDecoder runs of the reference, dropout masks recorded
MIMIC_GOLDEN_NAMES = ["mimic_p0", "mimic_drop", "mimic_mixed", "mimic_c3_small"]


def spec_from_cfg(c):
    kinds = c.get("enc_kinds", ["mlp"] * len(c["F"]))
    encs = [O.EncoderSpec(f, tuple(c["H"]), ACT_ID[c["act"]], kind=k,
                          dropout=c.get("dropout", 0.0) if k == "mimic" else 0.0) for f, k in zip(c["F"], kinds)]
    decs = [O.DecoderSpec(k, tuple(h)) for k, h in c["dec"]] if "dec" in c else None
    return O.ModelSpec(c["S"], encs, c["D"], c["pen"][0], c["pen"][1], decoders=decs)


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.cfg = json.loads(str(self.z["config_json"]))
        c = self.cfg
        self.spec = spec_from_cfg(c)
        self.n_batches = len([k for k in self.z.files if k.endswith("/y")])
        self.epochs = c["epochs"]

    def init_params(self):
        return {n: self.z["init/" + n].copy() for n in self.spec.param_names()}

    def final_params(self):
        return {n: self.z["final/" + n] for n in self.spec.param_names()}

    def batch(self, bi):
        n_slots = len([k for k in self.z.files if k.startswith(f"batch{bi}/x")])
        xs = [self.z[f"batch{bi}/x{k}"] for k in range(n_slots)]
        y = self.z[f"batch{bi}/y"]
        key = f"batch{bi}/seq"
        return (xs, y, self.z[key]) if key in self.z.files else (xs, y)

    def batches(self):
        return [self.batch(i) for i in range(self.n_batches)]

    def step_grads(self, s):
        pre = f"step{s}/grad/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}

    def step_masks(self, s):
        """Dropout multipliers the reference drew in step s: {encoder id: [B, F_e + S]} (MIMIC family)."""
        pre = f"step{s}/mask"
        return {int(k[len(pre):]): self.z[k] for k in self.z.files if k.startswith(pre)}

    def has_step(self, s):
        return f"step{s}/grad_none" in self.z.files


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def build_torch_model(spec, params, device, lib):
    """Instantiate multimodn_amd modules for an oracle ModelSpec and load `params`."""
    import torch
    import torch.nn.functional as F
    acts = {O.ACT_RELU: F.relu, O.ACT_SIGMOID: torch.sigmoid}
    acts[O.ACT_IDENTITY] = lib.encoders._identity
    encoders = []
    for e in spec.encoders:
        if e.kind == "mimic":
            enc = lib.MIMIC_MLPEncoder(spec.state_size, e.n_features, tuple(e.hidden), dropout=e.dropout,
                                       activation=acts[e.activation])
        else:
            enc = lib.MLPEncoder(spec.state_size, e.n_features, tuple(e.hidden), acts[e.activation])
        encoders.append(enc)
    decoders = [lib.LogisticDecoder(spec.state_size) if spec.dec(d).kind == "class" else
                lib.MLPDecoder(spec.state_size, tuple(spec.dec(d).hidden), 2, hidden_activation=acts[spec.dec(d).hidden_activation])
                for d in range(spec.D)]
    model = lib.MultiModN(spec.state_size, encoders, decoders, spec.err_penalty, spec.state_change_penalty,
                          device=torch.device(device))
    sd = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in params.items()}
    model.load_state_dict(sd)
    return model
