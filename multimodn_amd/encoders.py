"""Encoder plugins with the reference's constructor signatures and state_dict keys
(multimodn/encoders/multimod_encoder.py:8-17, mlp_encoder.py:49-80, slp_encoders.py:5-34).

`forward` keeps the module-level contract `encoder(state, x) -> new_state` for users who call a
plugin directly; `MultiModN.train_epoch` does not call it: it hands the parameters to the HIP
chain kernels (multimodn_amd/engine.py)."""
from abc import ABC, abstractmethod
from typing import Callable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class MultiModEncoder(nn.Module, ABC):
    def __init__(self, state_size: int):
        super().__init__()
        self.state_size = state_size

    @abstractmethod
    def forward(self, state: Tensor, x: Tensor) -> Tensor:
        ...


def _identity(x):
    return x


class MLPEncoder(MultiModEncoder):
    """Hidden Linear+activation layers see x only; the state joins the LAST Linear as
    cat([h, state]) and the output has no activation (mlp_encoder.py:64-80)."""

    def __init__(self, state_size: int, n_features: int, hidden_layers: Tuple[int, ...],
                 activation: Callable = F.relu, device: Optional[torch.device] = None):
        super().__init__(state_size)
        self.activation = activation
        self.n_features = n_features
        self.hidden_layers = tuple(hidden_layers)
        widths = [n_features, *self.hidden_layers]
        self.layers = nn.ModuleList()
        for fan_in, fan_out in zip(widths, widths[1:]):
            self.layers.append(nn.Linear(fan_in, fan_out, device=device))
        self.layers.append(nn.Linear(widths[-1] + state_size, state_size, device=device))

    def forward(self, state: Tensor, x: Tensor) -> Tensor:
        h = x
        for lin in list(self.layers)[:-1]:
            h = self.activation(lin(h))
        return self.layers[-1](torch.cat([h, state], dim=1))


class MLPFeatureEncoder(MLPEncoder):
    """One feature through one hidden layer: MLPEncoder(state_size, 1, (hidden_size,)) (mlp_encoder.py:81-94; the
    encoders of pipelines/titanic/titanic_featurewise_pipeline.py:70 and titanic_missingness_pipeline.py:71 over a
    FeatureWiseDataset).  The reference's forward re-wraps x with Tensor(x), which for the [B, 1] float32 batches it is fed
    is the same tensor; here x becomes float32 on the state's device.  On the HIP path it is an MLPEncoder like any other."""

    def __init__(self, state_size: int, hidden_size: int, activation: Callable = F.relu,
                 device: Optional[torch.device] = None):
        super().__init__(state_size, 1, (hidden_size,), activation, device)

    def forward(self, state: Tensor, x) -> Tensor:
        return super().forward(state, torch.as_tensor(x, dtype=torch.float32, device=state.device))


class MIMIC_MLPEncoder(MultiModEncoder):
    """The MIMIC pipelines' encoder (mlp_encoder.py:9-47): Dropout(p) on cat([x, state]) feeds the
    FIRST Linear, the activation follows EVERY Linear including the last one, whose output is the
    new state.  `layers[0]` is the nn.Dropout, so the Linears are layers.1 .. layers.n exactly as in
    the reference's state_dict."""

    def __init__(self, state_size: int, n_features: int, hidden_layers: Tuple[int, ...], dropout: float = .2,
                 activation: Callable = F.relu, device: Optional[torch.device] = None):
        super().__init__(state_size)
        self.activation = activation
        self.dropout = dropout
        self.n_features = n_features
        self.hidden_layers = tuple(hidden_layers)
        widths = [n_features + state_size, *self.hidden_layers, state_size]
        self.layers = nn.ModuleList([nn.Dropout(dropout)])
        for fan_in, fan_out in zip(widths, widths[1:]):
            self.layers.append(nn.Linear(fan_in, fan_out, device=device))

    @property
    def linears(self):
        return list(self.layers)[1:]

    def forward(self, state: Tensor, x: Tensor) -> Tensor:
        h = self.layers[0](torch.cat([x, state], dim=1))
        for lin in self.linears:
            h = self.activation(lin(h))
        return h


class SLPEncoder(MLPEncoder):
    """hidden_layers=() : a single Linear on cat([x, state]); `activation` is never applied
    (slp_encoders.py:5-14)."""

    def __init__(self, state_size: int, n_features: int, activation: Callable = torch.sigmoid):
        super().__init__(state_size, n_features, (), activation)


class LinearEncoder(SLPEncoder):
    def __init__(self, state_size: int, n_features: int):
        super().__init__(state_size, n_features, _identity)


class LogisticEncoder(SLPEncoder):
    def __init__(self, state_size: int, n_features: int):
        super().__init__(state_size, n_features, torch.sigmoid)
