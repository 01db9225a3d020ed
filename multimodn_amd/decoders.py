"""Decoder plugins (multimodn/decoders/multimod_decoder.py:7-16, decoders.py:9-20,49-53)."""
from abc import ABC, abstractmethod
from typing import Callable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class MultiModDecoder(nn.Module, ABC):
    def __init__(self, state_size: int):
        super().__init__()
        self.state_size = state_size

    @abstractmethod
    def forward(self, state: Tensor) -> Tensor:
        ...


class ClassDecoder(MultiModDecoder):
    """activation(Linear(S -> n_classes)); attribute n_classes is read by train_epoch
    (multimodn.py:153).  state_dict keys fc.weight / fc.bias."""

    def __init__(self, state_size: int, n_classes: int, activation: Callable,
                 device: Optional[torch.device] = None):
        super().__init__(state_size)
        self.n_classes = n_classes
        self.fc = nn.Linear(state_size, n_classes, device=device)
        self.activation = activation

    def forward(self, state: Tensor) -> Tensor:
        return self.activation(self.fc(state))


class MLPDecoder(MultiModDecoder):
    """Multi-layer perceptron head (decoders.py:22-46): hidden_activation(Linear) layers, then
    output_activation(Linear(-> n_classes)).  state_dict keys layers.{l}.weight / .bias."""

    def __init__(self, state_size: int, hidden_layers: Tuple[int, ...], n_classes: int = 2,
                 output_activation: Callable = torch.sigmoid, hidden_activation: Callable = F.relu,
                 device: Optional[torch.device] = None):
        super().__init__(state_size)
        self.output_activation = output_activation
        self.hidden_activation = hidden_activation
        self.n_classes = n_classes
        widths = [state_size, *hidden_layers, n_classes]
        self.layers = nn.ModuleList(nn.Linear(i, o, device=device) for i, o in zip(widths, widths[1:]))

    def forward(self, x: Tensor) -> Tensor:
        for lin in list(self.layers)[:-1]:
            x = self.hidden_activation(lin(x))
        return self.output_activation(self.layers[-1](x))


class LogisticDecoder(ClassDecoder):
    """Two-class sigmoid head (decoders.py:49-53)."""

    def __init__(self, state_size: int, device: Optional[torch.device] = None):
        super().__init__(state_size, 2, torch.sigmoid, device)
