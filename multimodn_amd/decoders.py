"""Decoder plugins (multimodn/decoders/multimod_decoder.py:7-16, decoders.py:9-20,49-53)."""
from abc import ABC, abstractmethod
from typing import Callable, Optional

import torch
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


class LogisticDecoder(ClassDecoder):
    """Two-class sigmoid head (decoders.py:49-53)."""

    def __init__(self, state_size: int, device: Optional[torch.device] = None):
        super().__init__(state_size, 2, torch.sigmoid, device)
