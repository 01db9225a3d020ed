"""Initial-state plugins.  Same surface as the reference's multimodn/state.py:8-47."""
from abc import ABC, abstractmethod
from itertools import cycle
from typing import List, Optional

import torch
from torch import Tensor, nn


class InitState(nn.Module, ABC):
    """Base class: forward(batch_size) -> [batch_size, state_size]  (state.py:8-17)."""

    def __init__(self, state_size: int):
        super().__init__()
        self.state_size = state_size

    @abstractmethod
    def forward(self, batch_size) -> Tensor:
        ...


class TrainableInitState(InitState):
    """Learnable [1, S] vector broadcast over the batch (state.py:19-32).  state_dict key
    `state_value`.  The HIP chain never materialises the broadcast: kernels read the [S] vector."""

    def __init__(self, state_size: int, device: Optional[torch.device] = None):
        super().__init__(state_size)
        self.device = device
        self.state_value = nn.Parameter(torch.randn((1, state_size), device=device))

    def forward(self, batch_size) -> Tensor:
        return self.state_value.expand(batch_size, -1).clone()


class StaticInitState(InitState):
    """Non-trainable initial states cycled from a list (state.py:34-47).  Not on the HIP path."""

    def __init__(self, states: List[Tensor]):
        super().__init__(states[0].size(0))
        self.state_iterator = cycle(states)

    def forward(self, batch_size) -> Tensor:
        rows = [next(self.state_iterator).reshape(1, -1) for _ in range(batch_size)]
        return torch.cat(rows, dim=0).detach()
