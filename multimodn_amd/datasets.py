"""Dataset surface of the reference (datasets/multimod_dataset.py:10-114): items are
(List[Tensor[F_k]], ndarray[D] [, ndarray[E]]) which torch's default collate turns into the
batch format train_epoch unpacks (multimodn.py:119)."""
from abc import ABC, abstractmethod
from itertools import accumulate
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Generator, Tensor
from torch.utils.data import Dataset, Subset


class MultiModDataset(Dataset, ABC):
    @abstractmethod
    def __len__(self) -> int:
        ...

    def random_split(self, probabilities: Union[List[float], Tuple[float, ...]], seed: int,
                     balanced_target_idx: Optional[int] = None) -> List[Subset]:
        """Seeded split, optionally stratified on one target column (multimod_dataset.py:15-52)."""
        order = torch.randperm(len(self), generator=Generator().manual_seed(seed)).tolist()
        if balanced_target_idx is None:
            groups = {"all": order}
        else:
            groups = {}
            for idx in order:
                groups.setdefault(self[idx][1][balanced_target_idx], []).append(idx)
        total_p = sum(probabilities)
        parts: List[List[int]] = [[] for _ in probabilities]
        for members in groups.values():
            sizes = [int(len(members) * p / total_p) for p in probabilities]
            sizes[0] += len(members) - sum(sizes)
            for i, (end, size) in enumerate(zip(accumulate(sizes), sizes)):
                parts[i] = parts[i] + members[end - size:end]
        return [Subset(self, part) for part in parts]


class PartitionDataset(MultiModDataset):
    """Tabular X split column-wise into modalities (multimod_dataset.py:55-88)."""

    def __init__(self, X: np.ndarray, y: np.ndarray, partitions: Optional[List[int]] = None):
        self.partitions = [X.shape[1]] if partitions is None else partitions
        if sum(self.partitions) != X.shape[1]:
            raise ValueError("Paritions sum doesn't match data dimension. Expected: {}, got: {}"
                             .format(sum(self.partitions), X.shape[1]))
        self.n_partitions = len(self.partitions)
        self.X = np.split(X, list(accumulate(self.partitions[:-1])), axis=1)
        self.y = y

    def __len__(self) -> int:
        return len(self.y)

    def __getitem__(self, idx: int) -> Tuple[List[Tensor], np.ndarray]:
        return [Tensor(self.X[k][idx]) for k in range(self.n_partitions)], self.y[idx]


class FeatureWiseDataset(PartitionDataset):
    def __init__(self, X: np.ndarray, y: np.ndarray):
        super().__init__(X, y, [1] * X.shape[1])


class JointDatasets(MultiModDataset):
    def __init__(self, datasets: List[Dataset]):
        assert all(len(d) == len(datasets[0]) for d in datasets), "Datasets must have the same length"
        self.datasets = datasets

    def __len__(self) -> int:
        return len(self.datasets[0])

    def __getitem__(self, idx: int) -> Tuple[List[Tensor], np.ndarray]:
        return [torch.cat(d[idx][0]) for d in self.datasets], self.datasets[0][idx][1]
