"""Dataset surface of the reference (datasets/multimod_dataset.py:10-114): items are
(List[Tensor[F_k]], ndarray[D] [, ndarray[E]]) which torch's default collate turns into the
batch format train_epoch unpacks (multimodn.py:119)."""
from abc import ABC, abstractmethod
from itertools import accumulate
from collections.abc import Sequence
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

    def __getitems__(self, indices) -> "_Rows":
        """What torch's DataLoader calls for a whole batch of indices (torch >= 2.0; `Subset` forwards it): the rows are
        gathered ONCE per partition instead of one `Tensor(...)` per sample and partition, and torch's default collate
        recognises them (`_collate_rows`) and hands the gathered batch on as it is - the batch the per-sample path would
        have stacked, element for element and dtype for dtype (tests/test_host_logic.py).  Under the reference's loop a
        4096-row batch of four partitions took 79 ms of `__getitem__` + collate against 49 us for its training step
        (`bench.py` `stock_path`).  (Gathering whole rows of the undivided matrix with torch's index_select and cutting them into
        partitions afterwards is 3x faster on 8 cores and 70x SLOWER on the 128-thread host of an MI355X box - torch's
        intra-op pool on a 4 MB gather: measured, not kept.)  A custom `collate_fn` still sees a sequence of samples: every `_Row` unpacks to
        `(List[Tensor[F_k]], target row)` like `__getitem__`'s tuple."""
        # (a subclass that overrides __getitem__ - transforms, augmentation, a third element per sample - must see every
        #  sample: the reference has no __getitems__, its DataLoader always goes through __getitem__; likewise targets /
        #  partitions somebody replaced by another container: the samples themselves)
        if type(self).__getitem__ is not PartitionDataset.__getitem__ or not isinstance(self.y, np.ndarray) \
                or not all(isinstance(x, np.ndarray) for x in self.X):
            return [self[i] for i in indices]
        idx = np.asarray(indices, dtype=np.int64)
        src = _BatchRows()
        n = int(idx.shape[0])
        if n > 1 and int(idx[-1]) - int(idx[0]) == n - 1 and int(idx[0]) >= 0 and bool((np.diff(idx) == 1).all()):
            # consecutive rows (a loader without shuffling, a SequentialSampler over a Subset's range): a slice copies at memcpy
            # speed where the fancy-index gather walks an index array - 0.19 ms -> 0.03 ms per 4096 x 64 partition (the batch
            # still owns its memory: np.array copies, like Tensor(self.X[k][i]) in the reference's __getitem__)
            a, b = int(idx[0]), int(idx[-1]) + 1
            src.xs = [torch.from_numpy(np.array(self.X[k][a:b], order="C")).to(torch.float32) for k in range(self.n_partitions)]
            src.y_np = np.array(self.y[a:b])
        else:
            src.xs = [torch.from_numpy(np.ascontiguousarray(self.X[k][idx])).to(torch.float32) for k in range(self.n_partitions)]
            src.y_np = self.y[idx]
        src.y = torch.as_tensor(src.y_np)
        src.n = n
        return _Rows(src)


class _BatchRows:
    """The rows one `PartitionDataset.__getitems__` call gathered."""
    __slots__ = ("xs", "y", "y_np", "n")


class _Row:
    """Sample i of a `_BatchRows`: behaves like `__getitem__`'s `(List[Tensor[F_k]], target)` tuple where it is unpacked,
    indexed or iterated; costs two references until then."""
    __slots__ = ("src", "i")

    def __init__(self, src: _BatchRows, i: int):
        self.src, self.i = src, i

    def __len__(self) -> int:
        return 2

    def __getitem__(self, k: int):
        if k in (0, -2):
            return [x[self.i] for x in self.src.xs]
        if k in (1, -1):
            return self.src.y_np[self.i]
        raise IndexError(k)

    def __iter__(self):
        yield self[0]
        yield self[1]


class _Rows(Sequence):
    """The samples of one gathered batch, in order, made when somebody asks for one (a 4096-row batch would otherwise cost 4096
    objects that the default collate never looks at)."""
    __slots__ = ("src",)

    def __init__(self, src: _BatchRows):
        self.src = src

    def __len__(self) -> int:
        return self.src.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [_Row(self.src, k) for k in range(*i.indices(self.src.n))]
        if i < 0:
            i += self.src.n
        if not 0 <= i < self.src.n:
            raise IndexError(i)
        return _Row(self.src, i)


def _collate_rows(batch, *, collate_fn_map=None):
    """default_collate of `_Row` samples: the rows of ONE gathered batch in their order are that batch; anything else (rows of
    several fetches, a re-ordered list) is collated sample by sample like the tuples they stand for."""
    src = batch[0].src
    if isinstance(batch, _Rows) or (len(batch) == src.n and all(r.src is src and r.i == k for k, r in enumerate(batch))):
        return [list(src.xs), src.y]
    from torch.utils.data._utils.collate import default_collate
    return default_collate([(r[0], r[1]) for r in batch])


def _register_row_collate() -> None:
    try:
        from torch.utils.data._utils import collate as _c
        _c.default_collate_fn_map[_Row] = _collate_rows
    except (ImportError, AttributeError):                  # a torch without the registry: keep the per-sample path
        PartitionDataset.__getitems__ = None


_register_row_collate()


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


class DeviceResidentLoader:
    """Data feed for the HIP path: the whole partitioned dataset lives in
    HBM (288 GB per MI355X) and every batch is a set of device VIEWS in the batch format
    train_epoch unpacks (multimodn.py:119): `(List[[B, F_k] f32], [B, D] i64 [, [B, E] i64])`.
    No per-sample `Tensor(...)` construction, no collate, no host-to-device copy per step; with
    `MultiModN.nan_policy = "device"` a training epoch then runs without a single host sync.

    Drop-in where a pipeline builds `DataLoader(dataset, batch_size, shuffle=...)`
    (pipelines/titanic/titanic_mlp_pipeline.py:57-60): same iteration protocol and `len()`.
    `shuffle=True` draws a fresh device permutation per epoch and gathers each batch on the device.
    """

    def __init__(self, dataset, batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 device: Optional[torch.device] = None, generator: Optional[Generator] = None):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last, self.generator = shuffle, drop_last, generator
        data, targets, seq = self._materialise(dataset)
        self.data = [d.to(self.device, dtype=torch.float32).contiguous() for d in data]
        self.targets = targets.to(self.device, dtype=torch.int64).contiguous()
        if self.targets.dim() == 1:
            self.targets = self.targets.view(-1, 1)
        self.sequence = None if seq is None else seq.to(self.device, dtype=torch.int64).contiguous()
        self.n = int(self.targets.shape[0])
        self._views = None                                   # unshuffled: the batches are the same views every epoch

    @staticmethod
    def _materialise(dataset):
        """One pass over the dataset's items (or a ready (data, targets[, sequence]) tuple)."""
        if isinstance(dataset, (tuple, list)) and len(dataset) in (2, 3) and isinstance(dataset[0], (list, tuple)) \
                and all(isinstance(t, Tensor) for t in dataset[0]):
            seq = dataset[2] if len(dataset) == 3 else None
            return list(dataset[0]), torch.as_tensor(np.asarray(dataset[1]) if not isinstance(dataset[1], Tensor) else dataset[1]), \
                (None if seq is None else torch.as_tensor(np.asarray(seq) if not isinstance(seq, Tensor) else seq))
        if isinstance(dataset, PartitionDataset):                      # columns are already contiguous arrays
            return [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) for x in dataset.X], \
                torch.from_numpy(np.asarray(dataset.y)), None
        cols, ys, seqs = None, [], []
        for i in range(len(dataset)):
            item = dataset[i]
            d, y, sq = (list(item) + [None])[:3]
            if cols is None:
                cols = [[] for _ in d]
            for k, t in enumerate(d):
                cols[k].append(torch.as_tensor(t, dtype=torch.float32).reshape(1, -1))
            ys.append(np.asarray(y).reshape(1, -1))
            if sq is not None:
                seqs.append(np.asarray(sq).reshape(1, -1))
        if cols is None:
            raise ValueError("empty dataset")
        return [torch.cat(c, dim=0) for c in cols], torch.from_numpy(np.concatenate(ys, 0)), \
            (torch.from_numpy(np.concatenate(seqs, 0)) if seqs else None)

    @property
    def stable_batches(self) -> bool:
        """The same tuple objects every epoch (train_epoch may then cache what it derives from their addresses)."""
        return not self.shuffle

    def __len__(self) -> int:
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def __iter__(self):
        perm = None
        if not self.shuffle:
            # the SAME tuple objects every epoch: train_epoch recognises them and skips re-deriving what it derived from
            # their addresses last time (MultiModN._train_steps)
            if self._views is None:
                views = []
                for i in range(len(self)):
                    lo, hi = i * self.batch_size, min(self.n, (i + 1) * self.batch_size)
                    item = [[d[lo:hi] for d in self.data], self.targets[lo:hi]]
                    if self.sequence is not None:
                        item.append(self.sequence[lo:hi])
                    views.append(tuple(item))
                self._views = views
            yield from self._views
            return
        if self.shuffle:
            g = self.generator
            perm = (torch.randperm(self.n, generator=g) if g is not None else torch.randperm(self.n)).to(self.device)
        for i in range(len(self)):
            lo, hi = i * self.batch_size, min(self.n, (i + 1) * self.batch_size)
            if perm is None:
                item = [[d[lo:hi] for d in self.data], self.targets[lo:hi]]
                if self.sequence is not None:
                    item.append(self.sequence[lo:hi])
            else:
                idx = perm[lo:hi]
                item = [[d.index_select(0, idx) for d in self.data], self.targets.index_select(0, idx)]
                if self.sequence is not None:
                    item.append(self.sequence.index_select(0, idx))
            yield tuple(item)
