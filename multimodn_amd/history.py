"""MultiModNHistory: passive per-epoch store with the reference's field names, shapes and dtypes
(multimodn/history.py:9-32) plus its result-table helpers (:98-161)."""
from typing import Dict, List

import numpy as np

_METRICS = ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy")


class PendingEpoch:
    """The six History arrays of one training epoch whose device-side sums are still on their way to the host
    (MultiModN.train_epoch enqueues one asynchronous copy into pinned memory behind the epoch's last launch and returns;
    a second epoch can be submitted while the first still runs).  resolve() waits for the copy, forms the arrays
    (multimodn.py:222-242) and hands them to every list that holds a placeholder of this epoch."""

    def __init__(self, wait, compute):
        self._wait, self._compute = wait, compute
        self.arrays = None
        self.slots = []                                     # (list, index, key)

    def resolve(self):
        if self.arrays is None:
            self._wait()
            self.arrays = self._compute()
            self._wait = self._compute = None
            for lst, idx, key in self.slots:
                list.__setitem__(lst, idx, self.arrays[key])
            self.slots = []
        return self.arrays


class _Pending:
    """Placeholder of one array of a PendingEpoch; anything that looks at it gets the array."""
    __slots__ = ("epoch", "key")

    def __init__(self, epoch, key):
        self.epoch, self.key = epoch, key

    def value(self):
        return self.epoch.resolve()[self.key]

    def __array__(self, dtype=None, copy=None):
        v = self.value()
        return v if dtype is None else v.astype(dtype)

    def __getitem__(self, i):
        return self.value()[i]

    def __len__(self):
        return len(self.value())

    def __iter__(self):
        return iter(self.value())

    def __getattr__(self, name):
        return getattr(self.value(), name)

    def __repr__(self):
        return repr(self.value())


class HistoryList(list):
    """A list of per-epoch numpy arrays, as the reference's History holds them; entries appended by a training epoch
    may still be in flight and are resolved - every pending epoch, oldest first - the first time anything reads the
    list (indexing, iteration, pickling, comparison).  Reading is the only synchronisation point of train_epoch."""

    def append_pending(self, epoch: PendingEpoch, key: str) -> None:
        list.append(self, _Pending(epoch, key))
        epoch.slots.append((self, len(self) - 1, key))

    def _settle(self):
        for i in range(list.__len__(self)):
            v = list.__getitem__(self, i)
            if isinstance(v, _Pending):
                v.epoch.resolve()

    def __getitem__(self, i):
        self._settle()
        return list.__getitem__(self, i)

    def __iter__(self):
        self._settle()
        return list.__iter__(self)

    def __reversed__(self):
        self._settle()
        return list.__reversed__(self)

    def __eq__(self, other):
        self._settle()
        return list.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._settle()
        return list.__repr__(self)

    def __reduce_ex__(self, protocol):                      # pickles (and deep-copies) as a plain list of arrays
        self._settle()
        return (list, (list(list.__iter__(self)),))

    def copy(self):
        self._settle()
        return list(list.__iter__(self))

    def __add__(self, other):
        self._settle()
        return list(list.__iter__(self)) + list(other)

    # Everything that moves, removes or looks for entries settles first: a pending epoch writes its arrays back by the
    # position its placeholders were appended at.


def _settled(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        self._settle()
        return base(self, *a, **k)
    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


for _name in ("pop", "index", "count", "__contains__", "sort", "reverse", "insert", "remove", "__delitem__", "__setitem__",
              "__iadd__", "__mul__", "__rmul__", "__imul__", "clear", "__lt__", "__le__", "__gt__", "__ge__", "__ne__"):
    setattr(HistoryList, _name, _settled(_name))


def _extend(self, other):
    self._settle()
    list.extend(self, list(other))


HistoryList.extend = _extend


def display_title(key: str) -> str:
    return key.replace("_", " ").capitalize()


class MultiModNHistory:
    def __init__(self, targets: List[str]):
        self.decoder_names: List[str] = targets
        # (HistoryList: a list whose entries may still be on their way from the device - see PendingEpoch)
        self.state_change_loss: List[np.ndarray] = HistoryList()
        self.loss: Dict[str, List[np.ndarray]] = {"train": HistoryList()}
        self.accuracy: Dict[str, List[np.ndarray]] = {"train": HistoryList()}
        self.sensitivity: Dict[str, List[np.ndarray]] = {"train": HistoryList()}
        self.specificity: Dict[str, List[np.ndarray]] = {"train": HistoryList()}
        self.balanced_accuracy: Dict[str, List[np.ndarray]] = {"train": HistoryList()}

    def wait(self) -> None:
        """Resolve every entry that is still in flight (reading any entry does the same)."""
        for lst in [self.state_change_loss] + [v for m in _METRICS for v in getattr(self, m).values()]:
            if isinstance(lst, HistoryList):
                lst._settle()

    def get_results(self):
        """Last epoch, last encoder row, one line per decoder (history.py:98-150)."""
        import pandas as pd
        cols = ["State change loss"]
        table = [[self.state_change_loss[-1][-1]] for _ in self.decoder_names]
        for metric in _METRICS:
            label = metric.replace("_", " ")
            for tag, values in getattr(self, metric).items():
                cols.append(f"{display_title(tag)} {label}")
                for i in range(len(self.decoder_names)):
                    table[i].append(values[-1][-1][i])
        df = pd.DataFrame(np.asarray(table, dtype=float), columns=cols)
        df.index = self.decoder_names
        return df

    def print_results(self):
        print(self.get_results())

    def save_results(self, path):
        self.get_results().to_csv(path, index_label="Target")

    def plot(self, filepath: str, targets_to_display: List[str], show_state_change: bool = False):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        tags = list(self.loss.keys())
        fig, ax = plt.subplots(figsize=(10 * len(tags), 5 * len(_METRICS)), nrows=len(_METRICS),
                               ncols=len(tags), squeeze=False)
        for name in targets_to_display:
            if name not in self.decoder_names:
                raise ValueError(f"Target name '{name}' is not part of the MultiModN history")
            i = self.decoder_names.index(name)
            for r, metric in enumerate(_METRICS):
                for c, tag in enumerate(tags):
                    series = [v[-1][i] for v in getattr(self, metric).get(tag, [])]
                    ax[r][c].plot(series, label=name)
                    ax[r][c].set_title(f"{tag.capitalize()} {display_title(metric)}")
                    ax[r][c].legend(loc="best")
                    ax[r][c].grid(True)
        if show_state_change:
            ax[0][0].plot([v[-1] for v in self.state_change_loss], label="State change loss")
        plt.tight_layout()
        fig.savefig(filepath)
        plt.close(fig)
