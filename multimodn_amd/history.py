"""MultiModNHistory: passive per-epoch store with the reference's field names, shapes and dtypes
(multimodn/history.py:9-32) plus its result-table helpers (:98-161)."""
from typing import Dict, List

import numpy as np

_METRICS = ("loss", "accuracy", "sensitivity", "specificity", "balanced_accuracy")


def display_title(key: str) -> str:
    return key.replace("_", " ").capitalize()


class MultiModNHistory:
    def __init__(self, targets: List[str]):
        self.decoder_names: List[str] = targets
        self.state_change_loss: List[np.ndarray] = []
        self.loss: Dict[str, List[np.ndarray]] = {"train": []}
        self.accuracy: Dict[str, List[np.ndarray]] = {"train": []}
        self.sensitivity: Dict[str, List[np.ndarray]] = {"train": []}
        self.specificity: Dict[str, List[np.ndarray]] = {"train": []}
        self.balanced_accuracy: Dict[str, List[np.ndarray]] = {"train": []}

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
