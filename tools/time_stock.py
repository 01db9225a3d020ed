"""The import-swap pipeline (bench.py `stock_path`): torch DataLoader(PartitionDataset) + torch.optim.Adam through
MultiModN.train_epoch at BASELINE configs[2], with and without the batch loop's (opt-in) loader prefetch thread (round 6).
Usage: python tools/time_stock.py [--batches 32] [--epochs 3]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch
from torch.utils.data import DataLoader

import multimodn_amd as mm
from helpers import build_torch_model
from oracle import multimodn_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    spec = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.0)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((a.batches * a.batch, 256)).astype(np.float32)
    y = rng.integers(0, 2, (a.batches * a.batch, 3)).astype(np.int64)
    ds = mm.PartitionDataset(X, y, [64] * 4)
    res = {}
    for shuffle in (False, True):
        loader = DataLoader(ds, batch_size=a.batch, shuffle=shuffle)
        model = build_torch_model(spec, O.init_params(spec, 0), "cuda", mm)
        opt = torch.optim.Adam(model.parameters(), 1e-3)
        hist = mm.MultiModNHistory(["a", "b", "c"])
        crit = torch.nn.CrossEntropyLoss()
        model.train_epoch(loader, opt, crit, hist)
        for rep in range(3):
            for flag in (False, True):
                model.prefetch_loader = flag
                model.train_epoch(loader, opt, crit, hist)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.epochs):
                    model.train_epoch(loader, opt, crit, hist)
                torch.cuda.synchronize()
                res.setdefault((shuffle, flag), []).append((time.perf_counter() - t0) / (a.epochs * a.batches) * 1e6)
        t0 = time.perf_counter()
        for _ in loader:
            pass
        res[(shuffle, "loader alone")] = [(time.perf_counter() - t0) / a.batches * 1e6]
    for k, v in res.items():
        print(f"shuffle {int(k[0])} prefetch {k[1]}: {min(v):8.1f} us/step  {[round(x) for x in v]}")


if __name__ == "__main__":
    main()
