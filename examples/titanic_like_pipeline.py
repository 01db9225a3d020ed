#!/usr/bin/env python3
"""The reference's Titanic MLP pipeline (pipelines/titanic/titanic_mlp_pipeline.py:19-103) with the
import swapped to multimodn_amd: same objects, same calls, same loop.  The Titanic CSV does not ship
here (no network), so the rows are synthetic and Titanic-shaped: 712 passengers, 6 standardised
features, 1 binary target correlated with them.

    python examples/titanic_like_pipeline.py [--epochs 30] [--state-size 32] [--device-loader]
                                             [--featurewise | --missingness [--per-sample-batches]]

--featurewise: the body of titanic_featurewise_pipeline.py:26-73 instead - a FeatureWiseDataset, one
MLPFeatureEncoder(state 5, hidden 5) per feature, batch 32.  --missingness: titanic_missingness_pipeline.py:26-74 -
the same with missing values kept as NaN (most often in the last feature, like the Titanic's cabin number) at batch
size 1, so that a passenger's missing feature skips that feature's encoder (multimodn.py:167-171).
--per-sample-batches (with --missingness): the same model and data in batches of 32 with `model.per_sample = True` - every
passenger still skips exactly their own missing features (the reference can only do that one passenger per step), in 1 / 32
of the steps.

--device-loader swaps torch's DataLoader for multimodn_amd.DeviceResidentLoader (dataset in HBM,
no per-sample tensor construction, no H2D copy per step); everything else is unchanged.
"""
import argparse
import os
import pickle as pkl
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn import CrossEntropyLoss
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodn_amd import (DeviceResidentLoader, FeatureWiseDataset, LogisticDecoder, MLPEncoder,      # noqa: E402
                           MLPFeatureEncoder, MultiModN, MultiModNHistory, PartitionDataset)


def titanic_like(n=712, seed=0, featurewise=False, missing=False):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, 6)).astype(np.float32)                       # 'Fare', 'Pclass', 'Age', 'Sex_male', ...
    logit = X @ np.array([0.8, -0.9, -0.4, -1.6, 0.2, 0.1], np.float32)
    y = (logit + 0.8 * rng.standard_normal(n) > 0).astype(np.int64).reshape(-1, 1)
    if missing:                                                              # the age of one passenger in five, most cabin numbers
        X[rng.random(n) < 0.2, 2] = np.nan
        X[rng.random(n) < 0.4, 5] = np.nan
    return FeatureWiseDataset(X, y) if featurewise else PartitionDataset(X, y)   # one partition = one modality


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--state-size", type=int, default=32)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--device-loader", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--featurewise", action="store_true")
    ap.add_argument("--missingness", action="store_true")
    ap.add_argument("--per-sample-batches", action="store_true")
    args = ap.parse_args(argv)
    featurewise = args.featurewise or args.missingness
    if featurewise:
        args.state_size = 5                                  # (both pipelines: state_size = 5)
    if args.missingness:
        args.batch_size = 32 if args.per_sample_batches else 1

    torch.manual_seed(args.seed)
    targets = ['Survived']
    dataset = titanic_like(n=712 if not args.missingness else 200, seed=args.seed, featurewise=featurewise, missing=args.missingness)
    train_data, val_data, _ = dataset.random_split((0.8, 0.2, 0), args.seed, 0)
    if args.device_loader:
        def rows(subset):
            idx = np.asarray(subset.indices)
            return PartitionDataset(np.concatenate(dataset.X, axis=1)[idx], dataset.y[idx], dataset.partitions)
        train_loader = DeviceResidentLoader(rows(train_data), args.batch_size)
        val_loader = DeviceResidentLoader(rows(val_data), args.batch_size)
    else:
        train_loader = DataLoader(train_data, args.batch_size)
        val_loader = DataLoader(val_data, args.batch_size)

    if featurewise:
        encoders = [MLPFeatureEncoder(args.state_size, 5, F.relu) for _ in range(6)]
    else:
        encoders = [MLPEncoder(args.state_size, 6, (5, 5), F.relu)]
    decoders = [LogisticDecoder(args.state_size) for _ in targets]
    model = MultiModN(args.state_size, encoders, decoders, 0.7, 0.3)
    model.per_sample = bool(args.missingness and args.per_sample_batches)
    optimizer = torch.optim.Adam(list(model.parameters()), 0.01)
    criterion = CrossEntropyLoss()
    history = MultiModNHistory(targets)

    for _ in range(args.epochs):
        model.train_epoch(train_loader, optimizer, criterion, history)
        results = model.test(val_loader, criterion, history, tag='val')

    # the reference pickles the whole model and the history (:96,102) and plots from the history
    with tempfile.TemporaryDirectory() as d:
        pkl.dump(model, open(os.path.join(d, "model.pkl"), "wb"))
        pkl.dump(history, open(os.path.join(d, "history.pkl"), "wb"))
        model2 = pkl.load(open(os.path.join(d, "model.pkl"), "rb"))
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
    f1, auc, acc = (float(v) for v in results[0][:3])
    if not args.quiet:
        print(f"train loss (last state): {history.loss['train'][0][-1, 0]:.4f} -> {history.loss['train'][-1][-1, 0]:.4f}")
        print(f"val   loss (last state): {history.loss['val'][0][-1, 0]:.4f} -> {history.loss['val'][-1][-1, 0]:.4f}")
        print(f"val report: f1 {f1:.3f}  auc {auc:.3f}  accuracy {acc:.3f}" +
              (f"  (the {int(sum(int(results[0][k]) for k in (9, 10, 11, 12)))} passengers whose last feature is known)" if args.missingness else ""))
    return history, results


if __name__ == "__main__":
    main()
