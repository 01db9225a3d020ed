#!/usr/bin/env python3
"""The reference's MIMIC multi-task pipeline body (pipelines/mimic/mimic_multi_task_pipeline.py:113-190) with the
import swapped to multimodn_amd: MIMIC_MLPEncoder per modality, MLPDecoder per target, train_epoch + test(val) every
epoch (last_epoch=True on the final one), the best checkpoint by cumulative validation AUROC + balanced accuracy
written with torch.save and loaded back for the test split, the whole model and the history pickled, the history
plotted and printed.  MIMIC itself does not ship here (credentialed data, no network): the rows are synthetic and
MIMIC-shaped (tabular modalities of different widths, several binary targets).

    python examples/mimic_like_pipeline.py [--epochs 8] [--state-size 50] [--rows 4096]
"""
import argparse
import os
import pickle as pkl
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F
from torch import sigmoid
from torch.nn import CrossEntropyLoss
from torch.utils.data import DataLoader, Subset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodn_amd import (MIMIC_MLPEncoder, MLPDecoder, MultiModN, MultiModNHistory, PartitionDataset)   # noqa: E402


def mimic_like(n, partitions, n_targets, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, sum(partitions))).astype(np.float32)
    w = rng.standard_normal((sum(partitions), n_targets)).astype(np.float32) / np.sqrt(sum(partitions))
    y = ((X @ w + 0.5 * rng.standard_normal((n, n_targets)).astype(np.float32)) > 0).astype(np.int64)
    return PartitionDataset(X, y, list(partitions))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--state-size", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=16)            # mimic_multi_task_pipeline.py: batch_size_train
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)

    # hyper-parameters as in the reference pipeline (:60-75)
    state_size, learning_rate, epochs = args.state_size, 0.001, args.epochs
    err_penalty, state_change_penalty = 1, 0
    encoder_hidd_units = decoder_hidd_units = 32
    dropout = 0.2
    batch_size_train, batch_size_val = args.batch_size, 2 * args.batch_size
    targets = ['Cardiomegaly', 'Enlarged Cardiomediastinum', 'Edema']
    partitions = [6, 11, 22, 19]                                     # demographics / chart / lab / procedure events-like

    seed = args.seed
    torch.manual_seed(seed)
    dataset_modn = mimic_like(args.rows, partitions, len(targets), seed)
    idx = np.random.default_rng(seed).permutation(args.rows)
    train_ind, val_ind, test_ind = idx[:int(.8 * args.rows)], idx[int(.8 * args.rows):int(.9 * args.rows)], idx[int(.9 * args.rows):]
    train_data, val_data = Subset(dataset_modn, train_ind), Subset(dataset_modn, val_ind)
    train_loader = DataLoader(train_data, batch_size_train)
    val_loader = DataLoader(val_data, batch_size_val)

    # ModN model specification (:118-120)
    encoders = [MIMIC_MLPEncoder(state_size, partition, (encoder_hidd_units, encoder_hidd_units), activation=F.relu, dropout=dropout)
                for partition in partitions]
    decoders = [MLPDecoder(state_size, (decoder_hidd_units, decoder_hidd_units), 2, output_activation=sigmoid) for _ in targets]
    model_modn = MultiModN(state_size, encoders, decoders, err_penalty, state_change_penalty)
    optimizer = torch.optim.Adam(list(model_modn.parameters()), learning_rate)
    criterion = CrossEntropyLoss()
    history = MultiModNHistory(targets)

    with tempfile.TemporaryDirectory() as directory:
        model_path_modn = os.path.join(directory, 'modn_model.pkl')
        best_model_path_modn = os.path.join(directory, 'modn_best_model.pt')
        # ModN training (:133-153)
        best_auc_bac_sum = 0
        train_buff_modn = None
        for epoch in range(epochs):
            if epoch == epochs - 1:
                train_buff_modn = model_modn.train_epoch(train_loader, optimizer, criterion, history, last_epoch=True)
            else:
                model_modn.train_epoch(train_loader, optimizer, criterion, history)
            val_buff_modn = model_modn.test(val_loader, criterion, history, tag='val')
            auc_val = bac_val = 0
            for val_buff_item in val_buff_modn:
                auc_val += val_buff_item[1]
                bac_val += (val_buff_item[3] + val_buff_item[4]) / 2
            auc_bac_sum = auc_val + bac_val
            if auc_bac_sum > best_auc_bac_sum:                       # checkpoint with the best validation auroc + bac
                torch.save({'epoch': epoch + 1, 'model_state_dict': model_modn.state_dict(),
                            'auc_bac_val_cum': auc_bac_sum}, best_model_path_modn)
                best_auc_bac_sum = auc_bac_sum
        pkl.dump(model_modn, open(model_path_modn, 'wb'))
        pkl.dump(history, open(os.path.join(directory, 'history.pkl'), 'wb'))
        history.plot(os.path.join(directory, 'plot.png'), targets, show_state_change=False)
        if not args.quiet:
            history.print_results()
        # ModN testing on the best checkpoint (:173-180)
        test_loader = DataLoader(Subset(dataset_modn, test_ind), batch_size_val)
        checkpoint = torch.load(best_model_path_modn)
        model_modn.load_state_dict(checkpoint['model_state_dict'])
        test_modn_best = model_modn.test(test_loader, criterion)
        rows = [[target] + list(map(lambda metric: np.asarray(metric), test_modn_best[t])) for t, target in enumerate(targets)]
    if not args.quiet:
        for r in rows:
            print(f"{r[0]:28s} f1 {float(r[1]):.3f}  auc {float(r[2]):.3f}  accuracy {float(r[3]):.3f}")
    return history, train_buff_modn, rows, checkpoint['epoch']


if __name__ == "__main__":
    main()
