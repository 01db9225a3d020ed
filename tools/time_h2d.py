"""Host batches through the staging ring: one stream against copy stream + compute stream (round 6).

BASELINE configs[2] (4 x 64 features, state 128, 3 tasks, batch 4096), pinned and pageable HOST batches through
MultiModN._train_steps with multimodn_amd.optim.Adam; the same process times both layouts (the stager's switch).
Usage: python tools/time_h2d.py [--batch 4096] [--steps 32] [--reps 5]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch

import multimodn_amd as mm
from helpers import build_torch_model
from oracle import multimodn_oracle as O


class _Sized:
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    spec = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.0)
    model = build_torch_model(spec, O.init_params(spec, 0), "cuda", mm)
    opt = mm.optim.Adam(model.parameters(), lr=1e-3)
    batches = O.synthetic_batches(spec, a.steps * a.batch, a.batch, seed=3)
    res = {}
    for kind, thr in (("pinned", 8), ("pageable", 8), ("pageable", 4), ("pageable", 16), ("pageable", 32)):
        put = (lambda t: t.pin_memory()) if kind == "pinned" else (lambda t: t)
        hb = [([put(torch.from_numpy(x)) for x in xs], put(torch.from_numpy(y))) for xs, y in batches]
        model.stage_threads = thr                            # threads of torch's intra-op pool while a batch is packed
        kind = f"{kind}/{thr}thr"
        for flag in (False, True, False, True):
            model._train_steps(_Sized(hb), opt)              # (the stager exists after the first call)
            model._stager.use_copy_stream = flag
            model._train_steps(_Sized(hb), opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                model._train_steps(_Sized(hb), opt)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (a.reps * len(hb)) * 1e6
            res.setdefault((kind, flag), []).append(us)
    nbytes = sum(x.nbytes for x in batches[0][0]) + batches[0][1].nbytes
    for (kind, flag), v in res.items():
        best = min(v)
        print(f"{kind:15s} copy stream {int(flag)}: {best:7.1f} us/step ({[round(x, 1) for x in v]}), {nbytes / best / 1e3:5.1f} GB/s over the bus, "
              f"{a.batch / best:6.2f} M samples/s")


if __name__ == "__main__":
    main()
