import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import multimodn_amd as mm
from helpers import build_torch_model
from oracle import multimodn_oracle as O
spec = O.ModelSpec(128, [O.EncoderSpec(64, (32, 32), O.ACT_RELU) for _ in range(4)], 3, 1.0, 0.3)
B, NB = 4096, 16
host = O.synthetic_batches(spec, B * NB, B, seed=1)
loader = [([torch.from_numpy(x) for x in xs], torch.from_numpy(y)) for xs, y in host]
model = build_torch_model(spec, O.init_params(spec, 0), "cuda", mm)
opt = mm.optim.Adam(list(model.parameters()), 1e-3)
crit = torch.nn.CrossEntropyLoss()
model.train_epoch(loader[:2], opt, crit)
pr = cProfile.Profile(); pr.enable()
model.train_epoch(loader, opt, crit)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
