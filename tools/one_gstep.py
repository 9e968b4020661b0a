#!/usr/bin/env python3
"""Run only the generator step (kinetic-gan.py:167-174) a few times, eagerly (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kinetic_gan_amd.wgan_gp import Trainer
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["ntu"]
G, D = bench.build_models(cfg, dev)
tr = Trainer(G, D)
real, labels, z, alpha = bench.synth_batch(cfg, 64, 0, dev)
which = os.environ.get("WHICH", "g")
for _ in range(int(os.environ.get("REPS", "5")) + 2):
    if which == "g":
        tr.g_step(labels, z, None)
    else:
        tr.d_step(real, labels, z, alpha, None)
torch.cuda.synchronize()
