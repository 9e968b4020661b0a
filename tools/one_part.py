#!/usr/bin/env python3
"""Replay ONE hipGraph-captured part of the iteration a few times (for rocprofv3 timelines).  PART=d_step|g_step|iteration"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kinetic_gan_amd.wgan_gp import Trainer
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["ntu"]
G, D = bench.build_models(cfg, dev)
tr = Trainer(G, D)
real, labels, z, alpha = bench.synth_batch(cfg, 64, 0, dev)
part = os.environ.get("PART", "g_step")
fn = {"d_step": lambda: tr.d_step(real, labels, z, alpha, None), "g_step": lambda: tr.g_step(labels, z, None),
      "iteration": lambda: tr.iteration(real, labels, z, alpha, None, None, with_g=True)}[part]
rep = bench._capture(fn)
for _ in range(4):
    rep()
torch.cuda.synchronize()
