#!/usr/bin/env python3
"""Shapes of the tensors copied / cloned during one D step (eager, GPU box only)."""
import os, sys, collections
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
for _ in range(2): tr.d_step(real, labels, z, alpha, None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.d_step(real, labels, z, alpha, None)
torch.cuda.synchronize()
cnt = collections.Counter()
order = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::to"):
        cnt[(ev.name, str(ev.input_shapes)[:90])] += 1
for (n, sh), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d} {n:18s} {sh}")
