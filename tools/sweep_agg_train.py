#!/usr/bin/env python3
"""Persistent-grid size of the matrix-core aggregation (KG_AGG_MFMA_GRID) at the training shapes (bench.agg_train_leg)
and at the C5a shape (bench.agg_leg)  (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from kinetic_gan_amd import _native as nv
dev = torch.device("cuda:0")
for grid in (0, 256, 512, 768, 1024, 1536, 2048, 4096):
    os.environ["KG_AGG_MFMA_GRID"] = str(grid); nv.reload_env()
    r = bench.agg_train_leg(64, dev)
    c = bench.agg_leg(dev)
    print("grid", grid, "train pair us", r["avg_pair_us"], "GB/s", r["achieved"], "| C5a ms", c["avg_launch_ms"], "GB/s", c["achieved"], flush=True)
