#!/usr/bin/env python3
"""Run one kg_conv shape a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, Group, WView
dev = torch.device("cuda:0")
# SHAPE="N,cin,cout,T,V,W,stride" (default: discriminator block 1 tail at bs=64)
N, cin, cout, T, V, W, s = [int(v) for v in os.environ.get("SHAPE", "64,32,64,64,11,11,1").split(",")]
z = nv.new_plane(N, cout, T, W, dev).normal_()
x = nv.new_plane(N, cin, T, V, dev).normal_()
wt = torch.randn(cout, cout, 3, 1, device=dev); wr = torch.randn(cout, cin, 1, 1, device=dev)
keep = torch.arange(W, dtype=torch.int32, device=dev)
gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, s, False, None),
      Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, s, False, keep)]
for _ in range(int(os.environ.get("REPS", "10"))):
    nv.conv(gs, N, cout, T // s, W, act=nv.ACT_LRELU)
torch.cuda.synchronize()
