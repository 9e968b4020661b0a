#!/usr/bin/env python3
"""Run one kg_wgrad shape a few times (for rocprofv3 --pmc): D1 temporal conv, 64 -> 64 channels, 3 taps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, WView
dev = torch.device("cuda:0")
N, Cin, M, T, V, taps = 128, 64, 64, 64, 11, 3
x = nv.new_plane(N, Cin, T, V, dev).normal_()
g = nv.new_plane(N, M, T, V, dev).normal_()
for _ in range(int(os.environ.get("REPS", "10"))):
    nv.wgrad(g, x, Cin, taps, TAP_TIME, 1, None, M * Cin * taps, WView(1, Cin * taps, taps))
torch.cuda.synchronize()
