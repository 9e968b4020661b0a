#!/usr/bin/env python3
"""Time kg_wgrad (image / per-tap kernels) at the discriminator's shapes (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
if os.environ.get("KG_LIB"):
    nv.LIB_PATH = os.environ["KG_LIB"]
from kinetic_gan_amd._native import TAP_TIME, TAP_CHANBLOCK, WView
from time_agg import timeit

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "128"))
CASES = [("D1 gcn 32->64", 32, 64, 64, 11, 3, TAP_CHANBLOCK, 1), ("D1 tcn 64", 64, 64, 64, 11, 3, TAP_TIME, 1),
         ("D1 res 32->64", 32, 64, 64, 11, 1, TAP_TIME, 1), ("D2 gcn 64->128", 64, 128, 64, 5, 3, TAP_CHANBLOCK, 1),
         ("D2 tcn 128 s2", 128, 128, 64, 5, 3, TAP_TIME, 2), ("D3 gcn 128->256", 128, 256, 32, 5, 3, TAP_CHANBLOCK, 1),
         ("D3 tcn 256 s2", 256, 256, 32, 5, 3, TAP_TIME, 2), ("D4 gcn 256->512", 256, 512, 16, 1, 3, TAP_CHANBLOCK, 1),
         ("D4 tcn 512 s2", 512, 512, 16, 1, 3, TAP_TIME, 2), ("D5 tcn 512 s2", 512, 512, 8, 1, 3, TAP_TIME, 2)]
for name, Cin, M, T, V, taps, mode, s in CASES:
    xc = Cin * (taps if mode == TAP_CHANBLOCK else 1)
    x = nv.new_plane(N, xc, T, V, dev).normal_()
    g = nv.new_plane(N, M, T // s, V, dev).normal_()
    if mode == TAP_CHANBLOCK:
        wv, numel = WView(M * Cin, Cin, 1), taps * M * Cin
    else:
        wv, numel = WView(1, Cin * taps, taps), M * Cin * taps
    fn = lambda: nv.wgrad(g, x, Cin, taps, mode, s, None, numel, wv)
    fl = 2.0 * N * (T // s) * V * M * Cin * taps
    t0 = timeit(fn)
    print(f"{name:18s} {t0:6.1f} us  ({fl / t0 * 1e-6:5.1f} TF incl. slab reduction)", flush=True)
