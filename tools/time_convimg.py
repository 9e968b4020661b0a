#!/usr/bin/env python3
"""kg_conv image form (kg_convimg.hip) against the direct kernel at the D0 / D1 shapes it takes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
if os.environ.get("KG_LIB"):                       # experiment builds (tools/gpu_ab.sh)
    nv.LIB_PATH = os.environ["KG_LIB"]
from kinetic_gan_amd._native import TAP_TIME, Group, WView
from tools.time_aggconv import timeit

dev = torch.device("cuda:0")
CASES = [("D1 tail  M64 K224", 64, 64, 3, 32, False), ("D1 tcn^T M64 K192", 64, 64, 3, 0, True),
         ("D0 tail  M32 K96", 32, 32, 3, 0, False), ("D1 res^T M32 K64", 32, 64, 1, 0, True)]
for N in (64, 192):
    for name, M, C0, taps, C1, tr in CASES:
        T, V = 64, 11
        x = nv.new_plane(N, C0, T, V, dev).normal_()
        if tr:
            w = torch.randn(C0, M, taps, 1, device=dev) * 0.1
            gs = [Group(x, w, WView(1 if taps > 1 else 0, taps, M * taps), C0, taps, TAP_TIME, 1, True, None)]
        else:
            w = torch.randn(M, C0, taps, 1, device=dev) * 0.1
            gs = [Group(x, w, WView(1 if taps > 1 else 0, C0 * taps, taps), C0, taps, TAP_TIME, 1, False, None)]
        if C1:
            x2 = nv.new_plane(N, C1, T, V, dev).normal_()
            gs.append(Group(x2, torch.randn(M, C1, 1, 1, device=dev) * 0.1, WView(0, C1, 1), C1, 1, TAP_TIME, 1, False, None))
        b = torch.randn(M, device=dev)
        fn = lambda: nv.conv(gs, N, M, T, V, bias0=b, act=nv.ACT_LRELU)
        res = {}
        for img in ("1", "0"):
            os.environ["KG_CONV_IMG"] = img; nv.reload_env()
            res[img] = timeit(fn)
        os.environ.pop("KG_CONV_IMG"); nv.reload_env()      # (1 = image form, 0 / unset = direct kernel)
        fl = 2.0 * M * (C0 * taps + C1) * N * T * V
        print("N=%3d %-20s image %6.1f us (%5.1f TF, %.2f of peak) | direct %6.1f us" % (N, name, res["1"], fl / res["1"] / 1e6, fl / res["1"] / 1e6 / 157.3, res["0"]), flush=True)
