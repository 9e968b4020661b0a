#!/usr/bin/env python3
"""kg_wgrad_many on the critic step's 16 weight gradients (two operand pairs each: 128 + 64 samples), all layers in
one call and each layer alone (GPU box).  profiles/r02_time_wgrad_many.log also holds the numbers of the three-tap
workgroup experiment (not kept: slower inside the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
if os.environ.get("KG_LIB"):                       # experiment builds (tools/gpu_ab.sh)
    nv.LIB_PATH = os.environ["KG_LIB"]
from kinetic_gan_amd._native import TAP_CHANBLOCK, TAP_TIME, WView
from tools.time_aggconv import timeit

dev = torch.device("cuda:0")
# (name, Cin, M, T_in, V_in(=W of x for gcn), V_out, taps, mode, stride)
L = [("D0 gcn", 3, 32, 64, 11, 11, 3, TAP_CHANBLOCK, 1), ("D0 tcn", 32, 32, 64, 11, 11, 3, TAP_TIME, 1),
     ("D1 gcn", 32, 64, 64, 11, 11, 3, TAP_CHANBLOCK, 1), ("D1 tcn", 64, 64, 64, 11, 11, 3, TAP_TIME, 1), ("D1 res", 32, 64, 64, 11, 11, 1, TAP_TIME, 1),
     ("D2 gcn", 64, 128, 64, 5, 5, 3, TAP_CHANBLOCK, 1), ("D2 tcn", 128, 128, 64, 5, 5, 3, TAP_TIME, 2), ("D2 res", 64, 128, 64, 5, 5, 1, TAP_TIME, 2),
     ("D3 gcn", 128, 256, 32, 5, 5, 3, TAP_CHANBLOCK, 1), ("D3 tcn", 256, 256, 32, 5, 5, 3, TAP_TIME, 2), ("D3 res", 128, 256, 32, 5, 5, 1, TAP_TIME, 2),
     ("D4 gcn", 256, 512, 16, 1, 1, 3, TAP_CHANBLOCK, 1), ("D4 tcn", 512, 512, 16, 1, 1, 3, TAP_TIME, 2), ("D4 res", 256, 512, 16, 1, 1, 1, TAP_TIME, 2),
     ("D5 gcn", 512, 512, 8, 1, 1, 3, TAP_CHANBLOCK, 1), ("D5 tcn", 512, 512, 8, 1, 1, 3, TAP_TIME, 2)]
jobs, flops = [], []
for name, cin, m, T, V, Vo, taps, mode, s in L:
    prs = []
    for n in (128, 64):
        xc = cin * (taps if mode == TAP_CHANBLOCK else 1)
        prs.append((nv.new_plane(n, m, T // s, Vo, dev).normal_(), nv.new_plane(n, xc, T, V, dev).normal_()))
    wv = WView(1, cin * taps, taps) if mode == TAP_TIME else WView(m * cin, cin, 1)
    jobs.append(dict(g=prs[0][0], x=prs[0][1], Cin=cin, taps=taps, tap_mode=mode, t_stride=s, vmap=None, wv=wv,
                     out=torch.zeros(m * cin * taps, device=dev), accumulate=True, extra=[prs[1]]))
    flops.append(2.0 * taps * m * cin * 192 * (T // s) * Vo)
for mode in ("per-tap",):
    t = timeit(lambda: nv.wgrad_many(jobs), reps=5)
    print("%-10s all 16 layers: %7.1f us  (%.1f TF)" % (mode, t, sum(flops) / t / 1e6), flush=True)
    for j, f, l in zip(jobs, flops, L):
        t1 = timeit(lambda: nv.wgrad_many([j]), reps=10)
        print("    %-8s %7.1f us  %5.1f TF" % (l[0], t1, f / t1 / 1e6), flush=True)
