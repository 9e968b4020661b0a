#!/usr/bin/env python3
"""round 5 (GPU box): the tails whose packed weights the trainer caches (D0: 32 -> 32 channels, no residual conv; D1: 64 channels
+ 1x1 residual from 32) at the batch sizes of an iteration's passes (64: generator step, 128: the penalty's double backward,
192 / 384: critic forward), direct fp32 kernel against the bf16-split tile kernel on packed weights (hipGraph replay of 20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, Group, WView
dev = torch.device("cuda:0")
T, V = 64, 11


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for name, cout, cin in (("D0 tail 32", 32, 0), ("D1 tail 64", 64, 32)):
    wt = torch.randn(cout, cout, 3, 1, device=dev) * 0.05
    wr = torch.randn(cout, cin, 1, 1, device=dev) * 0.1 if cin else None
    bt = torch.randn(cout, device=dev)
    for n in (64, 128, 192, 384):
        z = nv.new_plane(n, cout, T, V, dev).normal_()
        gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, 1, False, None)]
        if cin:
            gs.append(Group(nv.new_plane(n, cin, T, V, dev).normal_(), wr, WView(0, cin, 1), cin, 1, TAP_TIME, 1, False, None))
        pack = nv.conv_pack(gs, n, cout, T, V)
        for lin in (False, True):
            mask = nv.new_plane(n, cout, T, V, dev).normal_() if lin else None
            kw = dict(mask=mask) if lin else dict(bias0=bt, act=nv.ACT_LRELU)
            td = timed(lambda: nv.conv(gs, n, cout, T, V, **kw))
            tp = timed(lambda: nv.conv(gs, n, cout, T, V, wpack=pack, **kw))
            print("%-11s n=%3d %-8s direct %6.2f us   packed %6.2f us   (%.3f)" % (name, n, "linear" if lin else "forward", td, tp, tp / td), flush=True)
    print("%-11s pack launch %.2f us" % (name, timed(lambda: nv.conv_pack(gs, 1, cout, T, V, out=pack))))
