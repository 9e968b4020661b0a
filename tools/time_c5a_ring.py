#!/usr/bin/env python3
"""SURVEY 8(d) C5a stress block (512 channels, V=25, T=256, 64 samples): kg_conv's direct kernel against the ring tiles
(KG_CONV_RING=1, KG_CONV_RING_TILE) on the gcn contraction (1536 -> 512) and the temporal-conv tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, TAP_CHANBLOCK, Group, WView
dev = torch.device("cuda:0")
N, C, T, V = int(os.environ.get("N", "64")), 512, 256, 25
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cols = N * T * V
fl = 2.0 * cols * C * 3 * C
def setenv(tile):
    for k in ("KG_CONV_RING", "KG_CONV_RING_TILE"): os.environ.pop(k, None)
    if tile is not None:
        os.environ["KG_CONV_RING"] = "1"; os.environ["KG_CONV_RING_TILE"] = str(tile)
    nv.reload_env()
tiles = [None] + [int(t) for t in os.environ.get("TILES", "2,6,7").split(",")]
xa = nv.new_plane(N, 3 * C, T, V, dev).normal_()
wg = torch.randn(3 * C, C, 1, 1, device=dev) / (3 * C) ** 0.5
g = Group(xa, wg, WView(C * C, C, 1), C, 3, TAP_CHANBLOCK, 1, False, None)
ref = None
for t in tiles:
    setenv(t)
    nv.last_conv_plan = []
    out = nv.conv([g], N, C, T, V)
    plan = list(nv.last_conv_plan); nv.last_conv_plan = None
    if ref is None: ref = out.clone()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    ms = timeit(lambda: nv.conv([g], N, C, T, V))
    print(f"C5a gcn 1536->512, {cols} columns, ring tile {t}: plan {plan} {ms:7.2f} ms  {fl / ms * 1e-9:6.1f} TFLOP/s  err {err:.1e}", flush=True)
del xa, ref, out
z = nv.new_plane(N, C, T, V, dev).normal_()
x = nv.new_plane(N, C, T, V, dev).normal_()
wt = torch.randn(C, C, 3, 1, device=dev) / (3 * C) ** 0.5
gt = [Group(z, wt, WView(1, C * 3, 3), C, 3, TAP_TIME, 1, False, None)]
ref = None
for t in tiles:
    setenv(t)
    nv.last_conv_plan = []
    out = nv.conv(gt, N, C, T, V, add=x, act=nv.ACT_LRELU)
    plan = list(nv.last_conv_plan); nv.last_conv_plan = None
    if ref is None: ref = out.clone()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    ms = timeit(lambda: nv.conv(gt, N, C, T, V, add=x, act=nv.ACT_LRELU))
    print(f"C5a tail 3-tap 512->512 + identity residual + LeakyReLU, ring tile {t}: plan {plan} {ms:7.2f} ms  {fl / ms * 1e-9:6.1f} TFLOP/s  err {err:.1e}", flush=True)
