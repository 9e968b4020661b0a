#!/usr/bin/env python3
"""SURVEY 8(d) C5a stress shapes: kg_conv on one D-style block with 512 channels, V=25, T=256, 64 samples
(GPU box only).  Also times the vendor fp32 GEMM (torch.mm -> hipBLASLt/rocBLAS) of the same M x K x columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
if os.environ.get("KG_LIB"):
    nv.LIB_PATH = os.environ["KG_LIB"]
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
xa = nv.new_plane(N, 3 * C, T, V, dev).normal_()
wg = torch.randn(3 * C, C, 1, 1, device=dev) / (3 * C) ** 0.5
g = Group(xa, wg, WView(C * C, C, 1), C, 3, TAP_CHANBLOCK, 1, False, None)
fl = 2.0 * cols * C * 3 * C
for plan in (None, "0,1", "1,1", "2,1"):
    if plan: os.environ["KG_CONV_PLAN"] = plan; nv.reload_env()
    ms = timeit(lambda: nv.conv([g], N, C, T, V))
    print(f"C5a gcn 1536->512, {cols} columns, plan {plan or 'auto':5s}: {ms:7.2f} ms  {fl / ms * 1e-9:6.1f} TFLOP/s", flush=True)
os.environ.pop("KG_CONV_PLAN", None); nv.reload_env()
del xa
z = nv.new_plane(N, C, T, V, dev).normal_()
x = nv.new_plane(N, C, T, V, dev).normal_()
wt = torch.randn(C, C, 3, 1, device=dev) / (3 * C) ** 0.5
gt = [Group(z, wt, WView(1, C * 3, 3), C, 3, TAP_TIME, 1, False, None)]
for plan in (None, "0,1", "1,1", "2,1"):
    if plan: os.environ["KG_CONV_PLAN"] = plan; nv.reload_env()
    ms = timeit(lambda: nv.conv(gt, N, C, T, V, add=x, act=nv.ACT_LRELU))
    print(f"C5a tail 3-tap 512->512 + identity residual + LeakyReLU, plan {plan or 'auto':5s}: {ms:7.2f} ms  {fl / ms * 1e-9:6.1f} TFLOP/s", flush=True)
os.environ.pop("KG_CONV_PLAN", None); nv.reload_env()
del z, x
a = torch.randn(C, 3 * C, device=dev); b = torch.randn(3 * C, cols, device=dev)
ms = timeit(lambda: torch.mm(a, b))
print(f"vendor fp32 GEMM (torch.mm) {C} x {3 * C} x {cols}: {ms:7.2f} ms  {fl / ms * 1e-9:6.1f} TFLOP/s", flush=True)
