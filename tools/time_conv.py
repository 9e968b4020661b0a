#!/usr/bin/env python3
"""Per-segment cycle breakdown of kg_conv's K-slice loop (instrumented build, GPU box only)."""
import os, subprocess, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = "/tmp/libkgan_timing.so"
src = [os.path.join(ROOT, "kinetic-gan_amd/csrc", f) for f in ("kg_conv.hip", "kg_wgrad.hip", "kg_agg.hip", "kg_misc.hip")]
EXTRA = os.environ.get("KG_EXTRA_DEFS", "").split()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DKG_CONV_TIMING"] + EXTRA + [
                       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "kinetic-gan_amd/csrc"), "-o", lib] + src)
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
nv.LIB_PATH = lib
from kinetic_gan_amd._native import TAP_TIME, Group, WView
import ctypes as C
dev = torch.device("cuda:0")
N, cin, cout, T, V = 64, 32, 64, 64, 11
z = nv.new_plane(N, cout, T, V, dev).normal_(); x = nv.new_plane(N, cin, T, V, dev).normal_()
wt = torch.randn(cout, cout, 3, 1, device=dev); wr = torch.randn(cout, cin, 1, 1, device=dev)
gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, 1, False, None), Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, 1, False, None)]
# monkeypatch conv to keep the workspace tensor
keep = {}
orig_empty = torch.empty
def conv():
    return nv.conv(gs, N, cout, T, V, act=nv.ACT_LRELU)
for plan in ("2,1", "1,1", "3,1"):
    os.environ["KG_CONV_PLAN"] = plan
    # capture ws: wrap torch.empty used inside nv.conv
    last = {}
    def spy(*a, **k):
        t = orig_empty(*a, **k)
        if len(a) == 1 and isinstance(a[0], int) and a[0] >= (1 << 18): last["ws"] = t
        return t
    torch.empty = spy
    for _ in range(3): conv()
    torch.cuda.synchronize()
    last.clear()
    conv(); torch.cuda.synchronize()
    torch.empty = orig_empty
    ws = last["ws"]
    raw = ws.view(torch.int64)[-(1 << 17):].cpu()   # last 1 MiB as int64
    recs = raw.view(-1, 8)
    recs = recs[recs[:, 6] > 0]
    nsl = recs[:, 6].double()
    names = ["advance+rsrc", "W loads", "X loads", "mfma issue", "wait+stash", "barrier"]
    tot = recs[:, :6].double().sum(1)
    print(f"plan {plan}: {len(recs)} workgroups, slices/WG {nsl.mean():.1f}, loop cycles/WG {tot.mean():.0f} (memtime ticks)")
    for i, n in enumerate(names):
        print(f"    {n:12s} {(recs[:, i].double() / nsl).mean():8.0f} ticks/slice")
