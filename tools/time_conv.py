#!/usr/bin/env python3
"""Workgroup timeline of one kg_conv launch (instrumented build, GPU box only): when every workgroup starts,
how long its setup / K-slice loop / epilogue take, and how the launch's span splits between them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = "/tmp/libkgan_timing.so"
from importlib import import_module
sys.path.insert(0, os.path.join(ROOT, "kinetic-gan_amd"))
src = [os.path.join(ROOT, "kinetic-gan_amd/csrc", f) for f in import_module("build").SOURCES]
EXTRA = os.environ.get("KG_EXTRA_DEFS", "").split()
if os.environ.get("KG_LIB"):       # a prebuilt instrumented variant (tools/build_variant.sh <tag> "-DKG_CONV_TIMING ...")
    lib = os.environ["KG_LIB"]
else:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "-DKG_CONV_TIMING"] + EXTRA + [
                       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "kinetic-gan_amd/csrc"), "-o", lib] + src + ["-ldl"])
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
nv.LIB_PATH = lib
from kinetic_gan_amd._native import TAP_TIME, TAP_CHANBLOCK, Group, WView
dev = torch.device("cuda:0")
N = int(os.environ.get("KG_TIME_N", "64"))

def tail(cin, cout, T, V, W, s):
    z = nv.new_plane(N, cout, T, W, dev).normal_(); x = nv.new_plane(N, cin, T, V, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev); wr = torch.randn(cout, cin, 1, 1, device=dev)
    keep = torch.arange(W, dtype=torch.int32, device=dev) if W != V else None      # (as the trunk: no vertex map without down-sampling)
    gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, s, False, None),
          Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, s, False, keep)]
    return lambda: nv.conv(gs, N, cout, T // s, W, act=nv.ACT_LRELU)

def gcn(cin, cout, T, W):
    xa = nv.new_plane(N, 3 * cin, T, W, dev).normal_()
    w = torch.randn(3 * cout, cin, 1, 1, device=dev)
    g = Group(xa, w, WView(cout * cin, cin, 1), cin, 3, TAP_CHANBLOCK, 1, False, None)
    return lambda: nv.conv([g], N, cout, T, W)

CASES = [("D1 tail 64 (s1)", tail(32, 64, 64, 11, 11, 1), ("2,1", "1,1")),
         ("D1 gcn 32->64", gcn(32, 64, 64, 11), ("2,1",)),
         ("D2 tail 128 (s2)", tail(64, 128, 64, 11, 5, 2), ("2,1",)),
         ("D3 gcn 128->256", gcn(128, 256, 32, 5), ("2,1",)),
         ("D3 tail 256 (s2)", tail(128, 256, 32, 5, 5, 2), ("2,4",))]
if os.environ.get("KG_TIME_CASES"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["KG_TIME_CASES"].split(","))]
orig_empty = torch.empty
for name, conv, plans in CASES:
    if os.environ.get("KG_TIME_PLANS"):
        plans = tuple(os.environ["KG_TIME_PLANS"].split(";"))
    for plan in plans:
        if plan == "bs":              # the bf16-split form (its tile kernel carries the same stamps)
            os.environ.pop("KG_CONV_PLAN", None); os.environ["KG_CONV_BS"] = "1"
        else:
            os.environ.pop("KG_CONV_BS", None); os.environ["KG_CONV_PLAN"] = plan
        nv.reload_env()
        last = {}
        def spy(*a, **k):
            t = orig_empty(*a, **k)
            if len(a) == 1 and isinstance(a[0], int) and a[0] >= (1 << 18): last["ws"] = t
            return t
        torch.empty = spy
        for _ in range(3): conv()
        torch.cuda.synchronize()
        last["ws"].zero_()
        last.clear()
        conv(); torch.cuda.synchronize()
        torch.empty = orig_empty
        raw = last["ws"].view(torch.int64)[-(1 << 17):].cpu()   # last 1 MiB as int64
        recs = raw.view(-1, 16)
        recs = recs[recs[:, 6] > 0].double()
        t0 = recs[:, 0].min()
        us = lambda x: x * 0.01
        start, setup, loop, epi = us(recs[:, 0] - t0), us(recs[:, 1] - recs[:, 0]), us(recs[:, 2] - recs[:, 1]), us(recs[:, 3] - recs[:, 2])
        end = us(recs[:, 3] - t0)
        hw = recs[:, 4].long(); xcc = recs[:, 5].long() & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        ncu = len(torch.unique(cu)); per_cu = torch.bincount(torch.unique(cu, return_inverse=True)[1])
        q = lambda v: "min %.2f med %.2f p90 %.2f max %.2f" % (v.min(), v.median(), v.quantile(0.9), v.max())
        if os.environ.get("KG_TIME_DUMP_CU"):      # which workgroups (linear launch index) share a CU, in start order
            allr = raw.view(-1, 16)
            idx = torch.nonzero(allr[:, 6] > 0).flatten()
            for c in torch.unique(cu)[:4].tolist() + torch.unique(cu)[-2:].tolist():
                sel = (cu == c).nonzero().flatten()
                order = sel[torch.argsort(recs[sel, 0])]
                print("    CU %5x:" % c, " ".join("%d@%.1f-%.1f" % (idx[i].item(), start[i].item(), end[i].item()) for i in order.tolist()))
        print(f"{name} plan {plan}: {len(recs)} workgroups on {ncu} CUs (per CU min {per_cu.min()} max {per_cu.max()}), "
              f"slices/WG {recs[:, 6].mean():.1f}, span first-start..last-end {end.max():.2f} us")
        print(f"    start offset  {q(start)}")
        print(f"    setup         {q(setup)}")
        print(f"    slice loop    {q(loop)}")
        print(f"    epilogue      {q(epi)}")
        print(f"    end offset    {q(end)}")
        if recs[:, 8:12].sum() > 0:
            nst = recs[:, 6] - (recs[:, 6] % 2) if not os.environ.get("KG_CONV_PERSIST") else recs[:, 6]
            seg = recs[:, 8:12] / nst.clamp(min=1)[:, None]
            print("    per slice (s_memtime cycles; direct kernel: mfma+issue / wait+stash / barrier / - (persistent: tile epilogues, per slice); LDS kernel: fetch issue / mfma / wait+stash / barrier): %.1f %.1f %.1f %.1f" % tuple(seg.mean(0).tolist()), flush=True)
