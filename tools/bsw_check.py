#!/usr/bin/env python3
"""round 5 (GPU box): where the hand-scheduled all-window bf16-split kernel differs from the direct kernel on one 3-tap conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, Group, WView
dev = torch.device("cuda:0")
N, Cin, M, T, V, taps, stride = [int(v) for v in os.environ.get("KG_CASE", "2,32,64,64,11,3,1").split(",")]
torch.manual_seed(0)
w = (torch.randn(M, Cin, taps, 1) / (Cin * taps) ** 0.5).to(dev)
x = nv.new_plane(N, Cin, T, V, dev).normal_()
g = Group(x, w, WView(sT=1, sO=Cin * taps, sI=taps), Cin, taps, TAP_TIME, stride, False, None)
RES = int(os.environ.get("KG_RES", "0"))        # channels of a 1 x 1 residual group (0: none)
gs = [g]
if RES:
    wr = (torch.randn(M, RES, 1, 1) / RES ** 0.5).to(dev)
    gs.append(Group(nv.new_plane(N, RES, T, V, dev).normal_(), wr, WView(0, RES, 1), RES, 1, TAP_TIME, stride, False, None))
outs = {}
TAGS = os.environ.get("KG_TAGS", "direct,bs,bsw").split(",")
for tag, env in [t for t in (("direct", {"KG_CONV_BS": "0"}), ("bs", {"KG_CONV_BS": "1", "KG_CONV_BS_ASM": "0"}), ("bsw", {"KG_CONV_BS": "1"})) if t[0] in TAGS]:
    for k in ("KG_CONV_BS", "KG_CONV_BS_ASM"):
        os.environ.pop(k, None)
    os.environ.update(env); nv.reload_env()
    nv.last_conv_plan = []
    outs[tag] = nv.conv(gs, N, M, T // stride, V).clone(); torch.cuda.synchronize()
    print(tag, "plan", nv.last_conv_plan)
ref = outs["direct"]
for tag in [t for t in ("bs", "bsw") if t in TAGS]:
    err = (outs[tag] - ref).abs()          # (N, M, T, V)
    print(tag, "max err", err.max().item(), "ref max", ref.abs().max().item())
    e = err.permute(1, 0, 2, 3).reshape(M, -1)      # rows x columns
    bad = (e > 1e-4)
    print("  bad rows:", bad.any(1).nonzero().flatten().tolist()[:70])
    cols = bad.any(0).nonzero().flatten()
    print("  bad columns: count", len(cols), "first", cols[:20].tolist(), "last", cols[-10:].tolist())
    print("  nan:", torch.isnan(outs[tag]).sum().item())
if os.environ.get("KG_REPEAT"):
    os.environ["KG_CONV_BS"] = "1"; os.environ.pop("KG_CONV_BS_ASM", None); nv.reload_env()
    first = nv.conv(gs, N, M, T // stride, V).clone()
    nbad = 0
    for i in range(int(os.environ["KG_REPEAT"])):
        o = nv.conv(gs, N, M, T // stride, V)
        d_ = (o - first).abs().max().item()
        if d_ != 0.0:
            nbad += 1
            e = (o - first).abs().permute(1, 0, 2, 3).reshape(M, -1)
            cols = (e > 0).any(0).nonzero().flatten()
            print("  repeat %d differs: max %.3e, %d columns, first %s" % (i, d_, len(cols), cols[:8].tolist()))
    print("repeats that differ from the first bsw run:", nbad)
