#!/usr/bin/env python3
"""Round 5 debugging aid: the critic's input gradient on the C5b stress config for the whole batch and for its halves, on the
direct kernel and with a ring tile forced (RT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
import bench
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["stress"]
G, D = bench.build_models(cfg, dev)
real, labels, z, alpha = bench.synth_batch(cfg, 64, 0, dev)
def critic(x, lab):
    x = x.clone().requires_grad_(True)
    D.zero_grad()
    out = D(x, lab)
    out.sum().backward()
    torch.cuda.synchronize()
    return out.detach().clone(), x.grad.detach().clone()
def l2(a, b): return ((a - b).double().norm() / b.double().norm()).item()
res = {}
for tag, rt in (("direct", None), ("ring", os.environ.get("RT", "6"))):
    for v in ("KG_CONV_RING", "KG_CONV_RING_TILE"): os.environ.pop(v, None)
    if rt is not None:
        os.environ["KG_CONV_RING"] = "1"; os.environ["KG_CONV_RING_TILE"] = rt
    nv.reload_env()
    f = critic(real, labels); a = critic(real[:32], labels[:32]); b = critic(real[32:], labels[32:])
    res[tag] = (f, a, b)
    print(tag, "split=cat: out", l2(torch.cat([a[0], b[0]]), f[0]), "gx", l2(torch.cat([a[1], b[1]]), f[1]), flush=True)
for i, name in enumerate(("full", "first half", "second half")):
    print(name, "ring vs direct: out", l2(res["ring"][i][0], res["direct"][i][0]), "gx", l2(res["ring"][i][1], res["direct"][i][1]))
# twice the same thing on the ring: deterministic?
f2 = critic(real, labels)
print("ring, full batch again: gx", l2(f2[1], res["ring"][0][1]))
