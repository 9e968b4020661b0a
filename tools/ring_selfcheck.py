#!/usr/bin/env python3
"""Round 5: every kg_conv problem a critic forward + backward issues (config CFG, N samples), run on the direct kernel and on
a ring tile (RT) with the same operands; prints the problems whose outputs differ by more than 1e-5 relative."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
import bench
dev = torch.device("cuda:0")
cfgname = os.environ.get("CFG", "stress"); N = int(os.environ.get("N", "64")); RT = os.environ.get("RT", "6")
cfg = bench.CONFIGS[cfgname]
G, D = bench.build_models(cfg, dev)
calls = []
orig_conv, orig_many = nv.conv, nv.conv_many
def rec_conv(groups, N, M, T_out, V_out, **kw):
    calls.append(dict(groups=groups, N=N, M=M, T_out=T_out, V_out=V_out, **kw))
    return orig_conv(groups, N, M, T_out, V_out, **kw)
def rec_many(jobs):
    if len(jobs) > 1:
        for j in jobs: calls.append(dict(j))
    return orig_many(jobs)
nv.conv, nv.conv_many = rec_conv, rec_many
real, labels, z, alpha = bench.synth_batch(cfg, N, 0, dev)
x = real.clone().requires_grad_(True)
out = D(x, labels)
out.sum().backward()
torch.cuda.synchronize()
nv.conv, nv.conv_many = orig_conv, orig_many
print(f"{len(calls)} kg_conv problems recorded", flush=True)
bad = 0
for i, c in enumerate(calls):
    c = dict(c); c.pop("out", None); c.pop("out_t0", None); c.pop("out_tstride", None)
    for v in ("KG_CONV_RING", "KG_CONV_RING_TILE"): os.environ.pop(v, None)
    nv.reload_env()
    ref = orig_conv(**c).clone()
    os.environ["KG_CONV_RING"] = "1"; os.environ["KG_CONV_RING_TILE"] = RT; nv.reload_env()
    nv.last_conv_plan = []
    got = orig_conv(**c); plan = list(nv.last_conv_plan); nv.last_conv_plan = None
    torch.cuda.synchronize()
    err = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    desc = f"N={c['N']} M={c['M']} T={c['T_out']} V={c['V_out']} K=" + "+".join(
        f"{g.taps}x{g.Cin}{'T' if g.transposed else ''}{'s%d' % g.t_stride if g.t_stride > 1 else ''}{'g' if g.vmap is not None else ''}" for g in c["groups"])
    flag = "  <-- MISMATCH" if err > 1e-5 else ""
    if flag: bad += 1
    if flag or os.environ.get("VERBOSE"):
        print(f"#{i:3d} {desc:50s} plan {plan} add={c.get('add') is not None} mask={c.get('mask') is not None} err {err:.2e}{flag}", flush=True)
print("mismatching problems:", bad)
