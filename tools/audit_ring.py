#!/usr/bin/env python3
"""Ring audit (GPU box, round 5): record every kg_conv problem one G+D iteration issues (kg_conv and kg_conv_many jobs), then
time each distinct problem on the automatic (direct-kernel) plan and on every tile of the persistent LDS-ring form that is
eligible for it (KG_CONV_RING=1, KG_CONV_RING_TILE); prints where a ring tile is ahead, and the sums."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
import bench

dev = torch.device("cuda:0")
TILES = {0: "128x128", 1: "64x128", 2: "32x128", 3: "64x64", 4: "32x64", 9: "K32x32"}

def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3

# ---- record
calls = []
orig_conv, orig_many = nv.conv, nv.conv_many
def rec_conv(groups, N, M, T_out, V_out, **kw):
    calls.append(dict(groups=groups, N=N, M=M, T_out=T_out, V_out=V_out, **kw))
    return orig_conv(groups, N, M, T_out, V_out, **kw)
def rec_many(jobs):
    for j in jobs: calls.append(dict(j))
    return orig_many(jobs)
nv.conv, nv.conv_many = rec_conv, rec_many
cfg = bench.CONFIGS["ntu"]
G, D = bench.build_models(cfg, dev)
from kinetic_gan_amd.wgan_gp import Trainer
tr = Trainer(G, D)
batch = bench.synth_batch(cfg, 64, 0, dev)
real, labels, z, alpha = batch
tr.iteration(real, labels, z, alpha, None, None, with_g=True)
calls.clear()
tr.iteration(real, labels, z, alpha, None, None, with_g=True)
torch.cuda.synchronize()
nv.conv, nv.conv_many = orig_conv, orig_many

def key(c):
    gs = tuple((g.Cin, g.taps, g.tap_mode, g.t_stride, bool(g.transposed), g.vmap is not None, tuple(g.x.shape), g.wv) for g in c["groups"])
    return (c["N"], c["M"], c["T_out"], c["V_out"], gs, c.get("add") is not None, c.get("mask") is not None, c.get("out_tstride", 1))
uniq = collections.OrderedDict()
for c in calls:
    k = key(c)
    if k in uniq: uniq[k][1] += 1
    else: uniq[k] = [c, 1]
print(f"{len(calls)} kg_conv problems per iteration, {len(uniq)} distinct", flush=True)
RING = list(range(11))
tot_auto = tot_best = 0.0
wins = 0
for k, (c, cnt) in uniq.items():
    c = dict(c); c.pop("out", None); c.pop("out_t0", None); c.pop("out_tstride", None)
    fn = lambda: orig_conv(**c)
    for v in ("KG_CONV_RING", "KG_CONV_RING_TILE"): os.environ.pop(v, None)
    nv.reload_env()
    nv.last_conv_plan = []
    fn(); plan = list(nv.last_conv_plan); nv.last_conv_plan = None
    auto = timeit(fn)
    rows = []
    for t in RING:
        os.environ["KG_CONV_RING"] = "1"; os.environ["KG_CONV_RING_TILE"] = str(t); nv.reload_env()
        nv.last_conv_plan = []
        fn(); p2 = list(nv.last_conv_plan); nv.last_conv_plan = None
        if p2[0] != 20 + t:
            continue            # (not eligible on this tile: the direct kernel ran)
        rows.append((timeit(fn, 10), f"ring{t}"))
    for v in ("KG_CONV_RING", "KG_CONV_RING_TILE"): os.environ.pop(v, None)
    nv.reload_env()
    rows.sort()
    best = min(rows[0][0], auto) if rows else auto
    tot_auto += cnt * auto; tot_best += cnt * best
    desc = f"N={c['N']} M={c['M']} T={c['T_out']} V={c['V_out']} K=" + "+".join(f"{g.taps}x{g.Cin}{'T' if g.transposed else ''}" for g in c["groups"])
    flag = ""
    if rows and rows[0][0] < 0.97 * auto:
        flag = "  <-- ring wins"; wins += 1
    print(f"x{cnt} {desc:46s} auto {plan[0]}/k{plan[1]} {auto:6.1f} us | ring: " + ("  ".join(f"{n} {u:.1f}" for u, n in rows[:3]) if rows else "not eligible") + flag, flush=True)
print(f"sum over the iteration: direct plans {tot_auto:.0f} us, with the best ring tile where one wins {tot_best:.0f} us; problems a ring tile wins: {wins}")
