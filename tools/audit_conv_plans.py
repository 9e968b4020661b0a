#!/usr/bin/env python3
"""Plan audit (GPU box): record every kg_conv problem one G+D iteration issues (kg_conv and kg_conv_many jobs), then time
each distinct problem under the automatic plan and under forced (tile, K-split) plans; prints the problems whose automatic
plan is more than 4 % behind the best forced one, and the sum over the iteration."""
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
CFG_NAME, BATCH = os.environ.get("KG_AUDIT_CONFIG", "ntu"), int(os.environ.get("KG_AUDIT_BATCH", "64"))
print(f"plan audit: --config {CFG_NAME} --batch {BATCH}", flush=True)
cfg = bench.CONFIGS[CFG_NAME]
G, D = bench.build_models(cfg, dev)
from kinetic_gan_amd.wgan_gp import Trainer
tr = Trainer(G, D)
batch = bench.synth_batch(cfg, BATCH, 0, dev)
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
tot_auto = tot_best = 0.0
for k, (c, cnt) in uniq.items():
    c = dict(c); c.pop("out", None); c.pop("out_t0", None); c.pop("out_tstride", None)
    fn = lambda: orig_conv(**c)
    os.environ.pop("KG_CONV_PLAN", None); nv.reload_env()
    nv.last_conv_plan = []
    fn(); plan = list(nv.last_conv_plan); nv.last_conv_plan = None
    auto = timeit(fn)
    rows = []
    s_total = sum(g.taps * ((g.Cin + 31) // 32) for g in c["groups"])
    for t in TILES:
        for ns in (1, 2, 4, 8):
            if ns > max(1, s_total // 2): continue
            os.environ["KG_CONV_PLAN"] = f"{t},{ns}"; nv.reload_env()
            try:
                rows.append((timeit(fn, 10), f"{TILES[t]}/k{ns}"))
            except RuntimeError:
                pass
    os.environ.pop("KG_CONV_PLAN", None); nv.reload_env()
    rows.sort()
    best = min(rows[0][0], auto) if rows else auto
    tot_auto += cnt * auto; tot_best += cnt * best
    g0 = c["groups"][0]
    desc = f"N={c['N']} M={c['M']} T={c['T_out']} V={c['V_out']} K=" + "+".join(f"{g.taps}x{g.Cin}{'T' if g.transposed else ''}" for g in c["groups"])
    flag = "  <--" if rows and auto > 1.04 * rows[0][0] else ""
    print(f"x{cnt} {desc:46s} auto {TILES.get(plan[0], plan[0])}/k{plan[1]} {auto:6.1f} us | best " + "  ".join(f"{n} {u:.1f}" for u, n in rows[:3]) + flag, flush=True)
print(f"sum over the iteration: auto {tot_auto:.0f} us, best forced {tot_best:.0f} us")
