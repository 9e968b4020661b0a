#!/usr/bin/env python3
"""hipGraph-replayed time of the critic step, the generator step and the whole iteration (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kinetic_gan_amd.wgan_gp import Trainer

dev = torch.device("cuda:0")
cfg = bench.CONFIGS[os.environ.get("CFG", "ntu")]
n = int(os.environ.get("N", "64"))
G, D = bench.build_models(cfg, dev)
tr = Trainer(G, D)
real, labels, z, alpha = bench.synth_batch(cfg, n, 0, dev)

def timeit(fn, reps=30):
    rep = bench._capture(fn)
    for _ in range(3): rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): rep()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

parts = {
    "d_step": lambda: tr.d_step(real, labels, z, alpha, None),
    "g_step": lambda: tr.g_step(labels, z, None),
    "iteration": lambda: tr.iteration(real, labels, z, alpha, None, None, with_g=True),
    "G_fwd_nograd": lambda: torch.no_grad().__enter__() or G(z, labels),
}
def g_fwd():
    with torch.no_grad():
        return G(z, labels)
parts["G_fwd_nograd"] = g_fwd
def d_fwd():
    with torch.no_grad():
        return D(real, labels)
parts["D_fwd_nograd"] = d_fwd
def g_fwd2():
    with torch.no_grad():
        w = G.mapping(z, labels)
        return G.synthesis(torch.cat((w, w), 0))
parts["G_synth_2n_nograd"] = g_fwd2
def g_map():
    with torch.no_grad():
        return G.mapping(z, labels)
parts["G_map_nograd"] = g_map
gsum = torch.randn(n, cfg["channels"], cfg["t_size"], cfg["v"], device=dev)
def g_fb():
    tr.fG.zero_grad()
    out = G(z, labels)
    (out * gsum).sum().backward()
    tr.fG.gather_grads()
parts["G_fwd_bwd_n"] = g_fb
def g_fb_nomap():
    tr.fG.zero_grad()
    with torch.no_grad():
        w = G.mapping(z, labels)
    out = G.synthesis(w.requires_grad_(True))
    (out * gsum).sum().backward()
    tr.fG.gather_grads()
parts["G_synth_fwd_bwd_n"] = g_fb_nomap
only = os.environ.get("PARTS")
if only:
    parts = {k: v for k, v in parts.items() if k in only.split(",")}
for k, fn in parts.items():
    print("%-14s %.3f ms" % (k, timeit(fn)), flush=True)
