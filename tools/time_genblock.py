#!/usr/bin/env python3
"""Fused generator block launches (kg_genblock_fwd / _bwd) at the training shapes: hipGraph-replayed time per launch next to
the staged sequence, and - with a -DKG_GB_STAMP build (KG_LIB=build_ab/libkgan_gbstamp.so) - the phase stamps of workgroup 0.
   tools/build_variant.sh gbstamp "-DKG_GB_STAMP" kg_genblock.hip;  KG_LIB=build_ab/libkgan_gbstamp.so python tools/time_genblock.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kinetic_gan_amd  # noqa: F401,E402
from kinetic_gan_amd import _native as nv  # noqa: E402
from kinetic_gan_amd.graph import build_graph  # noqa: E402

d = torch.device("cuda:0")
gr = build_graph("ntu")
N = int(os.environ.get("KG_TIME_N", "128"))
BLOCKS = [  # name, lvl, up_s, Cin, C, Tc, rep, res, bn_t, act
    ("G3", 2, False, 128, 64, 4, 2, "conv", True, nv.ACT_LRELU),
    ("G4", 1, True, 64, 32, 8, 2, "conv", False, nv.ACT_LRELU),
    ("G5", 1, False, 32, 3, 16, 2, "conv", True, nv.ACT_LRELU),
    ("G6", 0, True, 3, 3, 32, 2, "identity", False, nv.ACT_TANH),
]
_only = os.environ.get("KG_TIME_BLOCKS")
if _only:
    BLOCKS = [b for b in BLOCKS if b[0] in _only.split(",")]
stamps_fn = None
raw = ctypes.CDLL(nv.LIB_PATH)
if hasattr(raw, "kg_gb_read_stamps"):
    stamps_fn = raw.kg_gb_read_stamps


def graph_time(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def stamps(direction):
    if stamps_fn is None:
        return ""
    buf = (ctypes.c_longlong * 32)()
    torch.cuda.synchronize()
    stamps_fn(buf)
    v = list(buf)[direction * 16:direction * 16 + 8]
    c = list(buf)[direction * 16 + 8:direction * 16 + 16]
    last = max(i for i in range(8) if v[i] > 0)
    mhz = (c[last] - c[0]) / max(1e-9, (v[last] - v[0]) / 100.0)
    # wall_clock64: 100 MHz; s_memtime: shader clock
    return "  phases us: " + " ".join("%.1f" % ((v[i + 1] - v[i]) / 100.0) for i in range(7) if v[i + 1] > v[i]) + "  | shader clock %.0f MHz" % mhz


for name, lvl, up_s, Cin, C, Tc, rep, res, bn_t, act in BLOCKS:
    V = gr.num_node[lvl]
    U = torch.as_tensor(gr.upsample_matrix(lvl), dtype=torch.float32, device=d).contiguous() if up_s else None
    Vc = U.shape[0] if up_s else V
    K, T = 3, Tc * rep
    dims = nv.GenBlockDims(Cin=Cin, C=C, K=K, Kp=K, Tc=Tc, Vc=Vc, T=T, V=V, rep=rep, res_kind={"none": 0, "identity": 1, "conv": 2}[res],
                           bn_t=bn_t, act=act)
    B = torch.rand(K, Vc, V, device=d).contiguous()
    wg = torch.randn(K * C, Cin, 1, 1, device=d) / Cin ** 0.5
    wr = torch.randn(C, Cin, 1, 1, device=d) / Cin ** 0.5 if res == "conv" else None
    br = torch.randn(C, device=d) if res == "conv" else None
    wt = torch.randn(C, C, 3, 1, device=d) / (3 * C) ** 0.5
    bt = torch.randn(C, device=d)
    nw = torch.randn(C, device=d) * 0.3
    noise = torch.randn(N, 1, T, V, device=d)
    x = nv.new_plane(N, Cin, Tc, Vc, d).normal_()

    def bn():
        return dict(gamma=torch.rand(C, device=d) + 0.5, beta=torch.randn(C, device=d), running_mean=torch.zeros(C, device=d),
                    running_var=torch.ones(C, device=d), num_batches_tracked=torch.zeros((), dtype=torch.int64, device=d), momentum=0.1, eps=1e-5)
    bts, brs = (bn() if bn_t else None), (bn() if res == "conv" else None)

    def fwd():
        return nv.genblock_fwd(dims, x=x, wg=wg, wr=wr, br=br, wt=wt, bt=bt, B=B, U=U, bn_t=bts, bn_r=brs, groups=2, noise=noise, nw=nw)
    t = graph_time(fwd)
    fwd()
    sub = ""
    if hasattr(raw, "kg_gb_read_sub"):
        b8 = (ctypes.c_longlong * 8)()
        torch.cuda.synchronize()
        raw.kg_gb_read_sub(b8)
        v8 = list(b8)
        sub = "  [expand phase: yc store %.1f, expand %.1f, residual %.1f us; first tile of wave 0: K loop done +%.1f, epilogue +%.1f]" % (
            tuple((v8[i + 1] - v8[i]) / 100.0 for i in range(3)) + ((v8[4] - v8[1]) / 100.0, (v8[5] - v8[4]) / 100.0))
    print("%s fwd  N=%d: %6.1f us%s%s" % (name, N, t, stamps(0), sub))
    nb = N // 2
    g = nv.new_plane(nb, C, T, V, d).normal_()
    out = nv.new_plane(nb, C, T, V, d).normal_().tanh_()
    u, r = nv.new_plane(nb, C, T, V, d).normal_(), nv.new_plane(nb, C, T, V, d).normal_()
    coef = torch.randn(6, C, device=d)
    px = nv.new_plane(nb, Cin, Tc, Vc, d).normal_().tanh_()
    pu, pr = nv.new_plane(nb, Cin, Tc, Vc, d).normal_(), nv.new_plane(nb, Cin, Tc, Vc, d).normal_()
    st = (torch.rand(Cin, device=d) + 0.5, torch.zeros(Cin, device=d), torch.ones(Cin, device=d))
    sinks = {k: torch.zeros(Cin, device=d) for k in ("nw", "gamma_t", "beta_t", "gamma_r", "beta_r")}
    prev = dict(x=px, u=pu, r=pr, noise=torch.randn(nb, 1, Tc, Vc, device=d), act=nv.ACT_LRELU, bn_t=st, bn_r=st, sinks=sinks)

    def bwd():
        return nv.genblock_bwd(dims, g=g, out=out, u=u if bn_t else None, r=r if res == "conv" else None, coef=coef, wg=wg, wr=wr, wt=wt,
                               B=B, U=U, prev=prev)
    t = graph_time(bwd)
    bwd()
    print("%s bwd  N=%d: %6.1f us%s" % (name, nb, t, stamps(1)))
