#!/usr/bin/env python3
"""kg_aggconv (fused aggregation + gcn GEMM) against kg_agg_expand + kg_conv at the discriminator's shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_CHANBLOCK, Group, WView
from kinetic_gan_amd.graph import build_graph

def timeit(fn, reps=20):
    """GPU time per call (us): the calls are captured in a hipGraph so host launch overhead is not measured."""
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

dev = torch.device("cuda:0")
g = build_graph("ntu")
def table(A):
    K, V, W = A.shape
    tab = np.full((K, W, 4), -1, np.int32); pc = [0, 0, 0]
    for k in range(K):
        for w in range(W):
            vs = np.nonzero(A[k, :, w])[0]; pc[k] = max(pc[k], len(vs)); tab[k, w, :len(vs)] = vs
    return torch.as_tensor(tab, device=dev), pc
CASES = [("D0 3->32 V25->11", 0, True, 3, 32, 64), ("D1 32->64 V11", 1, False, 32, 64, 64), ("D2 64->128 V11->5", 1, True, 64, 128, 64),
         ("D3 128->256 V5", 2, False, 128, 256, 32)]
if __name__ != "__main__":
    CASES = []
for N in (64, 128, 192):
    for name, lvl, dws, cin, cout, T in CASES:
        A = np.asarray(g.As[lvl], dtype=np.float32)
        if dws: A = A[:, :, np.asarray(g.map[lvl + 1][:, 1])]
        K, V, W = A.shape
        nbr, pc = table(A)
        At = torch.as_tensor(np.ascontiguousarray(A), device=dev)
        x = nv.new_plane(N, cin, T, V, dev).normal_()
        w = torch.randn(3 * cout, cin, 1, 1, device=dev)
        wv = WView(cout * cin, cin, 1)
        def unfused():
            xa = nv.agg_expand(x, At, 1)
            return nv.conv([Group(xa, w, wv, cin, 3, TAP_CHANBLOCK, 1, False, None)], N, cout, T, W)
        def fused(): return nv.aggconv(x, At, nbr, pc, w, wv, cout)[0]
        def fused_xa(): return nv.aggconv(x, At, nbr, pc, w, wv, cout, want_xa=True)[0]
        err = (unfused() - fused()).abs().max().item()
        tu, tf, tx = timeit(unfused), timeit(fused), timeit(fused_xa)
        fl = 2.0 * N * T * W * cout * 3 * cin
        plans = ""
        if os.environ.get("KG_AGGCONV_SWEEP"):          # forced tile plans "<BM><KS>" (fused + xa)
            for pl in ("321", "641", "322", "642"):
                os.environ["KG_AGGCONV_PLAN"] = pl; nv.reload_env()
                try:
                    plans += "  %s %.1f" % (pl, timeit(fused_xa))
                except RuntimeError:
                    plans += "  %s -" % pl
            os.environ.pop("KG_AGGCONV_PLAN"); nv.reload_env()
        print("N=%3d %-20s expand+conv %6.1f us | fused %6.1f us (%5.1f TF) | fused+xa %6.1f us   maxdiff %.1e%s" % (N, name, tu, tf, fl / tf / 1e6, tx, err, plans), flush=True)
