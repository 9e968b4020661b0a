#!/usr/bin/env python3
"""Time kg_conv at the discriminator's bs=64 shapes under forced (tile, nsplit) plans (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
if os.environ.get("KG_LIB"):                       # experiment builds (tools/gpu_ab.sh)
    nv.LIB_PATH = os.environ["KG_LIB"]
from kinetic_gan_amd._native import TAP_TIME, TAP_CHANBLOCK, Group, WView

dev = torch.device("cuda:0")
TILES = {0: "128x128", 1: "64x128", 2: "32x128", 3: "64x64", 4: "32x64", 9: "K32x32"}

def timeit(fn, reps=20):
    """GPU time per call: the calls are captured in a hipGraph so host launch overhead is not measured."""
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

def gcn(N, cin, cout, T, W):
    xa = nv.new_plane(N, 3 * cin, T, W, dev).normal_()
    w = torch.randn(3 * cout, cin, 1, 1, device=dev)
    g = Group(xa, w, WView(cout * cin, cin, 1), cin, 3, TAP_CHANBLOCK, 1, False, None)
    return (lambda: nv.conv([g], N, cout, T, W)), 2.0 * N * T * W * cout * 3 * cin

def tail(N, cin, cout, T, V, W, s, res=True):
    z = nv.new_plane(N, cout, T, W, dev).normal_()
    x = nv.new_plane(N, cin, T, V, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev); wr = torch.randn(cout, cin, 1, 1, device=dev)
    keep = torch.arange(W, dtype=torch.int32, device=dev)
    gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, s, False, None)]
    fl = 3 * cout * cout
    if res:
        gs.append(Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, s, False, keep)); fl += cin * cout
    return (lambda: nv.conv(gs, N, cout, T // s, W, act=nv.ACT_LRELU)), 2.0 * N * (T // s) * W * fl

def gcnT(N, cin, cout, T, W):
    gz = nv.new_plane(N, cout, T, W, dev).normal_()
    w = torch.randn(3 * cout, cin, 1, 1, device=dev)
    g = Group(gz, w, WView(0, 1, cin, cout * cin, cin), cout, 1)
    return (lambda: nv.conv([g], N, 3 * cin, T, W)), 2.0 * N * T * W * cout * 3 * cin

def tailT(N, cout, T, W, s):
    g = nv.new_plane(N, cout, T // s, W, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev)
    gr = Group(g, wt, WView(1, 3, cout * 3), cout, 3, TAP_TIME, s, True, None)
    return (lambda: nv.conv([gr], N, cout, T, W)), 2.0 * N * (T // s) * W * 3 * cout * cout

N = int(os.environ.get("N", "64"))
CASES = {
    "D0 gcn 63->32 (cols N*64*11)": gcn(N, 63, 32, 64, 11),
    "D1 gcn 32->64": gcn(N, 32, 64, 64, 11),
    "D1 tail 64 (s1)": tail(N, 32, 64, 64, 11, 11, 1),
    "D2 gcn 64->128 (W5)": gcn(N, 64, 128, 64, 5),
    "D2 tail 128 (s2)": tail(N, 64, 128, 64, 11, 5, 2),
    "D2 tailT 128 (s2)": tailT(N, 128, 64, 5, 2),
    "D3 gcn 128->256": gcn(N, 128, 256, 32, 5),
    "D3 tail 256 (s2)": tail(N, 128, 256, 32, 5, 5, 2),
    "D4 gcn 256->512 (W1)": gcn(N, 256, 512, 16, 1),
    "D4 tail 512 (s2)": tail(N, 256, 512, 16, 5, 1, 2),
    "D5 gcn 512->512": gcn(N, 512, 512, 8, 1),
    "D5 tail 512 (s2, no res conv)": tail(N, 512, 512, 8, 1, 1, 2, res=False),
    "D1 tailT 64 (s1)": tailT(N, 64, 64, 11, 1),
    "D1 gcnT 32<-64": gcnT(N, 32, 64, 64, 11),
    "D2 gcnT 64<-128": gcnT(N, 64, 128, 64, 5),
    "D3 gcnT 128<-256": gcnT(N, 128, 256, 32, 5),
    "D3 tailT 256 (s2) even frames": tailT(N, 256, 32, 5, 2),
    "D4 gcnT 256<-512": gcnT(N, 256, 512, 16, 1),
}
QUICK = os.environ.get("KG_TUNE_QUICK")                    # only the automatic plan + a few forced ones
for name, (fn, flops) in CASES.items():
    os.environ.pop("KG_CONV_PLAN", None); nv.reload_env()
    base = timeit(fn)
    best = (base, "auto")
    row = []
    ref = fn().clone()
    for t in TILES:
        if QUICK and TILES[t] not in QUICK.split(","):
            continue
        for ns in (1, 2, 4, 8, 16):
            os.environ["KG_CONV_PLAN"] = f"{t},{ns}"; nv.reload_env()
            try:
                out = fn()
                err = ((out - ref).abs().max() / ref.abs().max()).item()
                if not err < 2e-5:
                    print(f"   !! {name} plan {TILES[t]}/k{ns}: rel err {err:.3e}", flush=True)
                us = timeit(fn, 10)
            except RuntimeError as e:
                continue
            row.append((us, f"{TILES[t]}/k{ns}"))
    row.sort()
    print(f"{name:34s} auto {base:7.1f} us {flops/base/1e6:6.1f} TF | best: " + "  ".join(f"{n} {u:.1f}" for u, n in row[:6]), flush=True)
