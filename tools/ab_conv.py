#!/usr/bin/env python3
"""A/B of kg_conv switches at the discriminator's shapes (GPU box): KG_AB="KG_CONV_FAST=0" compares the default
library configuration with the given environment setting, automatic plan, at N = 64 and N = 192 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, TAP_CHANBLOCK, Group, WView

dev = torch.device("cuda:0")

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
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3

def gcn(N, cin, cout, T, W):
    xa = nv.new_plane(N, 3 * cin, T, W, dev).normal_()
    w = torch.randn(3 * cout, cin, 1, 1, device=dev)
    g = Group(xa, w, WView(cout * cin, cin, 1), cin, 3, TAP_CHANBLOCK, 1, False, None)
    return (lambda: nv.conv([g], N, cout, T, W)), 2.0 * N * T * W * cout * 3 * cin

def gcnT(N, cin, cout, T, W):
    gz = nv.new_plane(N, cout, T, W, dev).normal_()
    w = torch.randn(3 * cout, cin, 1, 1, device=dev)
    g = Group(gz, w, WView(0, 1, cin, cout * cin, cin), cout, 1)
    return (lambda: nv.conv([g], N, 3 * cin, T, W)), 2.0 * N * T * W * cout * 3 * cin

def tail(N, cin, cout, T, V, W, s, res=True):
    z = nv.new_plane(N, cout, T, W, dev).normal_()
    x = nv.new_plane(N, cin, T, V, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev); wr = torch.randn(cout, cin, 1, 1, device=dev)
    keep = torch.arange(W, dtype=torch.int32, device=dev) if W != V else None      # (as the trunk: no vertex map without down-sampling)
    gs = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, s, False, None)]
    fl = 3 * cout * cout
    if res:
        gs.append(Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, s, False, keep)); fl += cin * cout
    return (lambda: nv.conv(gs, N, cout, T // s, W, act=nv.ACT_LRELU)), 2.0 * N * (T // s) * W * fl

def tailT(N, cout, T, W, s):
    g = nv.new_plane(N, cout, T // s, W, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev)
    gr = Group(g, wt, WView(1, 3, cout * 3), cout, 3, TAP_TIME, s, True, None)
    return (lambda: nv.conv([gr], N, cout, T, W)), 2.0 * N * (T // s) * W * 3 * cout * cout

def cases(N):
    return {
        "D1 gcn 32->64": gcn(N, 32, 64, 64, 11),
        "D1 tail 64 (s1)": tail(N, 32, 64, 64, 11, 11, 1),
        "D1 tailT 64 (s1)": tailT(N, 64, 64, 11, 1),
        "D1 gcnT 32<-64": gcnT(N, 32, 64, 64, 11),
        "D2 gcn 64->128 (W5)": gcn(N, 64, 128, 64, 5),
        "D2 tail 128 (s2)": tail(N, 64, 128, 64, 11, 5, 2),
        "D2 gcnT 64<-128": gcnT(N, 64, 128, 64, 5),
        "D3 gcn 128->256": gcn(N, 128, 256, 32, 5),
        "D3 tail 256 (s2)": tail(N, 128, 256, 32, 5, 5, 2),
        "D3 gcnT 128<-256": gcnT(N, 128, 256, 32, 5),
        "D4 gcn 256->512 (W1)": gcn(N, 256, 512, 16, 1),
        "D4 tail 512 (s2)": tail(N, 256, 512, 16, 5, 1, 2),
        "D5 tail 512 (s2, no res conv)": tail(N, 512, 512, 8, 1, 1, 2, res=False),
    }

ab = os.environ.get("KG_AB", "KG_CONV_FAST=0")
k, v = ab.split("=")
for N in (64, 192):
    tot_a = tot_b = 0.0
    for name, (fn, flops) in cases(N).items():
        os.environ.pop(k, None); nv.reload_env()
        ref = fn().clone()
        a = timeit(fn)
        os.environ[k] = v; nv.reload_env()
        out = fn()
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        b = timeit(fn)
        os.environ.pop(k, None); nv.reload_env()
        tot_a += a; tot_b += b
        print(f"N={N:3d} {name:32s} default {a:7.1f} us {flops/a/1e6:6.1f} TF | {ab} {b:7.1f} us {flops/b/1e6:6.1f} TF | ratio {b/a:.3f} err {err:.1e}", flush=True)
    print(f"N={N:3d} total default {tot_a:.1f} us, {ab} {tot_b:.1f} us, ratio {tot_b/tot_a:.3f}", flush=True)
