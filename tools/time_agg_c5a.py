#!/usr/bin/env python3
"""the standalone aggregation (64, 3*512, 256, 25) -> (64, 512, 256, 25) and its expand twin (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
dev = torch.device("cuda:0")
N, C, T, V, K = int(os.environ.get("N", "64")), 512, 256, 25, 3
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
A = torch.rand(K, V, V, device=dev)
y = nv.new_plane(N, K * C, T, V, dev).normal_()
x = nv.new_plane(N, C, T, V, dev).normal_()
gb = 4.0 * (K + 1) * C * T * V * N / 1e9          # 4*(K+1)*C*T*V bytes per sample
for mode in ("0", "1", "mfma", None):
    os.environ["KG_AGG_MFMA"] = "0"; nv.reload_env()
    if mode is None:
        os.environ.pop("KG_AGG_STREAM", None); os.environ.pop("KG_AGG_MFMA", None); nv.reload_env()
    elif mode == "mfma":
        os.environ["KG_AGG_MFMA"] = "1"; nv.reload_env()
    else: os.environ["KG_AGG_STREAM"] = mode; nv.reload_env()
    tr = timeit(lambda: nv.agg_reduce(y, A, 1))
    te = timeit(lambda: nv.agg_expand(x, A, 1))
    name = {"0": "frame-per-thread", "1": "stream", "mfma": "matrix cores", None: "auto"}[mode]
    print(f"{name:17s} reduce {tr:6.2f} ms  {gb / tr:6.2f} TB/s   expand {te:6.2f} ms  {gb / te:6.2f} TB/s   ({gb:.2f} GB algorithmic)", flush=True)
