#!/usr/bin/env python3
"""Time kg_agg_outer (MFMA / element-wise kernels) at the discriminator's shapes (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv

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
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3

if __name__ == "__main__":
    N = int(os.environ.get("N", "128"))
    for name, C, T, V, W in [("D0", 3, 64, 25, 11), ("D1", 32, 64, 11, 11), ("D2", 64, 64, 11, 5), ("D3", 128, 32, 5, 5),
                             ("D4", 256, 16, 5, 1), ("D5", 512, 8, 1, 1)]:
        x = nv.new_plane(N, C, T, V, dev).normal_()
        y = nv.new_plane(N, 3 * C, T, W, dev).normal_()
        fn = lambda: nv.agg_outer(x, y, 3, 1)
        mb = (x.numel() + y.numel()) * 4 / 1e6
        os.environ["KG_AGG_OUTER_MFMA"] = "0"; nv.reload_env()
        t0 = timeit(fn)
        os.environ.pop("KG_AGG_OUTER_MFMA"); nv.reload_env()
        t1 = timeit(fn)
        print(f"{name} C={C:3d} T={T:2d} V={V:2d} W={W:2d}  outer: {mb:6.1f} MB  element-wise {t0:6.1f} us   mfma {t1:6.1f} us  ({mb / t1:.2f} TB/s)", flush=True)
        A = torch.randn(3, V, W, device=dev)
        At = torch.randn(3, W, V, device=dev)
        row = []
        os.environ["KG_AGG_MFMA"] = "0"; nv.reload_env()
        for env in ("0", "1", "mfma"):
            if env == "mfma": os.environ["KG_AGG_MFMA"] = "1"; nv.reload_env()
            else: os.environ["KG_AGG_STREAM"] = env; nv.reload_env()
            te = timeit(lambda: nv.agg_expand(x, A, 1))
            tr_ = timeit(lambda: nv.agg_reduce(y, At, 1))
            os.environ.pop("KG_AGG_STREAM", None); nv.reload_env()
            row.append(f"{ {'0': 'frame-per-thread', '1': 'stream', 'mfma': 'mfma'}[env] }: expand {te:5.1f} us ({mb / te:.2f} TB/s)  reduce {tr_:5.1f} us ({mb / tr_:.2f} TB/s)")
        os.environ.pop("KG_AGG_MFMA", None); nv.reload_env()
        te = timeit(lambda: nv.agg_expand(x, A, 1)); tr_ = timeit(lambda: nv.agg_reduce(y, At, 1))
        row.append(f"auto: expand {te:5.1f}  reduce {tr_:5.1f}")
        print("      " + "   ".join(row), flush=True)

