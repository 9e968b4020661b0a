"""Per-workgroup durations of the critic step's two kg_wgrad_many launches, by job (debug build: tools/build_variant.sh
wg_debug "-DKG_WG_DEBUG ..." kg_wgrad.hip; KG_LIB=build_ab/libkgan_wg_debug.so python tools/wg_times.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from kinetic_gan_amd import _native as nv

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["ntu"]
G, D = bench.build_models(cfg, dev)
from kinetic_gan_amd.wgan_gp import Trainer
tr = Trainer(G, D, world_size=1)
batch = bench.synth_batch(cfg, 64, 0, dev)
lib = nv.load_library()
calls = []
orig = nv.wgrad_many


def spy(jobs):
    torch.cuda.synchronize()
    lib.kg_wgrad_debug_clear()
    orig(jobs)
    torch.cuda.synchronize()
    buf = np.zeros(3 * 16384, dtype=np.uint64)
    lib.kg_wgrad_debug_times(buf.ctypes.data_as(ctypes.c_void_p), 16384)
    calls.append((len(jobs), buf.reshape(16384, 3).copy(), [(j['g'].shape[1], j['Cin'], j['taps'], tuple(j['g'].shape)) for j in jobs]))


step, _ = bench.make_step(tr, batch, False, False)
for _ in range(2):
    step()
torch.cuda.synchronize()
nv.wgrad_many = spy
step()
torch.cuda.synchronize()
for njobs, t, descr in calls:
    print(f"--- wgrad_many call with {njobs} jobs")
    for bank in (0, 1):
        b = t[bank * 8192:(bank + 1) * 8192]
        live = b[:, 1] > 0
        if not live.any():
            continue
        b = b[live]
        t0 = b[:, 0].min()
        dur = (b[:, 1] - b[:, 0]).astype(np.float64) / 100.0        # s_memtime: 100 MHz
        print(f"  bank {bank}: {len(b)} workgroups, span {(b[:, 1].max() - t0) / 100.0:.1f} us")
        for ji in sorted(set(b[:, 2].tolist())):
            m = b[:, 2] == ji
            d = dur[m]
            st = (b[m, 0] - t0).astype(np.float64) / 100.0
            print(f"    job {int(ji):2d} {str(descr[int(ji)]) if int(ji) < len(descr) else '':34s}: {m.sum():4d} wgs  dur mean {d.mean():6.1f} min {d.min():6.1f} max {d.max():6.1f}   start {st.min():6.1f}..{st.max():6.1f}  end max {(st + d).max():6.1f}  share {d.sum() / dur.sum():.3f}")
