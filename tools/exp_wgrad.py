"""kg_wgrad_many in isolation on four discriminator layers (critic step: operand pairs of 128 + 64 samples), hipGraph
replay, for the library named by KG_LIB.  python tools/exp_wgrad.py -> one line per layer: us per launch (+reduce), TF/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_TIME, WView

dev = torch.device("cuda", 0)
#         name        Cin  M    T_in V   stride taps
LAYERS = [("D1 tcn",  64,  64,  64,  11, 1, 3),
          ("D2 tcn", 128, 128,  64,  5,  2, 3),
          ("D3 tcn", 256, 256,  32,  5,  2, 3),
          ("D4 tcn", 512, 512,  16,  1,  2, 3),
          ("D2 gcn", 64, 384,   64,  5,  1, 1),
          ("D3 gcn", 128, 768,  32,  5,  1, 1)]
for name, cin, m, T, V, s, taps in LAYERS:
    prs = [(nv.new_plane(n, m, T // s, V, dev).normal_(), nv.new_plane(n, cin, T, V, dev).normal_()) for n in (128, 64)]
    out = torch.zeros(m * cin * taps, device=dev)
    job = dict(g=prs[0][0], x=prs[0][1], Cin=cin, taps=taps, tap_mode=TAP_TIME, t_stride=s, vmap=None,
               wv=WView(1, cin * taps, taps), out=out, accumulate=True, extra=[prs[1]])
    ms = bench._time_launch(lambda: nv.wgrad_many([job]), reps=20)
    fl = 2.0 * taps * m * cin * (192 * (T // s) * V)
    print(f"{name:8s} {ms * 1e3:8.2f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
