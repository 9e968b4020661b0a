#!/usr/bin/env python3
"""Which Python lines launch the small aten kernels of one G+D iteration (eager mode, GPU box only)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kinetic_gan_amd.wgan_gp import Trainer

dev = torch.device("cuda:0")
cfg = bench.CONFIGS["ntu"] if hasattr(bench, "CONFIGS") else None
G, D = bench.build_models(cfg, dev)
tr = Trainer(G, D)
batch = bench.synth_batch(cfg, 64, 0, dev)
step, mode = bench.make_step(tr, batch, use_graph=False, segmented=False)
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::fill_", "aten::zero_", "aten::sub", "aten::div",
        "aten::neg", "aten::sum", "aten::mean", "aten::cat", "aten::index", "aten::clone", "aten::contiguous", "aten::zeros_like",
        "aten::ones_like", "aten::empty_like")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        st = [s for s in ev.stack if "/root/repo" in s or "kinetic" in s or "bench.py" in s]
        key = (ev.name, st[0].split("/")[-1] if st else "<autograd engine / no python frame>")
        cnt[key] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:5d}  {name:18s} {where}")
