#!/usr/bin/env python3
"""Every aten op of one critic / generator step with the repo frame that issued it (GPU box; eager, autograd on the
calling thread so that backward-side ops are seen too)."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from kinetic_gan_amd.wgan_gp import Trainer

dev = torch.device("cuda:0")
cfg = bench.CONFIGS["ntu"]
G, D = bench.build_models(cfg, dev)
tr = Trainer(G, D)
real, labels, z, alpha = bench.synth_batch(cfg, 64, 0, dev)
which = os.environ.get("WHICH", "d")
run = (lambda: tr.d_step(real, labels, z, alpha, None)) if which == "d" else ((lambda: tr.g_step(labels, z, None)) if which == "g" else
       (lambda: tr.iteration(real, labels, z, alpha, None, None, with_g=True)))       # WHICH=it: the whole G+D iteration (bench path)
for _ in range(2): run()
torch.cuda.synchronize()
cnt = collections.Counter()

class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        st = traceback.extract_stack()
        fr = [f"{f.filename.split('/')[-1]}:{f.lineno}:{f.name}" for f in st if ROOT in f.filename and "dispatch_trace" not in f.filename]
        shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
        cnt[(name, fr[-1] if fr else "<engine>", shape if name.startswith(("copy_", "clone", "_to_copy")) else None)] += 1
        return func(*args, **(kwargs or {}))

torch.autograd.grad_mode.set_multithreading_enabled(False)
with Trace():
    run()
torch.cuda.synchronize()
skip = ("view", "detach", "alias", "t.default", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "_unsafe_view",
        "permute", "as_strided", "empty", "reshape", "split", "is_", "sym_", "_local_scalar", "lift", "unbind", "stride", "size")
tot = 0
for (name, where, shape), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    if name.startswith(skip): continue
    tot += c
    print(f"{c:4d} {name:32s} {where} {shape or ''}")
print("total (non-view) aten calls:", tot)
