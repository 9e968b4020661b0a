import os, sys
sys.path.insert(0, "/root/repo")
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
from tools.time_aggconv import timeit
dev = torch.device("cuda:0")
N, C, T, V, W, K = 64, 3, 64, 25, 25, 3
x = nv.new_plane(N, C, T, V, dev).normal_()
y = nv.new_plane(N, K * C, T, V, dev).normal_()
A = torch.randn(K, V, W, device=dev)
for name, fn in (("expand", lambda: nv.agg_expand(x, A, 1)), ("reduce", lambda: nv.agg_reduce(y, A, 1))):
    res = []
    for env in ({}, {"KG_AGG_MFMA": "1"}, {"KG_AGG_STREAM": "1", "KG_AGG_MFMA": "0"}):
        for k in ("KG_AGG_MFMA", "KG_AGG_STREAM"):
            os.environ.pop(k, None)
        os.environ.update(env); nv.reload_env()
        res.append("%s %.1f us" % (env or "auto", timeit(fn)))
    print(name, " | ".join(res), flush=True)
