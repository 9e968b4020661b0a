#!/usr/bin/env python3
"""Times the mapping-network kernels (kg_linear_fwd / kg_linear_bwd / kg_embed_bwd) at the NTU-60 sizes next to the stock ops
they replace (hipGraph replay of 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_conv.py")).read().split("def gcn(")[0]
ns = {"__name__": "x", "__file__": __file__}
exec(compile(src, "ab_conv.py", "exec"), ns)
timeit = ns["timeit"]
d = torch.device("cuda:0")
for N, D, L in ((64, 512, 60), (32, 512, 120)):
    J = L; Din = D + J
    x = torch.randn(N, D, device=d); w = torch.randn(Din, Din, device=d) * 0.1; b = torch.randn(Din, device=d)
    emb = torch.randn(L, J, device=d); labels = torch.randint(0, L, (N,), device=d)
    y = nv.linear_fwd(x, w, b, nv.ACT_LRELU, 0.2, emb=emb, labels=labels)
    g = torch.randn(N, Din, device=d)
    dw = torch.zeros(Din, Din, device=d); db = torch.zeros(Din, device=d); demb = torch.zeros(L, J, device=d)
    xin = torch.cat((emb[labels], x), 1)
    print(f"N={N} Din={Din}")
    print("  kg_linear_fwd (first layer)   %6.2f us" % timeit(lambda: nv.linear_fwd(x, w, b, nv.ACT_LRELU, 0.2, emb=emb, labels=labels)))
    print("  kg_linear_fwd (inner layer)   %6.2f us" % timeit(lambda: nv.linear_fwd(y, w, b, nv.ACT_LRELU, 0.2)))
    print("  stock linear + leaky_relu     %6.2f us" % timeit(lambda: torch.nn.functional.leaky_relu(torch.nn.functional.linear(xin, w, b), 0.2)))
    print("  kg_linear_bwd (inner, all)    %6.2f us" % timeit(lambda: nv.linear_bwd(g, y, y, w, nv.ACT_LRELU, 0.2, dw=dw, db=db, accumulate=True)))
    print("  kg_linear_bwd (first, J cols) %6.2f us" % timeit(lambda: nv.linear_bwd(g, y, x, w, nv.ACT_LRELU, 0.2, emb=emb, labels=labels, gx_cols=J, dw=dw, db=db, accumulate=True)))
    print("  kg_embed_bwd                  %6.2f us" % timeit(lambda: nv.embed_bwd(g, labels, demb, accumulate=True)))
    def stock_bwd():
        gp = torch.ops.aten.leaky_relu_backward(g, y, 0.2, True)
        gx = gp @ w
        dw.addmm_(gp.t(), xin)
        db.add_(gp.sum(0))
        return gx
    print("  stock backward of one layer   %6.2f us" % timeit(stock_bwd))
