#!/usr/bin/env python3
"""Where does the generator's forward error against the fp32 oracle come from (round-5 VERDICT weak 1: fake rel_err 3e-5 ..
8e-5 of the 1e-4 budget)?  Per block, at the configs of tests/test_parity_gpu.py::test_full_size_vs_oracle:
   HIP path (trunk) vs the oracle in float64  |  the fp32 oracle (host) vs the oracle in float64  |  HIP vs fp32 oracle
rel_err = max|a - b| / max|b|.  If the first two columns are alike the distance to the fp32 oracle is the two fp32 evaluations'
own round-off, amplified by the network (BatchNorm over small batches), not a summation-order defect of one kernel."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair  # noqa: E402
from oracle.fill import rand_inputs, rand_noise  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


d = torch.device("cuda:0")
for cfg, n in (("ntu", 64), ("ntu120", 32), ("h36m", 64), ("stress", 2)):
    c, G, D, Go, Do = build_pair(cfg, d)
    nn_ = G.graph.num_node
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=11)
    noise = rand_noise(n, c["t_size"], nn_, seed=12)
    Go64 = copy.deepcopy(Go).double()
    Go64.A = [a.double() for a in Go64.A]
    outs = {"hip": [], "o32": [], "o64": []}

    def hook(key):
        return lambda mod, inp, out: outs[key].append(out[0].detach())
    hs = [blk.register_forward_hook(hook("o32")) for blk in Go.st_gcn_networks] + \
         [blk.register_forward_hook(hook("o64")) for blk in Go64.st_gcn_networks]
    with torch.no_grad():
        f32 = Go(z, labels, noise=noise)
        f64 = Go64(z.double(), labels, noise=[t.double() for t in noise])
    for h in hs:
        h.remove()
    rows = {}
    for trunk in (True, False):
        G.use_trunk = trunk
        if not trunk:
            hs = [blk.register_forward_hook(hook("hip")) for blk in G.st_gcn_networks]
        with torch.no_grad():
            fh = G(z.to(d), labels.to(d), noise=[t.to(d) for t in noise])
        rows[trunk] = (rel(fh, f64), rel(fh, f32))
    print("%s n=%d: fake  HIP(trunk) vs f64 %.2e | HIP(blockwise) vs f64 %.2e | oracle fp32 vs f64 %.2e | HIP(trunk) vs oracle fp32 %.2e" % (
        cfg, n, rows[True][0], rows[False][0], rel(f32, f64), rows[True][1]))
    for i, (a, b, c64) in enumerate(zip(outs["hip"], outs["o32"], outs["o64"])):
        print("    G%d out %-18s HIP vs f64 %.2e | oracle fp32 vs f64 %.2e | HIP vs oracle fp32 %.2e" % (
            i, tuple(b.shape), rel(a, c64), rel(b, c64), rel(a, b)))
