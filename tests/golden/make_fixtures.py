#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the dev container (needs /root/reference); nothing here travels to the GPU box
except the .npz/.json it writes.  The reference's ``models`` package is imported under the two
CPU shims SURVEY.md 8(c) describes (``Tensor.cuda`` -> identity, ``torch.randn`` drops
``device=`` and pops pre-made noise), its parameters are filled by ``oracle.fill`` (weights are
never committed), and outputs for seeded inputs are stored.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.fill import (fill_module, rand_inputs, rand_noise, block_input,  # noqa: E402
                         gen_block_in_shapes, disc_block_in_shapes)

torch.set_num_threads(8)

# ---- shims -----------------------------------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
_real_randn = torch.randn
_noise_queue = []


def _randn(*size, **kw):
    kw.pop("device", None)
    if _noise_queue:
        t = _noise_queue.pop(0)
        assert tuple(t.shape) == tuple(size), (t.shape, size)
        return t
    return _real_randn(*size, **kw)


torch.randn = _randn
torch.cuda.FloatTensor = torch.FloatTensor          # generator.py:98 builds the truncation latents with it

from models.generator import Generator  # noqa: E402
from models.discriminator import Discriminator  # noqa: E402
from models.init_gan.graph_ntu import graph_ntu  # noqa: E402
from models.init_gan.graph_h36m import Graph_h36m  # noqa: E402

CFG = {
    "ntu": dict(channels=3, n_classes=60, t_size=64, latent=512, mlp=4),
    "h36m": dict(channels=2, n_classes=10, t_size=32, latent=512, mlp=4),
}


def tables(g):
    return {
        "num_node": [int(x) for x in g.num_node],
        "center": [int(x) for x in g.center],
        "map": [np.asarray(m).tolist() for m in g.map],
        "edge": [np.asarray(e).tolist() for e in g.edge],
        "mapping": [[np.asarray(h).tolist() for h in lv] for lv in g.mapping],
        "As": [np.asarray(a).tolist() for a in g.As],
    }


def main():
    json.dump({"ntu": tables(graph_ntu()), "h36m": tables(Graph_h36m())},
              open(os.path.join(HERE, "graph_tables.json"), "w"))

    for ds, c in CFG.items():
        torch.manual_seed(0)
        G = Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=ds)
        D = Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=ds)
        fill_module(G, seed=1)
        fill_module(D, seed=2)
        nn_ = G.graph.num_node
        out = {}

        # ---- per-block outputs, N=2, train mode (and eval mode for G) ---------------------
        n = 2
        noise = rand_noise(n, c["t_size"], nn_, seed=5)
        g_shapes = gen_block_in_shapes(n, c["latent"] + c["n_classes"], c["channels"], c["t_size"], nn_)
        for mode in ("train", "eval"):
            G.train(mode == "train")
            fill_module(G, seed=1)          # reset BN running stats
            for i, (blk, imp) in enumerate(zip(G.st_gcn_networks, G.edge_importance)):
                xin = block_input(g_shapes[i], 200 + i)
                _noise_queue.append(noise[i])
                y, _ = blk(xin, G.A[blk.lvl] * imp)
                out[f"G{i}_{mode}"] = y.detach().numpy()
            if mode == "train":
                for k, v in G.state_dict().items():
                    if "running_" in k:
                        out["Gstat_" + k] = v.numpy().copy()
        G.train(True)
        fill_module(G, seed=1)

        d_shapes = disc_block_in_shapes(n, c["channels"] + c["n_classes"], c["latent"], c["t_size"], nn_)
        for i, (blk, imp) in enumerate(zip(D.st_gcn_networks, D.edge_importance)):
            xin = block_input(d_shapes[i], 400 + i)
            y, _ = blk(xin, D.A[blk.lvl] * imp)
            out[f"D{i}"] = y.detach().numpy()

        # ---- whole models, N=4 --------------------------------------------------------------
        n = 4
        real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
        noise = rand_noise(n, c["t_size"], nn_, seed=6)
        for mode in ("train", "eval"):
            G.train(mode == "train")
            fill_module(G, seed=1)
            _noise_queue.extend(noise)
            out[f"G_out_{mode}"] = G(z, labels).detach().numpy()
        # ---- inference path of generate.py:90-93: eval mode + W-space truncation (generator.py:86,97-108);
        # the truncation latents come from numpy's global generator, pinned here by its seed
        G.train(False)
        fill_module(G, seed=1)
        _noise_queue.extend(noise)
        np.random.seed(77)
        with torch.no_grad():
            out["G_out_eval_trunc"] = G(z.clone(), labels, trunc=0.7).detach().numpy()
        G.train(True)
        fill_module(G, seed=1)
        out["D_out"] = D(real, labels).detach().numpy()

        # ---- one WGAN-GP iteration's scalars and gradients (kinetic-gan.py:137-174) ----------
        _noise_queue.extend(noise)
        fake = G(z, labels)
        real_v = D(real, labels)
        fake_v = D(fake, labels)
        inter = (alpha * real.data + (1 - alpha) * fake.data).requires_grad_(True)
        d_inter = D(inter, labels)
        grads = torch.autograd.grad(d_inter, inter, torch.ones(n, 1), create_graph=True,
                                    retain_graph=True, only_inputs=True)[0]
        out["gp_grads"] = grads.detach().numpy()
        gp = ((grads.reshape(n, -1).norm(2, dim=1) - 1) ** 2).mean()
        d_loss = -real_v.mean() + fake_v.mean() + 10 * gp
        D.zero_grad()
        G.zero_grad()
        d_loss.backward()
        out["real_validity"] = real_v.detach().numpy()
        out["fake_validity"] = fake_v.detach().numpy()
        out["gradient_penalty"] = gp.detach().numpy()
        out["d_loss"] = d_loss.detach().numpy()
        for k, p in D.named_parameters():
            gflat = p.grad.reshape(-1)
            out["Dgn_" + k] = np.float64(gflat.double().norm().item())
            out["Dgs_" + k] = gflat[:: max(1, gflat.numel() // 64)][:64].numpy().copy()

        fill_module(G, seed=1)
        G.zero_grad()
        D.zero_grad()
        _noise_queue.extend(noise)
        fake = G(z, labels)
        g_loss = -D(fake, labels).mean()
        g_loss.backward()
        out["g_loss"] = g_loss.detach().numpy()
        for k, p in G.named_parameters():
            gflat = p.grad.reshape(-1)
            out["Ggn_" + k] = np.float64(gflat.double().norm().item())
            out["Ggs_" + k] = gflat[:: max(1, gflat.numel() // 64)][:64].numpy().copy()

        assert not _noise_queue
        np.savez(os.path.join(HERE, f"ref_{ds}.npz"), **out)
        print(ds, "written", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) / 1e6, "MB")


if __name__ == "__main__":
    main()
