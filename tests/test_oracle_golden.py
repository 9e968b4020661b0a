"""Pin the oracle (oracle/modules_ref.py) against outputs of the reference itself
(tests/golden/ref_*.npz, produced by tests/golden/make_fixtures.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import modules_ref as M
from oracle.fill import (block_input, disc_block_in_shapes, fill_module, gen_block_in_shapes,
                         rand_inputs, rand_noise)

CFG = {"ntu": dict(channels=3, n_classes=60, t_size=64, latent=512, mlp=4),
       "h36m": dict(channels=2, n_classes=10, t_size=32, latent=512, mlp=4)}
TOL = dict(rtol=1e-5, atol=2e-6)


def build(ds):
    c = CFG[ds]
    G = M.Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=ds)
    D = M.Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=ds)
    fill_module(G, seed=1)
    fill_module(D, seed=2)
    return c, G, D


@pytest.fixture(scope="module", params=["ntu", "h36m"])
def setup(request, golden_dir):
    ds = request.param
    return (ds, np.load(os.path.join(golden_dir, f"ref_{ds}.npz")), *build(ds))


def test_blocks(setup):
    ds, gold, c, G, D = setup
    nn_ = G.graph.num_node
    n = 2
    noise = rand_noise(n, c["t_size"], nn_, seed=5)
    gs = gen_block_in_shapes(n, c["latent"] + c["n_classes"], c["channels"], c["t_size"], nn_)
    for mode in ("train", "eval"):
        G.train(mode == "train")
        fill_module(G, seed=1)
        for i, (blk, imp) in enumerate(zip(G.st_gcn_networks, G.edge_importance)):
            y, _ = blk(block_input(gs[i], 200 + i), G.A[blk.lvl] * imp, noise[i])
            np.testing.assert_allclose(y.detach().numpy(), gold[f"G{i}_{mode}"], **TOL)
        if mode == "train":
            for k, v in G.state_dict().items():
                if "running_" in k:
                    np.testing.assert_allclose(v.numpy(), gold["Gstat_" + k], **TOL)
    dsh = disc_block_in_shapes(n, c["channels"] + c["n_classes"], c["latent"], c["t_size"], nn_)
    for i, (blk, imp) in enumerate(zip(D.st_gcn_networks, D.edge_importance)):
        y, _ = blk(block_input(dsh[i], 400 + i), D.A[blk.lvl] * imp)
        np.testing.assert_allclose(y.detach().numpy(), gold[f"D{i}"], **TOL)


def test_models_and_step(setup):
    ds, gold, c, G, D = setup
    nn_ = G.graph.num_node
    n = 4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(n, c["t_size"], nn_, seed=6)
    for mode in ("train", "eval"):
        G.train(mode == "train")
        fill_module(G, seed=1)
        np.testing.assert_allclose(G(z, labels, noise=noise).detach().numpy(), gold[f"G_out_{mode}"], **TOL)
    np.random.seed(77)          # generate.py:90-93 inference path: eval + W-space truncation, latents pinned by seed
    with torch.no_grad():
        np.testing.assert_allclose(G(z, labels, trunc=0.7, noise=noise).numpy(), gold["G_out_eval_trunc"], **TOL)
    G.train(True)
    fill_module(G, seed=1)
    np.testing.assert_allclose(D(real, labels).detach().numpy(), gold["D_out"], **TOL)

    r = M.d_step_losses(G, D, real, labels, z, alpha, noise=noise)
    D.zero_grad()
    r["d_loss"].backward()
    for k in ("real_validity", "fake_validity", "gradient_penalty", "d_loss"):
        np.testing.assert_allclose(r[k].detach().numpy(), gold[k], rtol=2e-5, atol=2e-6)
    for k, p in D.named_parameters():
        g = p.grad.reshape(-1)
        assert abs(g.double().norm().item() - float(gold["Dgn_" + k])) <= 1e-4 * float(gold["Dgn_" + k]) + 2e-6, k
        s = g[:: max(1, g.numel() // 64)][:64].numpy()
        np.testing.assert_allclose(s, gold["Dgs_" + k], rtol=1e-3, atol=1e-5 * (np.abs(gold["Dgs_" + k]).max() + 1e-6) + 1e-7)

    fill_module(G, seed=1)
    G.zero_grad()
    r = M.g_step_loss(G, D, labels, z, noise=noise)
    r["g_loss"].backward()
    np.testing.assert_allclose(r["g_loss"].detach().numpy(), gold["g_loss"], rtol=2e-5, atol=2e-6)
    for k, p in G.named_parameters():
        g = p.grad.reshape(-1)
        assert abs(g.double().norm().item() - float(gold["Ggn_" + k])) <= 1e-4 * float(gold["Ggn_" + k]) + 2e-6, k
