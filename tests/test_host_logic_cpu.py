"""Host logic on CPU: the autograd composition in kinetic_gan_amd.ops / modules / wgan_gp run with
the native entry points swapped for their torch definitions (oracle/prim_ref.py) and compared with
the oracle modules (oracle/modules_ref.py, itself pinned to the reference by test_oracle_golden).

Gradient tolerances: LeakyReLU has a kink; an activation that sits within fp32 round-off of zero
may take the other slope in the two implementations and moves a gradient by O(1e-3) relative, so
gradients are compared by relative L2 error (forward values by max error, 1e-4 as BASELINE.json).
"""
import numpy as np
import pytest
import torch

from oracle import modules_ref as M
from oracle.fill import (block_input, disc_block_in_shapes, fill_module, gen_block_in_shapes,
                         rand_inputs, rand_noise)
from tests.util import CFG, build_pair, emulated_native, grad_close, grad_sample, l2_rel, rel_err

from kinetic_gan_amd import ops
from kinetic_gan_amd.wgan_gp import Trainer

FWD_TOL = 1e-4
GRAD_L2_TOL = 5e-3


@pytest.fixture(autouse=True)
def _emul():
    with emulated_native():
        yield


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_state_dict_keys_and_shapes(cfg):
    c, G, D, Go, Do = build_pair(cfg)
    for a, b in ((G, Go), (D, Do)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)
    assert "st_gcn_networks.1.tcn.0.weight" in G.state_dict() and "st_gcn_networks.1.tcn.weight" in D.state_dict()
    assert len(list(G.parameters())) == (70 if c["mlp"] == 4 else 78) and len(list(D.parameters())) == 35


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_blocks_forward_backward(cfg):
    c, G, D, Go, Do = build_pair(cfg)
    nn_ = G.graph.num_node
    n = 2
    noise = rand_noise(n, c["t_size"], nn_, seed=5)
    gs = gen_block_in_shapes(n, c["latent"] + c["n_classes"], c["channels"], c["t_size"], nn_)
    for mode in (True, False):
        G.train(mode)
        Go.train(mode)
        for i in range(7):
            b1, b2 = G.st_gcn_networks[i], Go.st_gcn_networks[i]
            x1 = block_input(gs[i], 200 + i).requires_grad_(True)
            x2 = x1.detach().clone().requires_grad_(True)
            y1, _ = b1(x1, G.A[b1.lvl] * G.edge_importance[i], noise[i])
            y2, _ = b2(x2, Go.A[b2.lvl] * Go.edge_importance[i], noise[i])
            assert rel_err(y1, y2) < FWD_TOL, (mode, i)
            go = torch.randn(y2.shape, generator=torch.Generator().manual_seed(9))
            G.zero_grad(); Go.zero_grad()
            y1.backward(go); y2.backward(go)
            assert grad_close(x1.grad, x2.grad, GRAD_L2_TOL), (mode, i)
            for (k, p), (_, q) in zip(G.named_parameters(), Go.named_parameters()):
                if mode and (k.endswith("residual.0.bias") or (k.endswith("tcn.0.bias") and len(b1.tcn) > 1)):
                    continue     # bias in front of a train-mode BN: gradient is analytically 0, pure round-off
                if q.grad is not None and q.grad.abs().max() > 0:
                    assert p.grad is not None, k
                    assert grad_close(p.grad, q.grad, GRAD_L2_TOL), (mode, i, k)
        if mode:
            for (k, v), (_, w) in zip(G.state_dict().items(), Go.state_dict().items()):
                if "running_" in k or "num_batches" in k:
                    np.testing.assert_allclose(v.numpy(), w.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    dsh = disc_block_in_shapes(n, c["channels"] + c["n_classes"], c["latent"], c["t_size"], nn_)
    for i in range(6):
        b1, b2 = D.st_gcn_networks[i], Do.st_gcn_networks[i]
        x1 = block_input(dsh[i], 400 + i).requires_grad_(True)
        x2 = x1.detach().clone().requires_grad_(True)
        y1, _ = b1(x1, D.A[b1.lvl] * D.edge_importance[i])
        y2, _ = b2(x2, Do.A[b2.lvl] * Do.edge_importance[i])
        assert rel_err(y1, y2) < FWD_TOL, i
        go = torch.randn(y2.shape, generator=torch.Generator().manual_seed(9))
        D.zero_grad(); Do.zero_grad()
        y1.backward(go); y2.backward(go)
        assert grad_close(x1.grad, x2.grad, GRAD_L2_TOL), i
        for (k, p), (_, q) in zip(b1.named_parameters(), b2.named_parameters()):
            assert grad_close(p.grad, q.grad, GRAD_L2_TOL), (i, k)
        assert grad_close(D.edge_importance[i].grad, Do.edge_importance[i].grad, GRAD_L2_TOL)


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_models_against_golden_reference_outputs(cfg, golden_dir):
    """Same inputs as tests/golden/make_fixtures.py -> compare with the REFERENCE's own outputs."""
    import os
    gold = np.load(os.path.join(golden_dir, f"ref_{cfg}.npz"))
    c, G, D, Go, Do = build_pair(cfg)
    nn_ = G.graph.num_node
    n = 4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(n, c["t_size"], nn_, seed=6)
    for mode in ("train", "eval"):
        G.train(mode == "train")
        fill_module(G, seed=1)
        out = G(z, labels, noise=noise)
        assert rel_err(out, torch.as_tensor(gold[f"G_out_{mode}"])) < FWD_TOL
    assert rel_err(D(real, labels), torch.as_tensor(gold["D_out"])) < FWD_TOL


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_wgan_gp_losses_and_double_backward(cfg, golden_dir):
    import os
    gold = np.load(os.path.join(golden_dir, f"ref_{cfg}.npz"))
    c, G, D, Go, Do = build_pair(cfg)
    G._pack_always = True       # the generator's packed-adjacency path (taken on the GPU) also on the host
    nn_ = G.graph.num_node
    n = 4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(n, c["t_size"], nn_, seed=6)
    tr = Trainer(G, D, flatten=False)
    r = tr.d_losses(real, labels, z, alpha, noise)
    D.zero_grad()
    r["d_loss"].backward()
    for k in ("real_validity", "fake_validity", "gradient_penalty", "d_loss"):
        assert rel_err(r[k], torch.as_tensor(gold[k])) < 2e-4, k
    # the penalty's gradient d D(inter) / d inter itself, element by element (fixture gp_grads)
    assert rel_err(r["gp_grads"], torch.as_tensor(gold["gp_grads"])) < 2e-4
    for k, p in D.named_parameters():
        ref_norm = float(gold["Dgn_" + k])
        assert abs(p.grad.double().norm().item() - ref_norm) <= GRAD_L2_TOL * ref_norm + 2e-6, k
        # ... and a strided 64-element sample of every gradient, element-wise (a permuted or sign-flipped gradient
        # has the right norm)
        assert grad_close(grad_sample(p.grad), torch.as_tensor(gold["Dgs_" + k]), GRAD_L2_TOL), k
    assert all(p.grad is None or p.grad.abs().max() == 0 for p in G.parameters())   # D step leaves G alone

    fill_module(G, seed=1)
    G.zero_grad(); D.zero_grad()
    for p in D.parameters():
        p.requires_grad_(False)
    r = tr.g_losses(labels, z, noise)
    r["g_loss"].backward()
    assert rel_err(r["g_loss"], torch.as_tensor(gold["g_loss"])) < 2e-4
    for k, p in G.named_parameters():
        ref_norm = float(gold["Ggn_" + k])
        assert abs(p.grad.double().norm().item() - ref_norm) <= GRAD_L2_TOL * ref_norm + 2e-6, k
        if not (k.endswith("residual.0.bias") or k in ("st_gcn_networks.%d.tcn.0.bias" % i for i in (1, 3, 5))):
            assert grad_close(grad_sample(p.grad), torch.as_tensor(gold["Ggs_" + k]), GRAD_L2_TOL), k
    assert all(p.grad is None or p.grad.abs().max() == 0 for p in D.parameters())


def test_gp_first_order_skips_param_grads(monkeypatch):
    """The penalty's autograd.grad must not launch weight-gradient kernels (they would be discarded)."""
    from kinetic_gan_amd import _native
    calls = {"wgrad": 0, "outer": 0}
    w0, o0 = _native.wgrad, _native.agg_outer
    monkeypatch.setattr(_native, "wgrad", lambda *a, **k: (calls.__setitem__("wgrad", calls["wgrad"] + 1), w0(*a, **k))[1])
    monkeypatch.setattr(_native, "agg_outer", lambda *a, **k: (calls.__setitem__("outer", calls["outer"] + 1), o0(*a, **k))[1])
    c, G, D, Go, Do = build_pair("h36m")
    real, labels, z, alpha = rand_inputs(2, 2, 32, 16, 10, 512, seed=1)
    inter = real.clone().requires_grad_(True)
    out = D(inter, labels)
    with ops.no_param_grads():
        torch.autograd.grad(out, inter, torch.ones_like(out), create_graph=True)
    assert calls == {"wgrad": 0, "outer": 0}


def test_trainer_iteration_matches_torch_adam():
    """d_step + g_step on flat buffers with kg_adam_step == the same losses stepped by torch.optim.Adam."""
    c, G, D, Go, Do = build_pair("h36m")
    nn_ = G.graph.num_node
    n = 4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(n, c["t_size"], nn_, seed=6)
    oG = torch.optim.Adam(Go.parameters(), lr=2e-4, betas=(0.5, 0.999))
    oD = torch.optim.Adam(Do.parameters(), lr=2e-4, betas=(0.5, 0.999))
    tr = Trainer(G, D)
    for it in range(2):
        tr.iteration(real, labels, z, alpha, noise, noise, with_g=True)
        oD.zero_grad()
        M.d_step_losses(Go, Do, real, labels, z, alpha, noise=noise)["d_loss"].backward()
        oD.step()
        oG.zero_grad()
        M.g_step_loss(Go, Do, labels, z, noise=noise)["g_loss"].backward()
        oG.step()
    # Adam normalises the gradient, so a sign-flipped tiny gradient moves a weight by up to 2*lr
    for (k, p), (_, q) in zip(list(D.named_parameters()) + list(G.named_parameters()),
                              list(Do.named_parameters()) + list(Go.named_parameters())):
        if k.endswith("residual.0.bias") or k in ("st_gcn_networks.%d.tcn.0.bias" % i for i in (1, 3, 5)):
            continue      # zero-gradient parameters (bias before a train-mode BN): Adam amplifies round-off to +-lr
        assert (p - q).abs().max().item() <= 2 * 2e-4 * 2 + 1e-6, k
        assert (p - q).abs().mean().item() <= 2e-5, k
    assert tr.fD.step.item() == 2 and tr.fG.step.item() == 2


def test_truncate_z_matches_generate_script():
    """generate.py:14-21 restated: latent[i] = m + truncation * (latent[i] - m), m = mean of mean_size normal draws
    taken from numpy's global generator."""
    import numpy as np
    from kinetic_gan_amd.generator import truncate_z
    lat = torch.randn(5, 512, generator=torch.Generator().manual_seed(3))
    np.random.seed(11)
    t = torch.as_tensor(np.random.normal(0, 1, (1000, 512)), dtype=torch.float32)
    want = lat.clone()
    m = t.mean(0, keepdim=True)
    for i in range(want.shape[0]):
        want[i] = m + 0.8 * (want[i] - m)
    np.random.seed(11)
    got = truncate_z(lat, 1000, 0.8)
    assert torch.allclose(got, want, rtol=0, atol=1e-6)
    assert torch.equal(truncate_z(lat, 1000, 0.8, t=t), got)


def test_step_launch_budget():
    """Native launches of one critic step and one generator step (h36m shapes).  Guards the structural savings:
    * the penalty's forward graph is NOT back-propagated with zero gradients (Functions return None for an absent
      gradient): 91 channel contractions per critic step instead of 107, two operand pairs per weight instead of three;
    * weight gradients are deferred and launched once per weight, their slab reductions in one call;
    * the two D passes of the critic step share one forward launch sequence and the masked adjacencies;
    * the real+fake backward pass rides along with the penalty's first backward pass (one launch sequence over 3n
      samples with the promised critic-loss gradient, disc_trunk.DiscTrunkFn): 16 contractions and 2 act_bwd fewer."""
    import collections
    from kinetic_gan_amd import _native
    from kinetic_gan_amd.wgan_gp import Trainer
    from oracle import prim_ref
    c, G, D, Go, Do = build_pair("h36m")
    nn_ = G.graph.num_node
    real, labels, z, alpha = rand_inputs(2, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    tr = Trainer(G, D)
    cnt, pairs = collections.Counter(), []
    saved = {}
    for name in prim_ref.NAMES:
        f = saved[name] = getattr(_native, name)

        def wrapped(*a, _n=name, _f=f, **k):
            cnt[_n] += 1
            if _n == "wgrad":
                pairs.append(1 + len(k.get("extra", ())))
            if _n == "wgrad_many":
                pairs.extend(1 + len(j.get("extra", ())) for j in a[0])
            if _n == "conv_many":
                cnt["conv_many_jobs"] += len(a[0])
            return _f(*a, **k)
        setattr(_native, name, wrapped)
    try:
        tr.d_compute(real, labels, z, alpha, None)
        d_cnt, d_pairs = dict(cnt), list(pairs)
        cnt.clear(); pairs.clear()
        tr.g_compute(labels, z, None)
        g_cnt, g_pairs = dict(cnt), list(pairs)
    finally:
        for name, f in saved.items():
            setattr(_native, name, f)
    n_dw = sum(1 for k, _ in D.named_parameters() if k.endswith("conv.weight") or k.endswith("tcn.weight") or k.endswith("residual.weight"))
    # every conv weight is ONE launch over two operand pairs (real+fake batch, the penalty's double backward);
    # block 0's gcn weight is addressed in place (data channels behind the label channels), not through a slice copy.
    # The trunk (disc_trunk.py) runs real+fake and the interpolates as one 3n forward: 12 contractions fewer.
    assert sorted(d_pairs) == [2] * n_dw, (d_cnt, d_pairs)
    # ... and all of them share ONE kg_wgrad_many call per pass
    assert d_cnt["wgrad_many"] == 1 and g_cnt["wgrad_many"] == 1 and "wgrad" not in d_cnt and g_pairs == [1] * 19
    # act_bwd: the LeakyReLU derivative is applied by the launch that produces the gradient (kg_conv mask epilogue)
    # or by kg_agg_reduce's epilogue behind the down-sampling / identity-residual blocks; the top of the chain gets it from
    # kg_head_bwd, which builds the top gradient from d loss / d validity: no separate act_bwd launch is left
    # (the four stride-2 blocks' transposed temporal convs run as two parity launches each: 59 + 4)
    # (... and share ONE kg_conv_many launch per block with the residual branch's small product: 6 launches for the
    # 10 + 3 problems of the merged backward pass)
    assert d_cnt["conv"] + d_cnt["conv_many_jobs"] == 63 and d_cnt["conv_many"] == 6 and d_cnt["conv_many_jobs"] == 13, d_cnt
    assert d_cnt["agg_outer"] == 12 and "act_bwd" not in d_cnt, d_cnt
    # the container-level fusions: one launch each for the 3n critic input, the head of the 3n forward, the top gradient
    # of the merged backward, the label bias and its gradients, the head's weight gradient (first order + the
    # penalty's double backward), the masked adjacencies and their gradient
    assert d_cnt["mix3"] == 1 and d_cnt["head_fwd"] == 1 and d_cnt["head_bwd"] == 1 and d_cnt["head_wgrad"] == 2, d_cnt
    assert d_cnt["label_bias_fwd"] == 1 and d_cnt["label_bias_bwd"] == 1, d_cnt
    assert d_cnt["masked_adj_fwd"] == 1 and d_cnt["masked_adj_bwd"] == 1, d_cnt
    assert g_cnt["conv"] + g_cnt["conv_many_jobs"] == 70 and g_cnt["conv_many"] == 6 and g_cnt.get("agg_outer", 0) == 7, g_cnt


@pytest.mark.parametrize("cfg", ["h36m"])
def test_trunk_equals_blockwise_path(cfg):
    """disc_trunk.py (one autograd node, hand-scheduled FWD / BWD / DBL) against the block-by-block ops.py path:
    losses, every parameter gradient of the critic step (incl. the penalty's double backward) and the gradient the
    generator step sends back to the fake batch."""
    c, G, D, Go, Do = build_pair(cfg)
    nn_ = G.graph.num_node
    n = 3
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=4)
    noise = rand_noise(n, c["t_size"], nn_, seed=7)
    tr = Trainer(G, D, flatten=False)
    res = {}
    for mode in (True, False):
        D.use_trunk = mode
        D.zero_grad()
        r = tr.d_losses(real, labels, z, alpha, noise)
        r["d_loss"].backward()
        grads = {k: p.grad.clone() for k, p in D.named_parameters()}
        x = real.clone().requires_grad_(True)
        D.zero_grad()
        D(x, labels).sum().backward()
        res[mode] = (r["d_loss"].detach(), r["gradient_penalty"].detach(), grads, x.grad.clone())
    assert torch.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[True][1], res[False][1], rtol=1e-5, atol=1e-7)
    for k in res[True][2]:
        assert l2_rel(res[True][2][k], res[False][2][k]) < 1e-5, k
    assert l2_rel(res[True][3], res[False][3]) < 1e-5


def test_param_sink_many_contributions_per_weight():
    """More than three deferred operand pairs for one weight (gradient accumulation over several backward passes
    before the bucket is read): every pair must arrive in the bucket exactly once."""
    c, G, D, Go, Do = build_pair("h36m")
    nn_ = G.graph.num_node
    n = 2
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=5)
    noise = rand_noise(n, c["t_size"], nn_, seed=8)
    tr = Trainer(G, D)
    tr.fD.zero_grad()
    for _ in range(3):        # 3 x (real+fake, double backward) = 6 pairs per weight, launched in two rounds
        tr.d_losses(real, labels, z, alpha, noise)["d_loss"].backward()
    assert max(len(e[2]) for e in ops._SINK.pending.values()) == 6
    tr.fD.gather_grads()
    acc = {k: p.grad.clone() for k, p in D.named_parameters()}
    tr.fD.zero_grad()
    tr.d_losses(real, labels, z, alpha, noise)["d_loss"].backward()
    tr.fD.gather_grads()
    for k, p in D.named_parameters():
        assert l2_rel(acc[k], 3 * p.grad) < 1e-5, k


@pytest.mark.parametrize("cfg", ["h36m", "ntu120"])
def test_mapping_node_equals_stock_ops(cfg):
    """ops.MappingFn (embedding + cat + the mlp's Linear / LeakyReLU pairs as one autograd node over kg_linear_*) against
    the same modules on stock ops (Generator.map_kernels = False; generator.py:80-85): the mapped latents, and through a
    generator step every gradient of the embedding and the mapping network - as fresh autograd tensors
    (Trainer(flatten=False)) and added straight into the flat bucket (Trainer())."""
    c, G, D, Go, Do = build_pair(cfg)
    c2, G2, D2, _, _ = build_pair(cfg)
    G2.map_kernels = False
    nn_ = G.graph.num_node
    n = 3
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=5)
    noise = rand_noise(n, c["t_size"], nn_, seed=8)
    with torch.no_grad():
        wa, wb = G.mapping(z, labels), G2.mapping(z, labels)
    assert rel_err(wa, wb) < 1e-5
    assert rel_err(wa, Go.mlp(torch.cat((Go.label_emb(labels), z), -1))) < 1e-5
    for flatten in (False, True):
        ta, tb = Trainer(G, D, flatten=flatten), Trainer(G2, D2, flatten=flatten)
        if flatten:
            ta.g_compute(labels, z, noise)
            tb.g_compute(labels, z, noise)
        else:
            for t, g_ in ((ta, G), (tb, G2)):
                g_.zero_grad()
                t.g_losses(labels, z, noise)["g_loss"].backward()
        for (k, p), (_, q) in zip(G.named_parameters(), G2.named_parameters()):
            if k.startswith("mlp.") or k.startswith("label_emb."):
                assert grad_close(p.grad, q.grad, 1e-4), (flatten, k, l2_rel(p.grad, q.grad))


def test_failed_backward_leaves_no_debris_for_the_next_step(monkeypatch):
    """A generator backward pass that raises midway (here: inside the deferred adjacency-gradient launch) leaves
    recorded outer-product problems and deferred weight-gradient pairs behind; FlatParams.zero_grad of the next step must
    drop them (ops.reset_param_sink(bucket), round-3 ADVICE): the next step's gradients equal a clean run's."""
    from kinetic_gan_amd import _native as nv
    c, G, D, Go, Do = build_pair("h36m")
    c2, G2, D2, _, _ = build_pair("h36m")
    nn_ = G.graph.num_node
    n = 2
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=5)
    noise = rand_noise(n, c["t_size"], nn_, seed=8)
    tr, tr2 = Trainer(G, D), Trainer(G2, D2)
    real_conv = nv.conv
    calls = {"n": 0, "fail_at": -1}

    def counting(*a, **k):
        calls["n"] += 1
        if calls["n"] == calls["fail_at"]:
            raise RuntimeError("injected failure in a launch of the backward pass")
        return real_conv(*a, **k)
    monkeypatch.setattr(nv, "conv", counting)
    tr2.g_compute(labels, z, noise)                  # clean run on the twin: counts the launches of one generator step
    total = calls["n"]
    want = {k: p.grad.clone() for k, p in G2.named_parameters()}
    calls["n"], calls["fail_at"] = 0, total - 2      # the last-but-one contraction of the backward pass raises
    with pytest.raises(RuntimeError, match="injected"):
        tr.g_compute(labels, z, noise)
    monkeypatch.setattr(nv, "conv", real_conv)
    assert ops._OUTER_PENDING or ops._SINK.pending or ops._SINK.pending_rows      # debris of the failed pass
    ops._OUTER_PENDING.append(dict(args=None, keep=()))                           # (and a stale adjacency-gradient record)
    tr.fG.zero_grad()
    assert not ops._OUTER_PENDING and not ops._SINK.pending and not ops._SINK.pending_rows
    tr.g_compute(labels, z, noise)
    for k, p in G.named_parameters():
        assert torch.equal(p.grad, want[k]), k


def test_trunk_fused_gcn_path(monkeypatch):
    """The trunk with the fused aggregation + gcn launch (kg_aggconv) forced on at test sizes: same losses and
    gradients as the unfused launches; the emulation also checks every neighbour table against its adjacency."""
    from kinetic_gan_amd import _native
    c, G, D, Go, Do = build_pair("ntu")
    nn_ = G.graph.num_node
    n = 2
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=4)
    noise = rand_noise(n, c["t_size"], nn_, seed=7)
    tr = Trainer(G, D, flatten=False)
    res = {}
    for fused in (False, True):
        calls = {"n": 0}
        if fused:
            f0 = _native.aggconv
            monkeypatch.setattr(_native, "aggconv_supported", lambda V, W, pc, ncols: (127 // W + 2) * V <= 384)
            monkeypatch.setattr(_native, "aggconv", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), f0(*a, **k))[1])
        D.zero_grad()
        r = tr.d_losses(real, labels, z, alpha, noise)
        r["d_loss"].backward()
        res[fused] = (r["d_loss"].detach(), {k: p.grad.clone() for k, p in D.named_parameters()})
        if fused:
            assert calls["n"] >= 6, calls        # blocks 0-2 (block 3 too where no aggregated planes are asked for) of the 3n forward and of the double backward
    assert torch.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    for k in res[True][1]:
        assert l2_rel(res[True][1][k], res[False][1][k]) < 1e-5, k


@pytest.mark.parametrize("cfg", ["ntu", "ntu120", "h36m", "stress"])
def test_reference_state_dict_loads_strict(cfg, golden_dir):
    """Checkpoint drop-in (generate.py:66, kinetic-gan.py:189-192): a state_dict with exactly the REFERENCE's keys,
    shapes and dtypes (tests/golden/state_dict_listing.json, written by make_state_dict_listing.py from the imported
    reference) loads into the HIP-path modules with strict=True, and the modules write back the same listing."""
    import json
    import os
    from kinetic_gan_amd.discriminator import Discriminator
    from kinetic_gan_amd.generator import Generator
    from tests.util import ds_name
    lst = json.load(open(os.path.join(golden_dir, "state_dict_listing.json")))[cfg]
    c = CFG[cfg]
    G = Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=ds_name(cfg))
    D = Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=ds_name(cfg))
    for m, key in ((G, "G"), (D, "D")):
        sd = {k: torch.full(shape, 0.25 if dt.startswith("float") else 3, dtype=getattr(torch, dt)) for k, shape, dt in lst[key]}
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        assert mine == lst[key]
        assert all(bool((v == (0.25 if v.is_floating_point() else 3)).all()) for v in m.state_dict().values())


def test_c5b_stress_config_blocks_and_step():
    """BASELINE configs[4] end to end: full G / D at t_size = 256 on the NTU graph, small N -
    forward of both models, the critic losses incl. the penalty and every D gradient against the oracle."""
    c, G, D, Go, Do = build_pair("stress")
    nn_ = G.graph.num_node
    n = 2
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=13)
    noise = rand_noise(n, c["t_size"], nn_, seed=14)
    ro = M.d_step_losses(Go, Do, real, labels, z, alpha, noise=noise)
    Do.zero_grad()
    ro["d_loss"].backward()
    tr = Trainer(G, D, flatten=False)
    r = tr.d_losses(real, labels, z, alpha, noise)
    D.zero_grad()
    r["d_loss"].backward()
    assert tuple(r["fake"].shape) == (n, 3, 256, 25)
    for k in ("fake", "real_validity", "fake_validity"):
        assert rel_err(r[k], ro[k]) < FWD_TOL, k
    assert rel_err(r["gradient_penalty"], ro["gradient_penalty"]) < 5e-4
    for (k, p), (_, q) in zip(D.named_parameters(), Do.named_parameters()):
        assert grad_close(p.grad, q.grad, GRAD_L2_TOL), (k, l2_rel(p.grad, q.grad))


def test_eval_bn_folding_and_sampling_loop(golden_dir):
    """Inference path (generate.py:66-105): eval-mode BatchNorm folded into the tcn / residual conv weights gives the
    reference's eval output (fixture G_out_eval) and the unfolded result; the sampling loop hands back >= gen_qtd
    samples per class in generate.py's label order, drawing its latents from numpy's global generator like the
    script."""
    import os
    from kinetic_gan_amd import _native
    from kinetic_gan_amd.sample import sample_actions
    gold = np.load(os.path.join(golden_dir, "ref_h36m.npz"))
    c, G, D, Go, Do = build_pair("h36m")
    nn_ = G.graph.num_node
    real, labels, z, alpha = rand_inputs(4, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(4, c["t_size"], nn_, seed=6)
    G.eval()
    unfolded = G(z, labels, noise=noise)
    calls = {"bn": 0}
    f0 = _native.bn_fwd
    _native.bn_fwd = lambda *a, **k: (calls.__setitem__("bn", calls["bn"] + 1), f0(*a, **k))[1]
    try:
        with torch.no_grad():
            folded = G(z, labels, noise=noise)
    finally:
        _native.bn_fwd = f0
    assert calls["bn"] == 0                         # no statistics / coefficient launch left on the inference path
    assert rel_err(folded, unfolded) < 1e-5
    assert rel_err(folded, torch.as_tensor(gold["G_out_eval"])) < FWD_TOL
    # a parameter update invalidates the folded weights
    with torch.no_grad():
        G.st_gcn_networks[1].tcn[1].weight.mul_(1.5)
        assert rel_err(G(z, labels, noise=noise), G(z, labels, noise=noise)) == 0
    G.train(True)
    unf2 = None
    G.eval()
    unf2 = G(z, labels, noise=noise)
    with torch.no_grad():
        assert rel_err(G(z, labels, noise=noise), unf2) < 1e-5
    # the sampling loop: 10 classes, 3 per round, at least 5 each -> two rounds of 30
    np.random.seed(5)
    imgs, labs, zs = sample_actions(G, c["n_classes"], c["latent"], gen_qtd=5, qtd=3)
    assert tuple(imgs.shape) == (60, c["channels"], c["t_size"], nn_[0]) and tuple(zs.shape) == (60, c["latent"])
    assert labs.tolist() == [k for _ in range(2) for _ in range(3) for k in range(c["n_classes"])]
    np.random.seed(5)
    z0 = torch.as_tensor(np.random.normal(0, 1, (30, c["latent"])), dtype=torch.float32)
    assert torch.equal(zs[:30], z0)
    assert G.training is False
    imgs1, labs1, _ = sample_actions(G, c["n_classes"], c["latent"], gen_qtd=2, qtd=2, label=7, trunc=0.9, trunc_mode="w")
    assert labs1.tolist() == [7, 7] and tuple(imgs1.shape) == (2, c["channels"], c["t_size"], nn_[0])


@pytest.mark.parametrize("cfg", ["h36m", "ntu"])
def test_generator_trunk_equals_blockwise_path(cfg):
    """gen_trunk.GenTrunkFn (one autograd node, every block contract-first on its INPUT grid: [W_gcn; W_res] x, then
    kg_gen_expand) against the block-by-block ops.py path, both on emulated kernels: the paired 2n synthesis of a
    WGAN-GP iteration (critic sample without history + generator-step sample with it), the generator loss, EVERY
    generator parameter's gradient-bucket slice, the BatchNorm running statistics after both updates, and a plain
    single-batch forward / backward."""
    from kinetic_gan_amd.wgan_gp import Trainer
    with emulated_native():
        res = {}
        for trunk in (False, True):
            c, G, D, Go, Do = build_pair(cfg)
            G._pack_always = True          # the packed-adjacency / trunk paths are GPU-only by default
            G.use_trunk = trunk
            nn_ = G.graph.num_node
            n = 3
            real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=11)
            nd, ng = rand_noise(n, c["t_size"], nn_, seed=12), rand_noise(n, c["t_size"], nn_, seed=13)
            tr = Trainer(G, D)
            with tr.sharing_mapping(ng):
                tr.d_compute(real, labels, z, alpha, nd)
            assert (G._trunk_state(z) is not None) == trunk
            g_loss = tr.g_compute(labels, z, ng)
            first = (g_loss.clone(), tr.fG.grad.clone())
            stats = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k or "num_batches" in k}
            # single batch, with history
            tr.fG.zero_grad()
            out = G(z, labels, noise=ng)
            (out * torch.linspace(-1, 1, out.numel()).view(out.shape)).sum().backward()
            tr.fG.gather_grads()
            res[trunk] = first + (stats, out.detach().clone(), tr.fG.grad.clone(),
                                  [(k, p.numel(), off) for (k, p), off in zip(G.named_parameters(), tr.fG.offsets)])
        a, b = res[False], res[True]
        assert rel_err(b[0], a[0]) < 1e-5
        assert rel_err(b[3], a[3]) < 1e-5
        for k in a[2]:
            assert rel_err(b[2][k].float(), a[2][k].float()) < 1e-5, k
        for which in (1, 4):
            for k, nel, off in a[5]:
                ga, gb = a[which][off:off + nel], b[which][off:off + nel]
                # analytically (near-)zero gradients are round-off on both sides: conv biases in front of a train-mode
                # BatchNorm, and the single non-zero adjacency entry of block 1 (a pure scale in front of BatchNorm)
                if _analytic_zero(k):
                    assert (ga - gb).abs().max().item() < 1e-5 * max(1.0, a[which].abs().max().item()), k
                else:
                    assert grad_close(gb, ga, 2e-5), (which, k, l2_rel(gb, ga))


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_fused_generator_blocks_equal_staged_blocks(cfg, monkeypatch):
    """gen_trunk.FUSED (kg_genblock_fwd / kg_genblock_bwd: one launch per block and direction, the previous block's tail
    applied at the front of the next launch, the previous block's tail statistics taken at the end of a backward launch)
    against the staged launch sequence of the same trunk, on emulated kernels: the paired 2n synthesis, the generator loss,
    every gradient-bucket slice, the BatchNorm running statistics, and which blocks took the fused form."""
    from kinetic_gan_amd import _native, gen_trunk
    from kinetic_gan_amd.wgan_gp import Trainer
    with emulated_native():
        res = {}
        for fused in (False, True):
            monkeypatch.setattr(gen_trunk, "FUSED", fused)
            calls = {"f": 0, "b": 0}
            f0, b0 = _native.genblock_fwd, _native.genblock_bwd
            monkeypatch.setattr(_native, "genblock_fwd", lambda *a, **k: (calls.__setitem__("f", calls["f"] + 1), f0(*a, **k))[1])
            monkeypatch.setattr(_native, "genblock_bwd", lambda *a, **k: (calls.__setitem__("b", calls["b"] + 1), b0(*a, **k))[1])
            c, G, D, Go, Do = build_pair(cfg)
            G._pack_always = True
            nn_ = G.graph.num_node
            n = 3
            real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=11)
            nd, ng = rand_noise(n, c["t_size"], nn_, seed=12), rand_noise(n, c["t_size"], nn_, seed=13)
            tr = Trainer(G, D)
            with tr.sharing_mapping(ng):
                tr.d_compute(real, labels, z, alpha, nd)
            g_loss = tr.g_compute(labels, z, ng)
            stats = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k or "num_batches" in k}
            res[fused] = (g_loss.clone(), tr.fG.grad.clone(), stats,
                          [(k, p.numel(), off) for (k, p), off in zip(G.named_parameters(), tr.fG.offsets)])
            # the blocks with >= 16 input-grid columns per sample (NTU: the last four, H36M at T = 32: the last two) take the
            # fused form: once forward (2n), once backward
            nf = {"ntu": 4, "h36m": 2}[cfg]
            assert (calls["f"], calls["b"]) == ((nf, nf) if fused else (0, 0)), calls
            monkeypatch.setattr(_native, "genblock_fwd", f0)
            monkeypatch.setattr(_native, "genblock_bwd", b0)
        a, b = res[False], res[True]
        assert rel_err(b[0], a[0]) < 1e-5
        for k in a[2]:
            assert rel_err(b[2][k].float(), a[2][k].float()) < 1e-5, k
        for k, nel, off in a[3]:
            ga, gb = a[1][off:off + nel], b[1][off:off + nel]
            if _analytic_zero(k):
                assert (ga - gb).abs().max().item() < 1e-5 * max(1.0, a[1].abs().max().item()), k
            else:
                # (a gradient that is small against the bucket - block 0's single adjacency entry, a scale two blocks in front
                # of a BatchNorm - is dominated by the other summation order of the fused blocks' BatchNorm merge)
                assert grad_close(gb, ga, 2e-5) or (ga - gb).abs().max().item() < 1e-5 * a[1].abs().max().item(), (k, l2_rel(gb, ga))


def _analytic_zero(k):
    return (k.endswith("residual.0.bias") or any(k.endswith("st_gcn_networks.%d.tcn.0.bias" % i) for i in (1, 3, 5))
            or k == "edge_importance.1")


def test_folded_inference_follows_training_steps():
    """train -> sample -> train -> sample: the folded (eval, no_grad) forward must use the CURRENT weights and running
    statistics after optimiser steps that rewrite them through the flat buffer / raw pointers (round-2 ADVICE: a fold
    cache keyed on tensor version counters kept serving the first fold)."""
    from kinetic_gan_amd.wgan_gp import Trainer
    with emulated_native():
        c, G, D, Go, Do = build_pair("h36m")
        nn_ = G.graph.num_node
        n = 3
        real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=31)
        noise = rand_noise(n, c["t_size"], nn_, seed=32)
        tr = Trainer(G, D)
        outs = []
        for rnd_ in range(3):
            G.eval()
            with torch.no_grad():
                folded = G(z, labels, noise=noise)
            unfolded = G(z, labels, noise=noise).detach()          # autograd on: BatchNorm applied, not folded
            assert rel_err(folded, unfolded) < 1e-5, rnd_
            outs.append(folded)
            G.train()
            tr.iteration(real, labels, z, alpha, noise, noise, with_g=True)
        assert rel_err(outs[1], outs[0]) > 1e-4 and rel_err(outs[2], outs[1]) > 1e-4      # the generator did move


def test_paired_synthesis_equals_two_forward_passes():
    """Generator.synthesis_pair (both syntheses of a WGAN-GP iteration as ONE 2n pass, BatchNorm statistics per half)
    against two separate forward passes: both samples, the running statistics after the two updates, and every
    generator gradient of a loss on the second sample."""
    c, G, D, Go, Do = build_pair("h36m")
    c2, G2, _, _, _ = build_pair("h36m")
    nn_ = G.graph.num_node
    n = 3
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=21)
    na, nb = rand_noise(n, c["t_size"], nn_, seed=22), rand_noise(n, c["t_size"], nn_, seed=23)
    with torch.no_grad():
        fa_ref = G2(z, labels, noise=na)
    fb_ref = G2(z, labels, noise=nb)
    go = torch.randn(fb_ref.shape, generator=torch.Generator().manual_seed(5))
    (fb_ref * go).sum().backward()
    fa, fb = G.synthesis_pair(G.mapping(z, labels), na, nb)
    assert not fa.requires_grad and fb.requires_grad
    (fb * go).sum().backward()
    assert rel_err(fa, fa_ref) < 1e-5 and rel_err(fb, fb_ref) < 1e-5
    for (k, v), (_, w) in zip(G.state_dict().items(), G2.state_dict().items()):
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(v.numpy(), w.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    for (k, p), (_, q) in zip(G.named_parameters(), G2.named_parameters()):
        if k.endswith("residual.0.bias") or k in ("st_gcn_networks.%d.tcn.0.bias" % i for i in (1, 3, 5)):
            continue
        # (batch reductions over 2n rows, half of them exact zeros, vs n rows: summation order only; a sum of
        # cancelling terms like the one-element edge_importance gradient moves by ~1e-3 relative)
        assert grad_close(p.grad, q.grad, GRAD_L2_TOL), (k, l2_rel(p.grad, q.grad))


def test_merged_critic_backward_equals_separate_passes(monkeypatch):
    """wgan_gp.Trainer.d_compute with the promised critic-loss gradient (the real+fake backward pass folded into the
    gradient penalty's first backward pass, disc_trunk.DiscTrunkFn) against the two separate passes: every entry of
    the flat gradient bucket, and the promise itself is checked against the gradient autograd delivers."""
    from kinetic_gan_amd import disc_trunk
    from kinetic_gan_amd.wgan_gp import Trainer
    monkeypatch.setattr(disc_trunk, "_CHECK_PROMISE", True)
    c, G, D, Go, Do = build_pair("h36m")
    nn_ = G.graph.num_node
    n = 3
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=21)
    noise = rand_noise(n, c["t_size"], nn_, seed=22)
    tr = Trainer(G, D)
    grads, losses = {}, {}
    for promise in (True, False):
        tr._promise = promise
        losses[promise] = tr.d_compute(real, labels, z, alpha, noise).clone()
        grads[promise] = tr.fD.grad.clone()
    assert tr.fD.grad.abs().max() > 0
    assert torch.allclose(losses[True], losses[False], rtol=1e-6, atol=1e-7)
    scale = grads[False].abs().max()
    assert (grads[True] - grads[False]).abs().max() <= 2e-5 * scale, (grads[True] - grads[False]).abs().max() / scale


def test_async_checkpoint_writer_roundtrip(tmp_path):
    """checkpoint.AsyncCheckpointWriter: the files are plain torch.save'd state dicts with the reference's keys - they
    load strictly into a fresh model and reproduce the parameters and buffers as of the save() call, not later ones."""
    from kinetic_gan_amd.checkpoint import AsyncCheckpointWriter
    c, G, D, Go, Do = build_pair("h36m")
    w = AsyncCheckpointWriter()
    want = {k: v.clone() for k, v in G.state_dict().items()}
    w.save(G, str(tmp_path / "generator_0.pth"))
    with torch.no_grad():
        for p in G.parameters():
            p.add_(1.0)                                  # training moves on while the writer works
    w.save(D, str(tmp_path / "discriminator_0.pth"))
    w.close()
    got = torch.load(str(tmp_path / "generator_0.pth"))
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    Go.load_state_dict(got, strict=True)                 # the oracle restatement has the reference's keys / shapes
    Do.load_state_dict(torch.load(str(tmp_path / "discriminator_0.pth")), strict=True)


def test_flat_params_keep_parameter_lists_adjacent():
    """round-4 ADVICE: the 16-byte alignment of large tensors in the flat buffers must not split a ParameterList (the edge
    importances): MaskedAdjacencyFn / disc_trunk._pack rely on their slices being adjacent, in G and D of every config."""
    from kinetic_gan_amd.wgan_gp import FlatParams
    for cfg_name in ("ntu", "h36m", "ntu120"):
        _, G, D, _, _ = build_pair(cfg_name)
        for net in (G, D):
            fp = FlatParams(net)
            offs = {name: (off, p.numel()) for (name, p), off in zip(net.named_parameters(), fp.offsets)}
            imp = sorted((int(n.split(".")[-1]), v) for n, v in offs.items() if n.startswith("edge_importance."))
            assert len(imp) >= 6
            for (_, (o0, n0)), (_, (o1, _)) in zip(imp, imp[1:]):
                assert o1 == o0 + n0, (cfg_name, type(net).__name__, imp)
            for name, p in net.named_parameters():           # the 16-byte loads of kg_linear_* / the ring form's DMA
                if p.numel() >= 1024 and not (name.startswith("edge_importance.") and not name.endswith(".0")):
                    assert offs[name][0] % 4 == 0, name
            packed = __import__("kinetic_gan_amd.disc_trunk", fromlist=["_pack"])._pack(list(net.edge_importance))
            assert packed.data_ptr() == net.edge_importance[0].data_ptr()          # a view, not a concatenation


def test_generator_mapping_on_cpu_tensors_uses_stock_ops():
    """round-4 ADVICE: without a GPU (and outside the emulation) Generator.mapping must not call the kernels."""
    _, G, _, Go, _ = build_pair("ntu")
    z = torch.randn(3, 512)
    labels = torch.tensor([1, 5, 59])
    with torch.no_grad():
        w = G.mapping(z, labels)
        wo = Go.mlp(torch.cat((Go.label_emb(labels), z), -1))
    assert rel_err(w, wo) < 1e-5
