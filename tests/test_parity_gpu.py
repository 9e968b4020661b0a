"""-m gpu: the HIP path (modules -> autograd ops -> C ABI -> gfx950 kernels) against
 (a) the REFERENCE's own outputs captured in tests/golden/ref_*.npz, and
 (b) the oracle modules run on the host with identical parameters / inputs / injected noise.

Tolerances: forward values max|a-b| <= 1e-4 * max|b| (BASELINE.json north_star: "forward outputs
within 1e-4 rel of reference"); gradients by relative L2 <= 5e-3 (LeakyReLU kink, see
tests/test_host_logic_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import modules_ref as M
from oracle.fill import (block_input, disc_block_in_shapes, fill_module, gen_block_in_shapes,
                         rand_inputs, rand_noise)
from tests.util import CFG, build_pair, ds_name, grad_close, grad_sample, l2_rel, rel_err

from kinetic_gan_amd.wgan_gp import Trainer

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-4
GRAD_TOL = 5e-3


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _log(msg):
    """Per-tensor error tables go to gpurun_out/parity_detail.log (kept even if a later assert fails)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_detail.log"), "a") as f:
        f.write(msg + "\n")


def _zero_grad_keys(k, blk=None):
    """GENERATOR parameters whose gradient is analytically zero, i.e. round-off (1e-8 .. 1e-4) on both sides: conv biases
    in front of a train-mode BatchNorm (the batch mean removes them) and block 1's adjacency entry - at the one-vertex
    level A[lvl] * importance is a single scalar that scales the gcn output right in front of BatchNorm."""
    return (k.endswith("residual.0.bias") or any(k.endswith("st_gcn_networks.%d.tcn.0.bias" % i) for i in (1, 3, 5))
            or k == "edge_importance.1")


@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_blocks_vs_reference_golden(cfg, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"ref_{cfg}.npz"))
    d = dev()
    c, G, D, Go, Do = build_pair(cfg, d)
    nn_ = G.graph.num_node
    n = 2
    noise = rand_noise(n, c["t_size"], nn_, seed=5, device=d)
    gs = gen_block_in_shapes(n, c["latent"] + c["n_classes"], c["channels"], c["t_size"], nn_)
    for mode in ("train", "eval"):
        G.train(mode == "train")
        fill_module(G, seed=1)
        for i, (blk, imp) in enumerate(zip(G.st_gcn_networks, G.edge_importance)):
            y, _ = blk(block_input(gs[i], 200 + i).to(d), G.A[blk.lvl] * imp, noise[i])
            assert tuple(y.shape) == gold[f"G{i}_{mode}"].shape
            assert rel_err(y, torch.as_tensor(gold[f"G{i}_{mode}"])) < FWD_TOL, (mode, i)
        if mode == "train":
            for k, v in G.state_dict().items():
                if "running_" in k:
                    np.testing.assert_allclose(v.cpu().numpy(), gold["Gstat_" + k], rtol=2e-4, atol=1e-5, err_msg=k)
    G.train(True)
    dsh = disc_block_in_shapes(n, c["channels"] + c["n_classes"], c["latent"], c["t_size"], nn_)
    for i, (blk, imp) in enumerate(zip(D.st_gcn_networks, D.edge_importance)):
        y, _ = blk(block_input(dsh[i], 400 + i).to(d), D.A[blk.lvl] * imp)
        assert tuple(y.shape) == gold[f"D{i}"].shape
        assert rel_err(y, torch.as_tensor(gold[f"D{i}"])) < FWD_TOL, i


@pytest.fixture
def conv_path(request, monkeypatch):
    """"direct": the launcher's own plans; "ring<t>": every full-slice kg_conv launch on tile t of the persistent LDS-ring
    form (tools/probe/kg_conv_ring.hip; only with a `build.py --with-ring` library and KG_TEST_RING=1), "bs": every eligible launch on the bf16-split LDS-staged form (kg_conv_bs_kernel) - the
    whole model, the WGAN-GP step and its gradients then run through it."""
    from kinetic_gan_amd import _native as nv
    if request.param == "bs":
        monkeypatch.setenv("KG_CONV_BS", "1")
    elif request.param != "direct":
        monkeypatch.setenv("KG_CONV_RING", "1")
        monkeypatch.setenv("KG_CONV_RING_TILE", request.param[4:])
    nv.reload_env()
    yield request.param
    monkeypatch.undo()
    nv.reload_env()


@pytest.mark.parametrize("conv_path", ["direct", "bs"] + (["ring1", "ring6"] if os.environ.get("KG_TEST_RING", "0") == "1" else []),
                         indirect=True)
@pytest.mark.parametrize("cfg", ["ntu", "h36m"])
def test_models_and_wgan_gp_step_vs_reference_golden(cfg, golden_dir, conv_path):
    gold = np.load(os.path.join(golden_dir, f"ref_{cfg}.npz"))
    d = dev()
    c, G, D, Go, Do = build_pair(cfg, d)
    nn_ = G.graph.num_node
    n = 4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3, device=d)
    noise = rand_noise(n, c["t_size"], nn_, seed=6, device=d)
    for mode in ("train", "eval"):
        G.train(mode == "train")
        fill_module(G, seed=1)
        assert rel_err(G(z, labels, noise=noise), torch.as_tensor(gold[f"G_out_{mode}"])) < FWD_TOL, mode
    # inference path of generate.py:90-93: eval mode, W-space truncation (generator.py:86,97-108); the 1000
    # truncation latents come from numpy's global generator, seeded as in tests/golden/make_fixtures.py
    np.random.seed(77)
    with torch.no_grad():
        assert rel_err(G(z, labels, trunc=0.7, noise=noise), torch.as_tensor(gold["G_out_eval_trunc"])) < FWD_TOL
    G.train(True)
    fill_module(G, seed=1)
    assert rel_err(D(real, labels), torch.as_tensor(gold["D_out"])) < FWD_TOL

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
        assert abs(p.grad.double().norm().item() - ref_norm) <= GRAD_TOL * ref_norm + 2e-6, k
        # ... and a strided 64-element sample of every gradient, element-wise (a permuted or sign-flipped gradient
        # has the right norm)
        assert grad_close(grad_sample(p.grad), torch.as_tensor(gold["Dgs_" + k]), GRAD_TOL), k
    fill_module(G, seed=1)
    G.zero_grad()
    for p in D.parameters():
        p.requires_grad_(False)
    r = tr.g_losses(labels, z, noise)
    r["g_loss"].backward()
    assert rel_err(r["g_loss"], torch.as_tensor(gold["g_loss"])) < 2e-4
    for k, p in G.named_parameters():
        if _zero_grad_keys(k):
            continue
        ref_norm = float(gold["Ggn_" + k])
        assert abs(p.grad.double().norm().item() - ref_norm) <= GRAD_TOL * ref_norm + 2e-6, k
        assert grad_close(grad_sample(p.grad), torch.as_tensor(gold["Ggs_" + k]), GRAD_TOL), k


@pytest.mark.parametrize("cfg,n", [("ntu", 64), ("ntu120", 32), ("h36m", 64), ("stress", 2), ("ntu", 5), ("h36m", 3)])
def test_full_size_vs_oracle(cfg, n):
    """BASELINE configs at their real batch sizes (C2: NTU bs=64; C3: NTU-120 mlp8 at its per-GPU shard of 32;
    C4: H36M bs=64) and C5b (full G / D at t_size=256) at a batch the host oracle finishes in seconds:
    forward of G and D, the WGAN-GP D-step losses and every parameter gradient vs. the oracle on the host."""
    d = dev()
    c, G, D, Go, Do = build_pair(cfg, d)
    nn_ = G.graph.num_node
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=11)
    noise = rand_noise(n, c["t_size"], nn_, seed=12)
    to = lambda t: t.to(d)
    noise_d = [to(t) for t in noise]
    from oracle.host import usable_cores
    torch.set_num_threads(usable_cores())
    ro = M.d_step_losses(Go, Do, real, labels, z, alpha, noise=noise)
    Do.zero_grad()
    ro["d_loss"].backward()
    tr = Trainer(G, D, flatten=False)
    with torch.no_grad():
        fake_hip = G(to(z), to(labels), noise=noise_d)
    # D is fed the SAME fake batch as the oracle: the generator's own deviation (checked just below) would
    # otherwise flip LeakyReLU slopes of near-zero activations and blur the gradient comparison
    r = tr.d_losses(to(real), to(labels), to(z), to(alpha), noise_d, fake=to(ro["fake"].detach()))
    r["fake"] = fake_hip
    D.zero_grad()
    r["d_loss"].backward()
    rows = ["%s n=%d" % (cfg, n)]
    for k in ("fake", "real_validity", "fake_validity", "gradient_penalty", "d_loss"):
        rows.append("  %-18s rel_err %.3e" % (k, rel_err(r[k], ro[k])))
    bad = []
    for (k, p), (_, q) in zip(D.named_parameters(), Do.named_parameters()):
        e = l2_rel(p.grad, q.grad)
        rows.append("  grad %-40s l2_rel %.3e  max|ref| %.3e" % (k, e, q.grad.abs().max().item()))
        if not grad_close(p.grad, q.grad, GRAD_TOL):
            bad.append((k, e))
    if cfg == "ntu":
        # how far apart are two stock-PyTorch runs of the SAME oracle (host vs. device)?  Sets the scale for
        # what "agreement" of these kinked (LeakyReLU) gradients can mean at this batch size.
        import copy
        Dg = copy.deepcopy(Do).to(d)
        Dg.zero_grad()
        rg = M.gradient_penalty(Dg, to(real), to(ro["fake"].detach()), to(labels), to(alpha))
        both = Dg(torch.cat((to(real), to(ro["fake"].detach())), 0), torch.cat((to(labels), to(labels)), 0))
        (-both[:n].mean() + both[n:].mean() + 10 * rg).backward()
        worst = max(l2_rel(pg.grad, q.grad) for pg, q in zip(Dg.parameters(), Do.parameters()) if q.grad.abs().max() > 0)
        rows.append("  [oracle on device vs oracle on host] worst grad l2_rel %.3e" % worst)
    # The generator's sample against the SAME network evaluated in float64 (round 6; profiles/r06_g_margin.log): the fp32
    # oracle is itself 1.3e-5 .. 1.5e-4 away from that value (its last block amplifies round-off 20-fold: 3 channels, tanh;
    # at t_size = 256 with 2 samples per BatchNorm batch the fp32 oracle alone exceeds the 1e-4 north-star tolerance, and its
    # result moves by ~1e-4 with the host's thread count), so the distance HIP <-> fp32 oracle mostly measures the oracle.
    # What is asserted: the HIP path is within the north-star tolerance of the exact (float64) value, and wherever the fp32
    # oracle is a meaningful reference at 1e-4 (its own error below half of it) the direct comparison holds too.
    import copy
    Go64 = copy.deepcopy(Go).double()
    Go64.A = [a_.double() for a_ in Go64.A]
    with torch.no_grad():
        f64 = Go64(z.double(), labels, noise=[t.double() for t in noise])
    e_hip64, e_o64, e_direct = rel_err(fake_hip, f64), rel_err(ro["fake"], f64), rel_err(r["fake"], ro["fake"])
    rows.append("  fake vs float64 oracle: HIP %.3e | fp32 oracle %.3e | HIP vs fp32 oracle %.3e" % (e_hip64, e_o64, e_direct))
    _log("\n".join(rows))
    assert e_hip64 < FWD_TOL, e_hip64
    if e_o64 < 0.5 * FWD_TOL:
        assert e_direct < FWD_TOL, e_direct
    for k in ("real_validity", "fake_validity"):
        assert rel_err(r[k], ro[k]) < FWD_TOL, k
    assert rel_err(r["gradient_penalty"], ro["gradient_penalty"]) < 5e-4
    assert rel_err(r["d_loss"], ro["d_loss"]) < 5e-4
    assert not bad, bad


def test_properties_at_full_size():
    """Size-independent properties at C2 size (bs=64): linearity of a D block in its input below the
    activation is not observable, so check (i) batch independence - D has no batch-coupled op, so each
    sample's validity is unchanged by what else is in the batch; (ii) D(cat(a,b)) == cat(D(a), D(b));
    (iii) permutation equivariance over the batch."""
    d = dev()
    c, G, D, Go, Do = build_pair("ntu", d)
    real, labels, z, alpha = rand_inputs(64, 3, 64, 25, 60, 512, seed=21, device=d)
    full = D(real, labels)
    half = torch.cat((D(real[:32], labels[:32]), D(real[32:], labels[32:])))
    assert rel_err(full, half) < 1e-5
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(d)
    assert rel_err(D(real[perm], labels[perm]), full[perm]) < 1e-5
    one = D(real[5:6], labels[5:6])
    assert rel_err(one, full[5:6]) < 1e-5


def _bucket_slices(flat_params, module):
    """(name, gradient-bucket slice viewed like the parameter) for every parameter of a FlatParams-managed module"""
    names = [k for k, _ in module.named_parameters()]
    assert len(names) == len(flat_params.views)
    return [(k, v.view(p.shape)) for k, v, p in zip(names, flat_params.views, flat_params.params)]


def _compare_bucket(tag, flat_params, module, ref_grads, rows, tol=GRAD_TOL, skip=None):
    """``skip``: predicate on the parameter name - the generator's conv biases in front of a train-mode BatchNorm have an
    analytically ZERO gradient (the batch mean removes them): both sides are round-off of size 1e-8 there."""
    bad = []
    for k, g in _bucket_slices(flat_params, module):
        if skip is not None and skip(k):
            continue
        q = ref_grads[k]
        e = l2_rel(g, q)
        rows.append("  %s grad %-44s l2_rel %.3e  max|ref| %.3e" % (tag, k, e, q.abs().max().item()))
        if not grad_close(g, q, tol):
            bad.append((tag, k, e))
    return bad


def _graph_of(fn):
    """fn captured in a hipGraph the way bench.py does it (allocator warm-up on a side stream, thread-local capture)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    torch.cuda.synchronize()
    return g


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("cfg,n", [("ntu", 64), ("h36m", 64), ("ntu", 5)])
def test_bench_path_fused_and_staged_generator_blocks_vs_oracle(cfg, n, fused, monkeypatch):
    """test_bench_path_vs_oracle with the generator's last blocks explicitly on either form: ONE launch per block and direction
    (gen_trunk.FUSED: kg_genblock_fwd / kg_genblock_bwd, the default) or the staged launch sequence (KG_GEN_FUSED=0) - eagerly
    and replayed from a hipGraph, against the host oracle; and that the fused launches really ran / did not run."""
    from kinetic_gan_amd import _native as nv_, gen_trunk
    monkeypatch.setattr(gen_trunk, "FUSED", fused)
    calls = {"f": 0, "b": 0}
    f0, b0 = nv_.genblock_fwd, nv_.genblock_bwd
    monkeypatch.setattr(nv_, "genblock_fwd", lambda *a, **k: (calls.__setitem__("f", calls["f"] + 1), f0(*a, **k))[1])
    monkeypatch.setattr(nv_, "genblock_bwd", lambda *a, **k: (calls.__setitem__("b", calls["b"] + 1), b0(*a, **k))[1])
    _bench_path_vs_oracle(cfg, n)
    assert (calls["f"] > 0 and calls["b"] > 0) if fused else (calls["f"] == 0 and calls["b"] == 0), calls


@pytest.mark.parametrize("cfg,n", [("ntu", 64), ("ntu120", 32), ("h36m", 64), ("stress", 2), ("ntu", 5), ("h36m", 3)])
def test_bench_path_vs_oracle(cfg, n):
    _bench_path_vs_oracle(cfg, n)


def _bench_path_vs_oracle(cfg, n):
    """The composition bench.py times - ``Trainer(G, D)``: flat buckets, merged 3n critic backward with the promised
    gradient, parameter-gradient sinks, deferred kg_wgrad_many / kg_rowsum_many launches, paired 2n synthesis - at
    the BASELINE configs' real batch sizes, eagerly AND replayed from a hipGraph, against the host oracle:
      critic step   : losses and EVERY parameter's slice of the D gradient bucket (D is fed the oracle's fake sample
                      so that LeakyReLU kinks of near-zero activations do not blur the gradient comparison);
      generator step: g_loss and EVERY parameter's slice of the G gradient bucket (kinetic-gan.py:167-173), the sample
                      synthesised next to the critic's as one 2n batch (sharing_mapping), D unchanged in between."""
    d = dev()
    c, G, D, Go, Do = build_pair(cfg, d)
    nn_ = G.graph.num_node
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=11)
    noise_d = rand_noise(n, c["t_size"], nn_, seed=12)
    noise_g = rand_noise(n, c["t_size"], nn_, seed=13)
    from oracle.host import usable_cores
    torch.set_num_threads(usable_cores())
    ro = M.d_step_losses(Go, Do, real, labels, z, alpha, noise=noise_d)
    Do.zero_grad()
    Go.zero_grad()
    ro["d_loss"].backward()
    ref_d = {k: p.grad.detach().clone() for k, p in Do.named_parameters()}
    Go.zero_grad()
    rg = M.g_step_loss(Go, Do, labels, z, noise=noise_g)
    rg["g_loss"].backward()
    ref_g = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in Go.named_parameters()}

    to = lambda t: t.to(d)
    real_d, labels_d, z_d, alpha_d = to(real), to(labels), to(z), to(alpha)
    nd, ng = [to(t) for t in noise_d], [to(t) for t in noise_g]
    ofake = to(ro["fake"].detach())
    ref_gp = _oracle_gp_grads(Do, real, ro["fake"].detach(), labels, alpha)
    tr = Trainer(G, D)                      # exactly bench.py's construction
    assert tr._promise and tr.fD is not None and getattr(D, "use_trunk", False)
    rows = ["bench path %s n=%d" % (cfg, n)]
    bad = []
    keep = {}

    def critic():
        tr.d_compute(real_d, labels_d, z_d, alpha_d, nd, fake=ofake, keep=keep)

    def both_steps():
        with tr.sharing_mapping(ng):
            tr.d_compute(real_d, labels_d, z_d, alpha_d, nd)
        keep["g_loss"] = tr.g_compute(labels_d, z_d, ng)

    for mode in ("eager", "graph"):
        if mode == "eager":
            critic()
        else:
            gr = _graph_of(critic)
            tr.fD.grad.fill_(float("nan"))      # the replay must rebuild the whole bucket
            gr.replay()
        torch.cuda.synchronize()
        for k in ("real_validity", "fake_validity"):
            assert rel_err(keep[k], ro[k]) < FWD_TOL, (mode, k)
        assert rel_err(keep["gradient_penalty"], ro["gradient_penalty"]) < 5e-4, mode
        assert rel_err(keep["d_loss"], ro["d_loss"]) < 5e-4, mode
        # the penalty's first-order gradient d D(inter) / d inter of all n samples: relative L2 (one LeakyReLU kink that
        # falls on the other side in ONE sample moves that sample's gradient visibly; a max-norm bound at n = 64 would
        # measure that, not the kernels)
        e_gp = l2_rel(keep["gp_grads"], ref_gp)
        rows.append("  %s gp_grads l2_rel %.3e" % (mode, e_gp))
        assert e_gp < 2e-2, (mode, e_gp)
        bad += _compare_bucket(mode + " D", tr.fD, D, ref_d, rows)
        if mode == "eager":
            both_steps()
        else:
            gr2 = _graph_of(both_steps)
            tr.fG.grad.fill_(float("nan"))
            gr2.replay()
        torch.cuda.synchronize()
        rows.append("  %s g_loss rel_err %.3e" % (mode, rel_err(keep["g_loss"], rg["g_loss"])))
        assert rel_err(keep["g_loss"], rg["g_loss"]) < 2e-4, mode
        bad += _compare_bucket(mode + " G", tr.fG, G, ref_g, rows, skip=_zero_grad_keys)
    _log("\n".join(rows))
    assert not bad, bad


def test_c5b_full_size_batch_properties():
    """C5b (full G / D at t_size = 256, NTU graph) at the 64 samples per GPU that bench.py --config stress times.  The host
    oracle needs minutes for that batch, so the full size is checked through properties that do not depend on it
    (the oracle comparison of the same configuration runs at n = 2, test_full_size_vs_oracle / test_bench_path_vs_oracle):
      * the critic has no batch coupling: D(x) of the 64-sample batch == cat(D(first half), D(second half)), although
        the two run different tile plans, K-splits and aggregation kernels;
      * split = cat for the gradients: with a loss that is a SUM over samples, every parameter gradient and the input
        gradient of the 64-sample backward pass equal the sum / concatenation of the two halves';
      * the generator in eval mode (BatchNorm folded, no batch statistics) is batch independent as well;
      * the WGAN-GP critic losses of the bench path (Trainer, 3n merged passes) are finite and the penalty's input
        gradient is batch independent: the first half's rows do not change when the second half changes."""
    d = dev()
    c, G, D, _, _ = build_pair("stress", d)
    nn_ = G.graph.num_node
    n, h = 64, 32
    real, labels, z, alpha = (t.to(d) for t in rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=21))
    noise = [t.to(d) for t in rand_noise(n, c["t_size"], nn_, seed=22)]
    rows = ["C5b n=64 batch properties"]

    def critic(x, lab):
        x = x.clone().requires_grad_(True)
        D.zero_grad()
        out = D(x, lab)
        out.sum().backward()
        return out.detach(), x.grad.detach(), {k: p.grad.detach().clone() for k, p in D.named_parameters()}
    o_f, gx_f, gp_f = critic(real, labels)
    o_a, gx_a, gp_a = critic(real[:h], labels[:h])
    o_b, gx_b, gp_b = critic(real[h:], labels[h:])
    e = rel_err(torch.cat([o_a, o_b]), o_f)
    rows.append("  D forward split=cat rel_err %.2e" % e)
    assert e < FWD_TOL        # (the two batch sizes take different tiles / K-splits in all six blocks: 3.9e-5 measured)
    # (round 6: the plan no longer takes the bf16-split tail by itself, so the whole batch and its halves run the same fp32
    # kernels up to tile / K-split choices and the round-4 aggregate bound holds again; per-sample figures are logged - a
    # LeakyReLU kink flip shows there as ONE sample near 1e-3 with every other sample of its half bit-identical)
    gx_s = torch.cat([gx_a, gx_b])
    es = ((gx_s - gx_f).flatten(1).norm(dim=1) / gx_f.flatten(1).norm(dim=1).clamp_min(1e-30)).cpu()
    e = l2_rel(gx_s, gx_f)
    rows.append("  D input gradient split=cat l2 %.2e; per sample: median %.2e, max %.2e, samples above 1e-4: %d" % (
        e, es.median().item(), es.max().item(), int((es > 1e-4).sum())))
    assert e < 1e-4, (e, es)
    assert es.max().item() < 2e-2, es
    bad = []
    for k in gp_f:
        ek = l2_rel(gp_a[k] + gp_b[k], gp_f[k])
        rows.append("  D %-44s sum of halves l2 %.2e" % (k, ek))
        if not grad_close(gp_a[k] + gp_b[k], gp_f[k], 1e-4):
            bad.append((k, ek))
    assert not bad, bad
    G.eval()
    with torch.no_grad():
        f_f = G(z, labels, noise=noise)
        f_a = G(z[:h], labels[:h], noise=[t[:h] for t in noise])
        f_b = G(z[h:], labels[h:], noise=[t[h:] for t in noise])
    G.train()
    e = rel_err(torch.cat([f_a, f_b]), f_f)
    rows.append("  G eval forward split=cat rel_err %.2e" % e)
    assert e < FWD_TOL
    tr = Trainer(G, D)
    fake = f_f.detach()
    r1 = tr.d_losses(real, labels, z, alpha, noise, fake=fake)
    assert all(torch.isfinite(r1[k]).all() for k in ("real_validity", "fake_validity", "gradient_penalty", "d_loss"))
    real2 = real.clone()
    real2[h:] = real2[h:].flip(0)                   # another second half: the first half's validities must not move
    r2 = tr.d_losses(real2, labels, z, alpha, noise, fake=fake)
    assert torch.equal(r1["real_validity"][:h], r2["real_validity"][:h])
    assert not torch.equal(r1["real_validity"][h:], r2["real_validity"][h:])
    _log("\n".join(rows))


def _oracle_gp_grads(Do, real, fake, labels, alpha):
    inter = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    out = Do(inter, labels)
    (g,) = torch.autograd.grad(out, inter, torch.ones_like(out))
    return g


def test_trainer_iteration_on_gpu_matches_host_oracle():
    """Two full iterations (critic step + Adam, generator step + Adam) of the bench path against the host oracle with
    torch.optim.Adam: per parameter the gradient bucket of each iteration, Adam's moment buffers and the parameter
    UPDATE (Adam's first steps move every weight by about +-lr whatever the gradient's size, so a bound on the
    parameter difference alone would say nothing: the moments carry the magnitudes, the update its direction)."""
    d = dev()
    c, G, D, Go, Do = build_pair("h36m", d)
    nn_ = G.graph.num_node
    n = 8
    lr = 2e-4
    real, labels, z, alpha = rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3)
    noise = rand_noise(n, c["t_size"], nn_, seed=6)
    oG = torch.optim.Adam(Go.parameters(), lr=lr, betas=(0.5, 0.999))
    oD = torch.optim.Adam(Do.parameters(), lr=lr, betas=(0.5, 0.999))
    tr = Trainer(G, D, fused_step=False)        # (the gradient buckets are read after the optimiser steps)
    # the default trainer - optimiser launches that clear the bucket they consumed (kg_adam_step_fused), zero_grad without
    # a fill - on twin models: must end bit-identical
    _, G_f, D_f, _, _ = build_pair("h36m", d)
    tr_f = Trainer(G_f, D_f)
    assert tr_f.fD.fused_step and tr_f.fG.fused_step
    to = lambda t: t.to(d)
    nd = [to(t) for t in noise]
    from oracle.host import usable_cores
    torch.set_num_threads(usable_cores())
    start = {id(q): q.detach().clone() for q in list(Do.parameters()) + list(Go.parameters())}
    rows = ["trainer iterations h36m n=8"]
    bad = []
    for it in range(2):
        tol = GRAD_TOL if it == 0 else 2e-2       # iteration 2 starts from parameters that differ by Adam round-off
        tr.iteration(to(real), to(labels), to(z), to(alpha), nd, nd, with_g=True)
        tr_f.iteration(to(real), to(labels), to(z), to(alpha), nd, nd, with_g=True)
        for fa, fb in ((tr.fD, tr_f.fD), (tr.fG, tr_f.fG)):
            assert torch.equal(fa.flat, fb.flat) and torch.equal(fa.exp_avg_sq, fb.exp_avg_sq)
            assert int(fa.step) == int(fb.step) == it + 1 and float(fb.grad.abs().max()) == 0.0 and fb._clean
        oD.zero_grad()
        M.d_step_losses(Go, Do, real, labels, z, alpha, noise=noise)["d_loss"].backward()
        ref_d = {k: p.grad.detach().clone() for k, p in Do.named_parameters()}
        oD.step()
        oG.zero_grad()
        M.g_step_loss(Go, Do, labels, z, noise=noise)["g_loss"].backward()
        ref_g = {k: p.grad.detach().clone() for k, p in Go.named_parameters()}
        oG.step()
        bad += _compare_bucket("it%d D" % it, tr.fD, D, ref_d, rows, tol)
        bad += _compare_bucket("it%d G" % it, tr.fG, G, ref_g, rows, tol, skip=_zero_grad_keys)
    for fp, mod, ref_mod, opt in ((tr.fD, D, Do, oD), (tr.fG, G, Go, oG)):
        for (k, p), q, off in zip(mod.named_parameters(), ref_mod.parameters(), fp.offsets):
            nel = p.numel()
            st = opt.state[q]
            m = fp.exp_avg[off:off + nel].view(p.shape)
            v = fp.exp_avg_sq[off:off + nel].view(p.shape)
            if _zero_grad_keys(k):
                continue
            if not grad_close(m, st["exp_avg"], 2e-2):
                bad.append(("exp_avg", k, l2_rel(m, st["exp_avg"])))
            if not grad_close(v, st["exp_avg_sq"], 4e-2, floor=1e-12):
                bad.append(("exp_avg_sq", k, l2_rel(v, st["exp_avg_sq"])))
            # update direction: where the first moment is not noise, both sides moved the weight the same way
            dp, dq = p.detach().cpu() - start[id(q)], q.detach() - start[id(q)]
            sig = st["exp_avg"].abs() > 1e-2 * st["exp_avg"].abs().max()
            if sig.any():
                miss = ((dp - dq).abs() > 0.5 * lr)[sig].float().mean().item()
                rows.append("  update %-44s moved %.2e  mismatched %.4f" % (k, dq.abs().max().item(), miss))
                if miss > 0.01:
                    bad.append(("update", k, miss))
    # the generator's BatchNorm buffers after two iterations = four train-mode syntheses (kinetic-gan.py:143,167; the
    # product runs each iteration's two as ONE paired 2n batch with per-half statistics updated in the reference's
    # order): running_mean / running_var / num_batches_tracked against the oracle's (round-3 VERDICT, weak 1)
    sd, sdo = G.state_dict(), Go.state_dict()
    nbuf = 0
    for k in sdo:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(sdo[k]) == 4, (k, int(sd[k]), int(sdo[k]))
            nbuf += 1
        elif "running_" in k:
            e = rel_err(sd[k], sdo[k])
            rows.append("  buffer %-44s rel_err %.2e" % (k, e))
            if e > 2e-3:      # (the second iteration's statistics come from parameters that differ by Adam round-off)
                bad.append(("buffer", k, e))
            nbuf += 1
    assert nbuf == 3 * 8, nbuf                      # 8 BatchNorm layers: tcn.1 of blocks 1, 3, 5 and residual.1 of blocks 1..5
    _log("\n".join(rows))
    assert not bad, bad


def test_stress_block_c5a_shapes():
    """BASELINE configs[4] kernel-level shape: one D-style block, 512 -> 512 channels, NTU level-0
    adjacency (V=25), T=256, identity residual, no down-sampling - at N=2 so that the host oracle finishes in seconds;
    forward, input gradient and every parameter gradient."""
    import kinetic_gan_amd.discriminator as KD
    from kinetic_gan_amd.graph import graph_ntu
    from oracle.host import usable_cores
    torch.set_num_threads(usable_cores())
    d = dev()
    g = graph_ntu()
    ks = ([3, 3, 3, 3], [3, 3, 3, 3])
    blk = KD.st_gcn(512, 512, ks, 1, graph=g, lvl=0, dw_s=False, dw_t=256).to(d)
    ref = M.DiscBlock(512, 512, ks, 1, graph=g, lvl=0, dw_s=False, dw_t=256)
    fill_module(ref, seed=7)
    blk.load_state_dict(ref.state_dict())
    A = torch.tensor(g.As[0], dtype=torch.float32) * (0.5 + torch.rand(3, 25, 25, generator=torch.Generator().manual_seed(3)))
    x = block_input((2, 512, 256, 25), 900)
    x1 = x.to(d).requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    y1, _ = blk(x1, A.to(d))
    y2, _ = ref(x2, A)
    assert rel_err(y1, y2) < FWD_TOL
    go = torch.randn(y2.shape, generator=torch.Generator().manual_seed(4))
    y1.backward(go.to(d))
    y2.backward(go)
    assert grad_close(x1.grad, x2.grad, GRAD_TOL)
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert grad_close(p.grad, q.grad, GRAD_TOL), (k, l2_rel(p.grad, q.grad))


def test_c5a_roofline_launch_values():
    """The 64-sample C5a launch bench.py times (512 -> 512 channels, 3 temporal taps + identity residual + LeakyReLU,
    T=256, V=25) checked for VALUES against the plain-torch definition evaluated on the device, and its standalone
    aggregation (64, 3*512, 256, 25) -> (64, 512, 256, 25) likewise."""
    from kinetic_gan_amd import _native as nv
    from kinetic_gan_amd._native import TAP_TIME, Group, WView
    from oracle import prim_ref as pr
    d = dev()
    n, c, T, V = 64, 512, 256, 25
    g = torch.Generator(device=d).manual_seed(5)
    z = nv.new_plane(n, c, T, V, d).normal_(generator=g)
    x = nv.new_plane(n, c, T, V, d).normal_(generator=g)
    wt = torch.randn(c, c, 3, 1, device=d, generator=g) * 0.02
    bt = torch.randn(c, device=d, generator=g)
    grp = Group(z, wt, WView(1, c * 3, 3), c, 3, TAP_TIME, 1, False, None)
    out = nv.conv([grp], n, c, T, V, bias0=bt, add=x, act=nv.ACT_LRELU)
    ref = pr.conv([grp], n, c, T, V, bias0=bt, add=x, act=nv.ACT_LRELU)
    assert rel_err(out, ref) < 2e-5
    del out, ref
    y = nv.new_plane(n, 3 * c, T, V, d).normal_(generator=g)
    A = torch.rand(3, V, V, device=d, generator=g)
    assert rel_err(nv.agg_reduce(y, A, 1), pr.agg_reduce(y, A, 1)) < 2e-5


def test_folded_inference_follows_graph_replayed_training():
    """train (hipGraph replay: no Python runs, Adam / BatchNorm statistics written through raw pointers) -> sample ->
    train -> sample: the folded eval / no_grad forward equals the unfolded one on the CURRENT parameters every time."""
    d = dev()
    c, G, D, _, _ = build_pair("h36m", d)
    nn_ = G.graph.num_node
    n = 8
    real, labels, z, alpha = (t.to(d) for t in rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3))
    noise = [t.to(d) for t in rand_noise(n, c["t_size"], nn_, seed=6)]
    tr = Trainer(G, D)
    gr = _graph_of(lambda: tr.iteration(real, labels, z, alpha, noise, noise, with_g=True))
    outs = []
    for rnd_ in range(3):
        G.eval()
        with torch.no_grad():
            folded = G(z, labels, noise=noise)
        unfolded = G(z, labels, noise=noise).detach()
        assert rel_err(folded, unfolded) < 1e-5, rnd_
        outs.append(folded.clone())
        G.train()
        gr.replay()
        torch.cuda.synchronize()
    assert rel_err(outs[1], outs[0]) > 1e-4 and rel_err(outs[2], outs[1]) > 1e-4


def test_training_loop_on_feeder_batches_follows_host_oracle(tmp_path):
    """The loop of kinetic-gan.py:125-192 end to end on the committed feeder files: DeviceBatches (pinned staging, side
    stream, on-device normalisation) -> critic step every batch, generator step every n_critic-th (kinetic-gan.py:152)
    -> checkpoints off the training stream; against the host oracle's loop (reference Feeder arithmetic, stock modules,
    torch.optim.Adam) on the same batches, latents, interpolation factors and noise: losses batch by batch, and the
    final checkpoint loaded into the oracle's modules."""
    from kinetic_gan_amd.checkpoint import AsyncCheckpointWriter
    from kinetic_gan_amd.discriminator import Discriminator
    from kinetic_gan_amd.feeder import DeviceBatches, Feeder
    from kinetic_gan_amd.generator import Generator
    from oracle.host import usable_cores
    torch.set_num_threads(usable_cores())
    d = dev()
    g = os.path.join(os.path.dirname(__file__), "golden")
    f = Feeder(os.path.join(g, "feeder_ntu_data.npy"), os.path.join(g, "feeder_ntu_label.pkl"), dataset="ntu")
    t_size, bs, n_critic, lr = 16, 4, 2, 2e-4
    G = Generator(512, 3, 60, t_size, 4, dataset="ntu")
    D = Discriminator(3, 60, t_size, 512, dataset="ntu")
    Go = M.Generator(512, 3, 60, t_size, 4, dataset="ntu")
    Do = M.Discriminator(3, 60, t_size, 512, dataset="ntu")
    for m, sd in ((G, 1), (Go, 1), (D, 2), (Do, 2)):
        fill_module(m, seed=sd)
    G, D = G.to(d), D.to(d)
    nn_ = G.graph.num_node
    oG = torch.optim.Adam(Go.parameters(), lr=lr, betas=(0.5, 0.999))
    oD = torch.optim.Adam(Do.parameters(), lr=lr, betas=(0.5, 0.999))
    tr = Trainer(G, D, n_critic=n_critic)
    w = AsyncCheckpointWriter()
    gen = torch.Generator().manual_seed(11)
    rows, step = ["training loop on feeder batches (ntu fixture, t_size 16, bs 4)"], 0
    for epoch in range(2):
        for real, labels in DeviceBatches(f, bs, t_size, d, seed=7):
            z = torch.randn(bs, 512, generator=gen)
            alpha = torch.rand(bs, 1, 1, 1, generator=gen)
            noise = rand_noise(bs, t_size, nn_, seed=100 + step)
            with_g = step % n_critic == 0
            nd = [t.to(d) for t in noise]
            d_loss, g_loss = tr.iteration(real, labels, z.to(d), alpha.to(d), nd, nd, with_g=with_g)
            rc, lc = real.cpu(), labels.cpu()
            oD.zero_grad()
            ro = M.d_step_losses(Go, Do, rc, lc, z, alpha, noise=noise)
            ro["d_loss"].backward()
            oD.step()
            e_d = abs(float(d_loss) - float(ro["d_loss"])) / max(1e-6, abs(float(ro["d_loss"])))
            line = "  step %d d_loss %.5f (oracle %.5f)" % (step, float(d_loss), float(ro["d_loss"]))
            assert e_d < 2e-2, line
            if with_g:
                oG.zero_grad()
                go = M.g_step_loss(Go, Do, lc, z, noise=noise)
                go["g_loss"].backward()
                oG.step()
                e_g = abs(float(g_loss) - float(go["g_loss"])) / max(1e-6, abs(float(go["g_loss"])))
                line += "  g_loss %.5f (oracle %.5f)" % (float(g_loss), float(go["g_loss"]))
                assert e_g < 2e-2, line
            rows.append(line)
            step += 1
        w.save(G, str(tmp_path / ("generator_%d.pth" % step)))
        w.save(D, str(tmp_path / ("discriminator_%d.pth" % step)))
    w.close()
    assert step == 6
    _log("\n".join(rows))
    # the last checkpoint is the trained state: strict load into fresh oracle modules, same critic values as the oracle's
    # own trained critic on a batch (the two trajectories differ by Adam round-off only)
    Dn = M.Discriminator(3, 60, t_size, 512, dataset="ntu")
    Dn.load_state_dict(torch.load(str(tmp_path / ("discriminator_%d.pth" % step))), strict=True)
    with torch.no_grad():
        assert rel_err(Dn(rc, lc), Do(rc, lc)) < 2e-2
    # ... and the generator's checkpoint (weights + BatchNorm running statistics of the paired-synthesis path: six critic
    # steps + three generator steps = nine train-mode syntheses) against the oracle's trained generator
    Gn = M.Generator(512, 3, 60, t_size, 4, dataset="ntu")
    gsd = torch.load(str(tmp_path / ("generator_%d.pth" % step)))
    Gn.load_state_dict(gsd, strict=True)
    gso = Go.state_dict()
    for k in gso:
        if k.endswith("num_batches_tracked"):
            assert int(gsd[k]) == int(gso[k]) == 9, (k, int(gsd[k]), int(gso[k]))
        elif "running_" in k:
            assert rel_err(gsd[k], gso[k]) < 2e-2, (k, rel_err(gsd[k], gso[k]))
    Gn.eval(); Go.eval()
    with torch.no_grad():
        ze = torch.randn(4, 512, generator=gen)
        le = torch.randint(0, 60, (4,), generator=gen)
        ne = rand_noise(4, t_size, nn_, seed=999)
        # (three Adam steps apart: a weight whose gradient is round-off moves +-lr per step on either side - 3.4e-2
        # measured; an un-trained or wrongly loaded generator is off by O(1))
        e_out = rel_err(Gn(ze, le, noise=ne), Go(ze, le, noise=ne))
        _log("  trained generator (checkpoint vs oracle), eval forward rel_err %.3e" % e_out)
        assert e_out < 8e-2


def test_async_checkpoint_between_replayed_iterations(tmp_path):
    """checkpoint.AsyncCheckpointWriter on the GPU: a snapshot taken between two graph-replayed iterations holds the
    parameters / BatchNorm buffers of exactly that moment (the copy is ordered behind the first replay on a side stream;
    the second replay does not wait for the file) and loads strictly into a fresh model."""
    from kinetic_gan_amd.checkpoint import AsyncCheckpointWriter
    d = dev()
    c, G, D, Go, _ = build_pair("h36m", d)
    nn_ = G.graph.num_node
    n = 8
    real, labels, z, alpha = (t.to(d) for t in rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3))
    noise = [t.to(d) for t in rand_noise(n, c["t_size"], nn_, seed=6)]
    tr = Trainer(G, D)
    gr = _graph_of(lambda: tr.iteration(real, labels, z, alpha, noise, noise, with_g=True))
    w = AsyncCheckpointWriter()
    gr.replay()
    w.save(G, str(tmp_path / "generator_1.pth"))
    torch.cuda.synchronize()
    want = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    # tearing check (round-3 ADVICE): a twin model runs the same two iterations and is read back after a device
    # synchronisation; the writer's file, taken between iteration 2 and an iteration 3 enqueued right behind save(),
    # must equal the twin key by key - BatchNorm buffers included, which the third replay rewrites within its first
    # few hundred microseconds
    c2, G2, D2, _, _ = build_pair("h36m", d)
    tr2 = Trainer(G2, D2)
    gr2 = _graph_of(lambda: tr2.iteration(real, labels, z, alpha, noise, noise, with_g=True))
    gr2.replay(); gr2.replay()
    torch.cuda.synchronize()
    twin2 = {k: v.detach().cpu().clone() for k, v in G2.state_dict().items()}
    twin2_d = {k: v.detach().cpu().clone() for k, v in D2.state_dict().items()}
    w2 = AsyncCheckpointWriter()
    gr.replay()                                            # state after iteration 2 ...
    w2.save(G, str(tmp_path / "generator_2a.pth"))         # ... snapshot enqueued behind it,
    w2.save(D, str(tmp_path / "discriminator_2a.pth"))
    gr.replay()                                            # and a third iteration right behind the snapshot
    w.close(); w2.close()
    torch.cuda.synchronize()
    got = torch.load(str(tmp_path / "generator_1.pth"))
    assert list(got.keys()) == list(want.keys())
    assert all(torch.equal(got[k], want[k]) for k in want)
    Go.load_state_dict(got, strict=True)
    after3 = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    got2 = torch.load(str(tmp_path / "generator_2a.pth"))
    got2_d = torch.load(str(tmp_path / "discriminator_2a.pth"))
    assert list(got2.keys()) == list(twin2.keys())
    bad = [k for k in twin2 if not torch.equal(got2[k], twin2[k])] + [k for k in twin2_d if not torch.equal(got2_d[k], twin2_d[k])]
    assert not bad, "torn / wrong snapshot: " + ", ".join(bad[:8])
    k0 = next(k for k in want if k.endswith("weight") and want[k].numel() > 1000)
    assert not torch.equal(got2[k0], want[k0]) and not torch.equal(got2[k0], after3[k0])     # iteration 2's state, neither 1's nor 3's
    moved = [k for k in twin2 if "running_" in k or "num_batches" in k]
    assert moved and all(not torch.equal(after3[k], twin2[k]) for k in moved)                # the third replay did rewrite them


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
@pytest.mark.parametrize("comm", ["torch", "kg"])
def test_two_rank_nccl_step_matches_single_process(comm):
    """Data parallel over RCCL when two devices are visible: two ranks (one process per GPU, backend nccl), each with
    its shard, must end up with the parameters of one process that averaged the two shards' gradients itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KG_DP_BACKEND="nccl", KG_DP_COMM=comm, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "tests", "dp_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("segmented", [False, True, "overlap"])
def test_hipgraph_replay_matches_eager(segmented):
    """The captured iteration (whole step, or the two compute halves with eager apply halves as data parallel runs do)
    replays the same arithmetic as eager launches: with pinned noise the parameters after two iterations are
    bit-identical.  Exercises the graph-safety of the host logic (gradient sink, deferred weight gradients, shared
    adjacencies, device-side step counter)."""
    d = dev()
    c, G, D, _, _ = build_pair("h36m", d)
    c2, G2, D2, _, _ = build_pair("h36m", d)
    nn_ = G.graph.num_node
    n = 8
    real, labels, z, alpha = (t.to(d) for t in rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3))
    noise = [t.to(d) for t in rand_noise(n, c["t_size"], nn_, seed=6)]
    ta, tb = Trainer(G, D), Trainer(G2, D2, overlap=(segmented == "overlap"))
    for _ in range(2):
        if segmented:       # the halves captured separately below do not pair the two generator syntheses: same here
            ta.d_step(real, labels, z, alpha, noise)
            ta.g_step(labels, z, noise)
        else:
            ta.iteration(real, labels, z, alpha, noise, noise, with_g=True)

    def snapshot(tr):
        return ([t.clone() for t in (tr.fD.flat, tr.fD.exp_avg, tr.fD.exp_avg_sq, tr.fD.step,
                                     tr.fG.flat, tr.fG.exp_avg, tr.fG.exp_avg_sq, tr.fG.step)],
                {k: v.clone() for k, v in tr.G.state_dict().items() if "running_" in k or "num_batches" in k})

    def restore(tr, snap):
        bufs, stats = snap
        for dst, src in zip((tr.fD.flat, tr.fD.exp_avg, tr.fD.exp_avg_sq, tr.fD.step,
                             tr.fG.flat, tr.fG.exp_avg, tr.fG.exp_avg_sq, tr.fG.step), bufs):
            dst.copy_(src)
        sd = tr.G.state_dict()
        for k, v in stats.items():
            sd[k].copy_(v)

    snap = snapshot(tb)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # allocator warm-up, then back to the initial state
        tb.iteration(real, labels, z, alpha, noise, noise, with_g=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    restore(tb, snap)
    if not segmented:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            tb.iteration(real, labels, z, alpha, noise, noise, with_g=True)
        restore(tb, snap)                               # capture does not execute, but stay on the safe side
        for _ in range(2):
            g.replay()
    elif segmented == "overlap":
        # the data-parallel launch structure with D's apply half on the side stream under the G forward: the
        # generator step is two graphs (G forward | D forward + backward + gather) sharing one pool
        gd, ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gd, capture_error_mode="thread_local"):
            tb.d_compute(real, labels, z, alpha, noise)
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            fake = tb.g_forward(labels, z, noise)
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            tb.g_backward(fake, labels)
        del fake
        restore(tb, snap)
        for _ in range(2):
            gd.replay(); tb.d_apply_async(); ga.replay(); tb.wait_d_apply(); gb.replay(); tb.g_apply()
    else:
        gd, gg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gd, capture_error_mode="thread_local"):
            tb.d_compute(real, labels, z, alpha, noise)
        with torch.cuda.graph(gg, capture_error_mode="thread_local"):
            tb.g_compute(labels, z, noise)
        restore(tb, snap)
        for _ in range(2):
            gd.replay(); tb.d_apply(); gg.replay(); tb.g_apply()
    torch.cuda.synchronize()
    assert torch.equal(ta.fD.flat, tb.fD.flat)
    assert torch.equal(ta.fG.flat, tb.fG.flat)
    for k, v in ta.G.state_dict().items():
        if "running_" in k or "num_batches" in k:
            assert torch.equal(v, tb.G.state_dict()[k]), k


def test_comm_c_abi_single_rank_is_callable_eagerly_and_under_capture():
    """kg_comm_init / kg_allreduce_flat / kg_comm_destroy on the one GPU of the test box: a one-rank communicator whose
    in-place sum all-reduce leaves the bucket unchanged, called eagerly and under stream capture, and a Trainer driven
    through it ends up bit-identical to one without.  What this does NOT show (round-3 VERDICT, weak 3): RCCL elides a
    one-rank in-place all-reduce, so the captured graph is EMPTY - no RCCL kernel is recorded here; the call is merely
    legal under capture.  Recording a real collective needs two devices
    (test_two_rank_nccl_step_matches_single_process, KG_DP_COMM=kg)."""
    from kinetic_gan_amd import _native as nv
    d = dev()
    comm = nv.Comm(0, 1, 0)
    comm.force = True
    try:
        x = torch.randn(1 << 20, device=d)
        ref = x.clone()
        comm.allreduce_(x)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        g = _graph_of(lambda: comm.allreduce_(x))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        c, G, D, _, _ = build_pair("h36m", d)
        c2, G2, D2, _, _ = build_pair("h36m", d)
        n = 4
        nn_ = G.graph.num_node
        real, labels, z, alpha = (t.to(d) for t in rand_inputs(n, c["channels"], c["t_size"], nn_[0], c["n_classes"], c["latent"], seed=3))
        noise = [t.to(d) for t in rand_noise(n, c["t_size"], nn_, seed=6)]
        ta, tb = Trainer(G, D), Trainer(G2, D2, comm=comm)
        for _ in range(2):
            ta.iteration(real, labels, z, alpha, noise, noise, with_g=True)
            tb.iteration(real, labels, z, alpha, noise, noise, with_g=True)
        torch.cuda.synchronize()
        assert torch.equal(ta.fD.flat, tb.fD.flat) and torch.equal(ta.fG.flat, tb.fG.flat)
    finally:
        comm.destroy()


def test_exact_batchnorm_function_on_device():
    """ops.SyncBatchNorm2dFn (the optional exact data-parallel BatchNorm mode) on the GPU with a one-rank RCCL
    group: forward, input / affine gradients and the running-statistics update equal torch's train-mode batch_norm (the
    2-rank equality with a single process on the global batch is tests/test_dp_gloo_cpu.py; round-4 VERDICT: no -m gpu
    test ran this mode)."""
    import socket
    import torch.distributed as dist
    from kinetic_gan_amd import ops
    d = dev()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=d)
    try:
        g = torch.Generator().manual_seed(11)
        x = (torch.randn(6, 32, 16, 11, generator=g) * 1.7 + 0.3).to(d).requires_grad_(True)
        gamma = (torch.rand(32, generator=g) + 0.5).to(d).requires_grad_(True)
        beta = torch.randn(32, generator=g).to(d).requires_grad_(True)
        gout = torch.randn(6, 32, 16, 11, generator=g).to(d)
        rm, rv = torch.zeros(32, device=d), torch.ones(32, device=d)
        nbt = torch.zeros((), dtype=torch.long, device=d)
        y = ops.SyncBatchNorm2dFn.apply(x, gamma, beta, rm, rv, nbt, 0.1, 1e-5, None)
        y.backward(gout)
        x2, g2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, gamma, beta))
        rm2, rv2 = torch.zeros(32, device=d), torch.ones(32, device=d)
        y2 = torch.nn.functional.batch_norm(x2, rm2, rv2, g2, b2, True, 0.1, 1e-5)
        y2.backward(gout)
        assert rel_err(y, y2) < 1e-5
        assert l2_rel(x.grad, x2.grad) < 1e-5 and l2_rel(gamma.grad, g2.grad) < 1e-5 and l2_rel(beta.grad, b2.grad) < 1e-5
        assert rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5 and int(nbt) == 1
    finally:
        if own:
            dist.destroy_process_group()
