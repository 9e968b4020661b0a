"""-m gpu: every libkgan_hip.so entry point against its plain-torch definition (oracle/prim_ref.py)
on the same seeded inputs, through the C ABI.  fp32 tolerance: |a-b| <= 2e-5 * max|b| (fp32 MFMA is
an exact-fp32 FMA chain; only the summation order differs)."""
import os

import pytest
import torch

import kinetic_gan_amd  # noqa: F401
from kinetic_gan_amd import _native as nv
from kinetic_gan_amd._native import TAP_CHANBLOCK, TAP_TIME, Group, WView
from oracle import prim_ref as pr

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture
def monkeypatch(monkeypatch):
    """the library caches its KG_* switches when it is loaded (no getenv on the launch path): every change of the
    environment made through this fixture is followed by kg_reload_env()"""
    class Reloading:
        def setenv(self, k, v):
            monkeypatch.setenv(k, v)
            nv.reload_env()

        def delenv(self, k, raising=True):
            monkeypatch.delenv(k, raising)
            nv.reload_env()

        def __getattr__(self, n):
            return getattr(monkeypatch, n)
    return Reloading()


def pytest_generate_tests(metafunc):
    # only the kg_conv tests have an alternative kernel path: every other test runs once (round-3 VERDICT: the blanket
    # parametrisation produced 134 phantom skips that hid the real ones)
    if "kernel_path" in metafunc.fixturenames:
        name = metafunc.function.__name__
        alt = "conv" in name and "aggconv" not in name
        # "bs": the bf16-split LDS-staged form wherever it can run, on each of its three tiles
        # "ring": the persistent LDS-ring form (tools/probe/kg_conv_ring.hip), only with a `build.py --with-ring` library
        # and KG_TEST_RING=1 - it is not part of the default build (round-5 VERDICT: 741 parametrisations of a kernel
        # no plan selects)
        ring = ["ring%d" % i for i in range(11)] if os.environ.get("KG_TEST_RING", "0") == "1" else []
        metafunc.parametrize("kernel_path", ["default", "alt"] + ring + ["bs%d" % i for i in range(3)]
                             if alt else ["default"], indirect=True)


@pytest.fixture(autouse=True)
def kernel_path(request, monkeypatch):
    """kg_conv tests run twice: with the kernels the launcher picks by default (full-slice instantiation of the tap GEMM,
    streaming kernel for the tiny-channel launches) and with those switched off (general instantiation with per-fragment
    validity, MFMA tiles for every launch)"""
    if request.param == "alt":
        monkeypatch.setenv("KG_CONV_FAST", "0")
        monkeypatch.setenv("KG_CONV_TINY", "0")
    if request.param.startswith("ring"):
        monkeypatch.setenv("KG_CONV_RING", "1")
        monkeypatch.setenv("KG_CONV_RING_TILE", request.param[4:])
    if request.param.startswith("bs"):
        monkeypatch.setenv("KG_CONV_BS", "1")
        monkeypatch.setenv("KG_CONV_BS_TILE", request.param[2:])
    nv.reload_env()               # the library reads its switches once at load
    yield request.param
    monkeypatch.undo()
    nv.reload_env()


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def layouts(x):
    """the same logical tensor in NCHW and in channel-major storage"""
    cm = x.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)
    return [("nchw", x.contiguous()), ("cntv", cm)]


def close(a, b, tol=TOL, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * ref + 1e-30, f"{what} max err {err:.3e} vs ref max {ref:.3e} (rel {err / max(ref, 1e-30):.2e})"


def cpu_group(g):
    return Group(g.x.cpu(), g.w.cpu(), g.wv, g.Cin, g.taps, g.tap_mode, g.t_stride, g.transposed,
                 None if g.vmap is None else g.vmap.cpu())


CONV_CASES = [
    # N, Cin, M, T, V, taps, stride
    (2, 63, 32, 64, 25, 1, 1), (2, 32, 64, 64, 11, 3, 1), (3, 64, 128, 64, 5, 3, 2), (2, 128, 256, 32, 5, 3, 2),
    (4, 256, 512, 16, 1, 3, 2), (2, 512, 512, 8, 1, 3, 2), (2, 572, 1536, 1, 1, 1, 1), (2, 3, 9, 64, 25, 1, 1),
    (1, 5, 3, 7, 16, 3, 1), (2, 17, 70, 9, 7, 3, 2), (2, 40, 130, 33, 11, 3, 1),
]


@pytest.mark.parametrize("N,Cin,M,T,V,taps,stride", CONV_CASES)
@pytest.mark.parametrize("transposed", [False, True])
def test_conv_time_taps(N, Cin, M, T, V, taps, stride, transposed):
    d = dev()
    if T % stride != 0:
        pytest.skip("stride must divide T")
    w = rnd(M, Cin, taps, 1, seed=1) / (Cin * taps) ** 0.5
    wv = WView(sT=1, sO=Cin * taps, sI=taps)
    if not transposed:
        x = rnd(N, Cin, T, V, seed=2)
        t_out = T // stride
        for name, xl in layouts(x):
            g = Group(xl.to(d), w.to(d), wv, Cin, taps, TAP_TIME, stride, False, None)
            out = nv.conv([g], N, M, t_out, V)
            close(out, pr.conv([cpu_group(g)], N, M, t_out, V))
    else:
        gy = rnd(N, M, T // stride, V, seed=3)
        for name, gl in layouts(gy):
            g = Group(gl.to(d), w.to(d), WView(wv.sT, wv.sI, wv.sO), M, taps, TAP_TIME, stride, True, None)
            out = nv.conv([g], N, Cin, T, V)
            close(out, pr.conv([cpu_group(g)], N, Cin, T, V))


BIG_CONV = [(64, 32, 64, 64, 11, 3, 1), (64, 64, 128, 64, 5, 3, 2), (128, 63, 32, 64, 11, 1, 1), (64, 128, 256, 32, 5, 3, 2)]


@pytest.mark.parametrize("N,Cin,M,T,V,taps,stride", BIG_CONV)
def test_conv_big_tiles(N, Cin, M, T, V, taps, stride):
    """bs=64 shapes: exercises the 128x128 / 64x128 / 32x128 tile configurations the launcher picks for large grids."""
    d = dev()
    w = (rnd(M, Cin, taps, 1, seed=1) / (Cin * taps) ** 0.5).to(d)
    wv = WView(sT=1, sO=Cin * taps, sI=taps)
    x = rnd(N, Cin, T, V, seed=2).to(d)
    gy = rnd(N, M, T // stride, V, seed=3).to(d)
    g = Group(x, w, wv, Cin, taps, TAP_TIME, stride, False, None)
    close(nv.conv([g], N, M, T // stride, V), pr.conv([g], N, M, T // stride, V))     # reference ops on the GPU too
    gt = Group(gy, w, WView(wv.sT, wv.sI, wv.sO), M, taps, TAP_TIME, stride, True, None)
    close(nv.conv([gt], N, Cin, T, V), pr.conv([gt], N, Cin, T, V))
    numel = M * Cin * taps
    ref_w = pr.wgrad(gy, x, Cin, taps, TAP_TIME, stride, None, numel, wv)
    close(nv.wgrad(gy, x, Cin, taps, TAP_TIME, stride, None, numel, wv), ref_w, 5e-5)
    close(nv.wgrad(plane(gy, d), plane(x, d), Cin, taps, TAP_TIME, stride, None, numel, wv), ref_w, 5e-5)


@pytest.mark.parametrize("N,C,T,V,W,K", [(64, 63, 64, 25, 11, 3), (128, 32, 64, 11, 11, 3), (64, 64, 64, 11, 5, 3)])
def test_agg_big(N, C, T, V, W, K):
    d = dev()
    A, x = rnd(K, V, W, seed=1).to(d), rnd(N, C, T, V, seed=2).to(d)
    y = rnd(N, K * C, T, W, seed=3).to(d)
    close(nv.agg_expand(x, A, 1), pr.agg_expand(x, A, 1))
    close(nv.agg_outer(x, y, K, 1), pr.agg_outer(x, y, K, 1), 1e-4)
    close(nv.agg_outer(plane(x, d), plane(y, d), K, 1), pr.agg_outer(x, y, K, 1), 1e-4)      # MFMA kernel (channel-major)
    close(nv.agg_expand(plane(x, d), A, 1), pr.agg_expand(x, A, 1))                            # stream / matrix-core kernels
    y2 = rnd(N, K * C, T, V, seed=4).to(d)
    close(nv.agg_reduce(y2, A, 1), pr.agg_reduce(y2, A, 1))
    close(nv.agg_reduce(plane(y2, d), A, 1), pr.agg_reduce(y2, A, 1))


def plane(t, d):
    """copy into a plane tensor allocated by the library (channel-major, with the lead-in the 128-bit path needs)"""
    out = nv.new_plane(*t.shape, d)
    out.copy_(t)
    return out


PLANE_CASES = [
    # N, Cin, M, T, V, taps, mode, transposed
    (64, 32, 64, 64, 11, 3, TAP_TIME, False), (64, 64, 64, 64, 11, 3, TAP_TIME, True),
    (64, 63, 32, 64, 11, 3, TAP_CHANBLOCK, False), (16, 128, 256, 32, 5, 3, TAP_CHANBLOCK, False),
    (3, 17, 70, 9, 25, 3, TAP_TIME, False), (5, 20, 33, 7, 5, 1, TAP_TIME, False), (70, 40, 96, 64, 1, 3, TAP_TIME, True),
    (2, 3, 9, 64, 25, 1, TAP_TIME, False), (33, 512, 512, 8, 1, 3, TAP_TIME, False),
]


@pytest.mark.parametrize("N,Cin,M,T,V,taps,mode,transposed", PLANE_CASES)
def test_conv_forced_tiles_on_plane_tensors(N, Cin, M, T, V, taps, mode, transposed, monkeypatch):
    """library-allocated plane tensors (lead-in in front of every channel row) under the automatic plan and under every
    forced tile / K-split: same numbers as the torch definition (ragged column counts, row/channel tails, padding
    frames, split-K)."""
    d = dev()
    xc = Cin * (taps if mode == TAP_CHANBLOCK else 1)
    if mode == TAP_CHANBLOCK:
        w = (rnd(taps * M, Cin, 1, 1, seed=1) / (taps * Cin) ** 0.5).to(d)
        wv = WView(M * Cin, Cin, 1)
    else:
        w = (rnd(M, Cin, taps, 1, seed=1) / (taps * Cin) ** 0.5).to(d)
        wv = WView(1, Cin * taps, taps)
    if not transposed:
        x = plane(rnd(N, xc, T, V, seed=2).to(d), d)
        g = Group(x, w, wv, Cin, taps, mode, 1, False, None)
        mo = M
    else:
        x = plane(rnd(N, M, T, V, seed=2).to(d), d)
        g = Group(x, w, WView(wv.sT, wv.sI, wv.sO), M, taps, TAP_TIME, 1, True, None)
        mo = Cin
    bias = rnd(mo, seed=5).to(d)
    addt = plane(rnd(N, mo, T, V, seed=6).to(d), d)
    nv.last_conv_plan = []
    try:
        for plan in ("", "0,1", "1,1", "3,1", "4,2", "2,3", "9,1"):
            if plan:
                monkeypatch.setenv("KG_CONV_PLAN", plan)
            out = nv.conv([g], N, mo, T, V, bias0=bias, add=addt, act=nv.ACT_LRELU)
            if plan:
                assert nv.last_conv_plan[0] == int(plan.split(",")[0]), (plan, nv.last_conv_plan)
            close(out, pr.conv([g], N, mo, T, V, bias0=bias, add=addt, act=nv.ACT_LRELU))
    finally:
        nv.last_conv_plan = None


def test_packed_weights_cached_across_launches():
    """kg_conv_pack once, then launches of several batch sizes on the packed weights (the bf16-split tile kernel alone: plan
    40..42, no workspace); a launch the form cannot run ignores the buffer; a re-pack after a weight change is picked up"""
    d = dev()
    Cin, M, T, V = 32, 64, 64, 11
    wt, wr = (rnd(M, M, 3, 1, seed=3) / (3 * M) ** 0.5).to(d), (rnd(M, Cin, 1, 1, seed=4) / Cin ** 0.5).to(d)
    kw = dict(bias0=rnd(M, seed=5).to(d), bias1=rnd(M, seed=6).to(d), act=nv.ACT_LRELU)

    def groups(n, stride=1, seed=0):
        z, x = plane(rnd(n, M, T, V, seed=1 + seed).to(d), d), plane(rnd(n, Cin, T, V, seed=2 + seed).to(d), d)
        return [Group(z, wt, WView(1, M * 3, 3), M, 3, TAP_TIME, stride, False, None),
                Group(x, wr, WView(0, Cin, 1), Cin, 1, TAP_TIME, stride, False, None)]
    pack = nv.conv_pack(groups(1), 1, M, T, V)
    assert pack is not None and pack.numel() * 4 == 7 * 12 * 128 * 16         # steps x (3 terms x 4 octets) x rows padded to 128
    nv.last_conv_plan = []
    try:
        for n in (2, 5, 64):
            gs = groups(n, seed=n)
            out = nv.conv(gs, n, M, T, V, wpack=pack, **kw)
            assert nv.last_conv_plan[0] in (40, 41, 42), nv.last_conv_plan
            close(out, pr.conv(gs, n, M, T, V, **kw))
        # a transposed launch cannot take the form: the buffer is ignored, the direct kernel runs
        gy = plane(rnd(2, M, T, V, seed=9).to(d), d)
        gt = [Group(gy, wt, WView(1, 3, M * 3), M, 3, TAP_TIME, 1, True, None)]
        out = nv.conv(gt, 2, M, T, V, wpack=pack)
        assert nv.last_conv_plan[0] < 20, nv.last_conv_plan
        close(out, pr.conv(gt, 2, M, T, V))
        # new weights, same buffer
        wt.mul_(-0.5); wr.add_(0.25)
        assert nv.conv_pack(groups(1), 1, M, T, V, out=pack) is pack
        gs = groups(3, seed=7)
        close(nv.conv(gs, 3, M, T, V, wpack=pack, **kw), pr.conv(gs, 3, M, T, V, **kw))
    finally:
        nv.last_conv_plan = None


def test_conv_two_groups_tail_default_plan(monkeypatch, kernel_path):
    """the D-block-1 tail launch (3 temporal taps + 1x1 residual group + two biases + LeakyReLU) on plane tensors"""
    d = dev()
    N, Cin, M, T, V = 8, 32, 64, 64, 11
    z, x = plane(rnd(N, M, T, V, seed=1).to(d), d), plane(rnd(N, Cin, T, V, seed=2).to(d), d)
    wt, wr = (rnd(M, M, 3, 1, seed=3) / (3 * M) ** 0.5).to(d), (rnd(M, Cin, 1, 1, seed=4) / Cin ** 0.5).to(d)
    gs = [Group(z, wt, WView(1, M * 3, 3), M, 3, TAP_TIME, 1, False, None), Group(x, wr, WView(0, Cin, 1), Cin, 1, TAP_TIME, 1, False, None)]
    kw = dict(bias0=rnd(M, seed=5).to(d), bias1=rnd(M, seed=6).to(d), act=nv.ACT_LRELU)
    nv.last_conv_plan = []
    try:
        out = nv.conv(gs, N, M, T, V, **kw)
        if kernel_path.startswith("ring"):      # the forced ring tile really ran (20 + tile code)
            assert nv.last_conv_plan[0] in (20 + int(kernel_path[4:]), 2, 1, 0), nv.last_conv_plan      # (a window tile may not fit)
        elif kernel_path.startswith("bs"):      # the forced bf16-split tile really ran (40 + tile code)
            assert nv.last_conv_plan[0] == 40 + int(kernel_path[2:]), nv.last_conv_plan
        else:
            assert nv.last_conv_plan[0] in (2, 1, 0), nv.last_conv_plan   # default plan = 32-bit-load kernel
        close(out, pr.conv(gs, N, M, T, V, **kw))
    finally:
        nv.last_conv_plan = None


def test_conv_transposed_is_adjoint():
    """<conv(x), g> == <x, convT(g)> for the strided 3-tap case with a vertex gather."""
    d = dev()
    N, Cin, M, T, V, W = 2, 20, 24, 16, 11, 5
    keep = torch.tensor([2, 4, 6, 8, 10], dtype=torch.int32)
    inv = torch.full((V,), -1, dtype=torch.int32)
    inv[keep.long()] = torch.arange(W, dtype=torch.int32)
    w = rnd(M, Cin, 3, 1, seed=1).to(d)
    wv = WView(1, Cin * 3, 3)
    x, g = rnd(N, Cin, T, V, seed=2).to(d), rnd(N, M, T // 2, W, seed=3).to(d)
    y = nv.conv([Group(x, w, wv, Cin, 3, TAP_TIME, 2, False, keep.to(d))], N, M, T // 2, W)
    xt = nv.conv([Group(g, w, WView(1, 3, Cin * 3), M, 3, TAP_TIME, 2, True, inv.to(d))], N, Cin, T, V)
    a, b = (y * g).sum().item(), (x * xt).sum().item()
    assert abs(a - b) <= 1e-4 * abs(a)
    close(y, pr.conv([cpu_group(Group(x, w, wv, Cin, 3, TAP_TIME, 2, False, keep.to(d)))], N, M, T // 2, W))
    close(xt, pr.conv([cpu_group(Group(g, w, WView(1, 3, Cin * 3), M, 3, TAP_TIME, 2, True, inv.to(d)))], N, Cin, T, V))


@pytest.mark.parametrize("N,Cin,M,T,W", [(2, 63, 32, 64, 11), (2, 64, 128, 64, 5), (3, 256, 512, 16, 1), (1, 7, 33, 5, 3)])
def test_conv_chanblock_and_blocked_transpose(N, Cin, M, T, W):
    d = dev()
    K = 3
    w = (rnd(K * M, Cin, 1, 1, seed=1) / (K * Cin) ** 0.5).to(d)
    xa = rnd(N, K * Cin, T, W, seed=2).to(d)
    wv = WView(sT=M * Cin, sO=Cin, sI=1)
    g = Group(xa, w, wv, Cin, K, TAP_CHANBLOCK, 1, False, None)
    z = nv.conv([g], N, M, T, W)
    close(z, pr.conv([cpu_group(g)], N, M, T, W))
    ref = torch.einsum("kmc,nkctw->nmtw", w.view(K, M, Cin).cpu(), xa.view(N, K, Cin, T, W).cpu())
    close(z, ref)
    gz = rnd(N, M, T, W, seed=3).to(d)
    gt = Group(gz, w, WView(0, wv.sI, wv.sO, wv.sT, Cin), M, 1)
    gxa = nv.conv([gt], N, K * Cin, T, W)
    ref = torch.einsum("kmc,nmtw->nkctw", w.view(K, M, Cin).cpu(), gz.cpu()).reshape(N, K * Cin, T, W)
    close(gxa, ref)


@pytest.mark.parametrize("res", ["conv", "identity", "none"])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv_fused_disc_tail(res, stride):
    """two K-slice groups + both biases + identity add + LeakyReLU (the D-block tail launch)."""
    d = dev()
    N, Cin, M, T, V = 3, 24 if res != "identity" else 40, 40, 16, 11
    keep = torch.tensor([0, 2, 5, 7, 9], dtype=torch.int32)
    W = 5 if res == "conv" else V
    z = rnd(N, M, T, W, seed=1).to(d)
    x = rnd(N, Cin, T, V, seed=2).to(d)
    wt = (rnd(M, M, 3, 1, seed=3) / (3 * M) ** 0.5).to(d)
    bt = rnd(M, seed=4).to(d)
    groups = [Group(z, wt, WView(1, M * 3, 3), M, 3, TAP_TIME, stride, False, None)]
    kw = dict(bias0=bt, act=nv.ACT_LRELU, slope=0.2)
    if res == "conv":
        wr = (rnd(M, Cin, 1, 1, seed=5) / Cin ** 0.5).to(d)
        groups.append(Group(x, wr, WView(0, Cin, 1), Cin, 1, TAP_TIME, stride, False, keep.to(d)))
        kw["bias1"] = rnd(M, seed=6).to(d)
    elif res == "identity":
        kw.update(add=x, add_tstride=stride)
    out = nv.conv(groups, N, M, T // stride, W, **kw)
    kc = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items()}
    close(out, pr.conv([cpu_group(g) for g in groups], N, M, T // stride, W, **kc))
    assert (out < 0).any() and (out > 0).any()


WG_CASES = [(2, 63, 32, 64, 11, 3, TAP_CHANBLOCK, 1), (2, 32, 64, 64, 11, 3, TAP_TIME, 1), (2, 64, 128, 64, 5, 3, TAP_TIME, 2),
            (4, 512, 512, 8, 1, 3, TAP_TIME, 2), (2, 572, 1536, 1, 1, 1, TAP_TIME, 1), (2, 3, 9, 64, 25, 1, TAP_TIME, 1),
            (3, 70, 65, 10, 7, 3, TAP_TIME, 2), (2, 256, 512, 16, 5, 1, TAP_TIME, 2), (2, 40, 70, 64, 25, 3, TAP_TIME, 1),
            (5, 33, 20, 9, 16, 3, TAP_TIME, 1), (3, 16, 16, 12, 2, 3, TAP_CHANBLOCK, 1), (1, 1, 1, 1, 1, 3, TAP_TIME, 1)]


@pytest.mark.parametrize("N,Cin,M,T,V,taps,mode,stride", WG_CASES)
def test_wgrad(N, Cin, M, T, V, taps, mode, stride):
    d = dev()
    xc = Cin * (taps if mode == TAP_CHANBLOCK else 1)
    x = rnd(N, xc, T, V, seed=1)
    g = rnd(N, M, T // stride, V, seed=2)
    if mode == TAP_CHANBLOCK:
        wv, numel = WView(M * Cin, Cin, 1), taps * M * Cin
    else:
        wv, numel = WView(1, Cin * taps, taps), M * Cin * taps
    ref = pr.wgrad(g, x, Cin, taps, mode, stride, None, numel, wv)
    for (_, xl), (_, gl) in zip(layouts(x), layouts(g)):
        out = nv.wgrad(gl.to(d), xl.to(d), Cin, taps, mode, stride, None, numel, wv)
        close(out, ref, 5e-5)


@pytest.mark.parametrize("Ns,Cin,M,T,V,taps,mode,stride", [((4, 2, 2), 32, 64, 64, 11, 3, TAP_TIME, 1), ((3, 5), 70, 65, 10, 7, 3, TAP_TIME, 2),
                                                            ((2, 1, 1), 16, 16, 12, 2, 3, TAP_CHANBLOCK, 1), ((128, 64, 64), 64, 64, 64, 11, 3, TAP_TIME, 1)])
def test_wgrad_operand_pairs(Ns, Cin, M, T, V, taps, mode, stride):
    """several (g, x) pairs of one layer (different batch sizes) summed into one gradient by one launch, written or
    accumulated - the three contributions a discriminator weight receives in a WGAN-GP backward pass"""
    d = dev()
    xc = Cin * (taps if mode == TAP_CHANBLOCK else 1)
    if mode == TAP_CHANBLOCK:
        wv, numel = WView(M * Cin, Cin, 1), taps * M * Cin
    else:
        wv, numel = WView(1, Cin * taps, taps), M * Cin * taps
    pairs = [(rnd(n, M, T // stride, V, seed=10 + i), rnd(n, xc, T, V, seed=20 + i)) for i, n in enumerate(Ns)]
    big = Ns[0] >= 64
    ref = sum(pr.wgrad(g.to(d) if big else g, x.to(d) if big else x, Cin, taps, mode, stride, None, numel, wv) for g, x in pairs)
    dp = [(plane(g, d), plane(x, d)) for g, x in pairs]
    out = nv.wgrad(dp[0][0], dp[0][1], Cin, taps, mode, stride, None, numel, wv, extra=dp[1:])
    close(out, ref, 5e-5)
    base = rnd(numel, seed=99).to(d)
    acc = base.clone()
    nv.wgrad(dp[0][0], dp[0][1], Cin, taps, mode, stride, None, numel, wv, out=acc, accumulate=True, extra=dp[1:])
    close(acc, base.cpu().double() + ref.cpu().double(), 5e-5)


def test_wgrad_deferred_reductions():
    """slab reductions of several kg_wgrad launches finished by one kg_wgrad_reduce_many launch (write and accumulate)"""
    d = dev()
    cases = [(2, 32, 64, 64, 11, 3, TAP_TIME, 1), (3, 70, 65, 10, 7, 3, TAP_TIME, 2), (2, 16, 16, 12, 2, 3, TAP_CHANBLOCK, 1),
             (4, 512, 512, 8, 1, 3, TAP_TIME, 2)]
    jobs, outs, refs = [], [], []
    for i, (N, Cin, M, T, V, taps, mode, stride) in enumerate(cases):
        xc = Cin * (taps if mode == TAP_CHANBLOCK else 1)
        wv, numel = (WView(M * Cin, Cin, 1), taps * M * Cin) if mode == TAP_CHANBLOCK else (WView(1, Cin * taps, taps), M * Cin * taps)
        x, g = rnd(N, xc, T, V, seed=30 + i), rnd(N, M, T // stride, V, seed=40 + i)
        base = rnd(numel, seed=50 + i)
        out = base.clone().to(d)
        acc = bool(i % 2)
        nv.wgrad(plane(g, d), plane(x, d), Cin, taps, mode, stride, None, numel, wv, out=out, accumulate=acc, defer=jobs)
        outs.append(out)
        r = pr.wgrad(g, x, Cin, taps, mode, stride, None, numel, wv).double()
        refs.append(base.double() + r if acc else r)
    assert len(jobs) == len(cases)
    nv.wgrad_reduce_many(jobs)
    assert not jobs
    for out, ref in zip(outs, refs):
        close(out, ref, 5e-5)


def test_wgrad_with_vertex_gather():
    d = dev()
    N, Cin, M, T, V, W = 2, 30, 20, 8, 11, 5
    keep = torch.tensor([2, 4, 6, 8, 10], dtype=torch.int32)
    x, g = rnd(N, Cin, T, V, seed=1), rnd(N, M, T // 2, W, seed=2)
    wv = WView(0, Cin, 1)
    out = nv.wgrad(g.to(d), x.to(d), Cin, 1, TAP_TIME, 2, keep.to(d), M * Cin, wv)
    close(out, pr.wgrad(g, x, Cin, 1, TAP_TIME, 2, keep, M * Cin, wv), 5e-5)


AGG_CASES = [(2, 63, 64, 25, 11, 3, 1), (2, 32, 64, 11, 11, 3, 1), (3, 64, 64, 11, 5, 3, 1), (2, 256, 16, 5, 1, 3, 1),
             (2, 512, 8, 1, 1, 3, 1), (2, 256, 4, 1, 5, 1, 1), (2, 64, 8, 5, 11, 1, 2), (2, 3, 32, 11, 25, 1, 2),
             (2, 512, 1, 1, 1, 1, 4), (1, 5, 3, 16, 7, 3, 1), (2, 40, 6, 7, 16, 1, 3), (3, 7, 5, 3, 25, 3, 1),
             (2, 9, 70, 25, 25, 3, 1), (5, 33, 13, 17, 2, 1, 1), (1, 1, 1, 1, 1, 3, 1)]


@pytest.mark.parametrize("N,C,T,V,W,K,rep", AGG_CASES)
def test_agg_family(N, C, T, V, W, K, rep, monkeypatch):
    d = dev()
    A = rnd(K, V, W, seed=1)
    x = rnd(N, C, T, V, seed=2)
    y = rnd(N, K * C, T * rep, W, seed=3)
    for (_, xl), (_, yl) in zip(layouts(x), layouts(y)):
        close(nv.agg_expand(xl.to(d), A.to(d), rep), pr.agg_expand(x, A, rep))
        close(nv.agg_outer(xl.to(d), yl.to(d), K, rep), pr.agg_outer(x, y, K, rep), 5e-5)
    # channel-major layouts take the MFMA kernel when rep == 1; the element-wise kernel must agree on them too
    monkeypatch.setenv("KG_AGG_OUTER_MFMA", "0")
    (_, xl), (_, yl) = layouts(x)[1], layouts(y)[1]
    close(nv.agg_outer(xl.to(d), yl.to(d), K, rep), pr.agg_outer(x, y, K, rep), 5e-5)
    monkeypatch.delenv("KG_AGG_OUTER_MFMA")
    # expand / reduce on channel-major layouts: frame-per-thread kernels ("0") and stream kernels ("1") forced, then the
    # matrix-core kernels (K = 3, rep = 1 launches)
    y2r = rnd(N, K * C, T * rep, V, seed=4)
    y2c = layouts(y2r)[1][1]
    monkeypatch.setenv("KG_AGG_MFMA", "0")
    for mode in ("0", "1", "mfma"):
        if mode == "mfma":
            monkeypatch.delenv("KG_AGG_STREAM")
            monkeypatch.setenv("KG_AGG_MFMA", "1")
            mode = "0"
        monkeypatch.setenv("KG_AGG_STREAM", mode)
        close(nv.agg_expand(xl.to(d), A.to(d), rep), pr.agg_expand(x, A, rep))
        close(nv.agg_reduce(y2c.to(d), A.to(d), rep), pr.agg_reduce(y2r, A, rep))
        At = A.transpose(1, 2).contiguous().to(d)
        close(nv.agg_expand(xl.to(d), At.transpose(1, 2), rep), pr.agg_expand(x, A, rep))
        close(nv.agg_reduce(y2c.to(d), At.transpose(1, 2), rep), pr.agg_reduce(y2r, A, rep))
    monkeypatch.delenv("KG_AGG_STREAM")
    monkeypatch.delenv("KG_AGG_MFMA")
    # reduce: y2 has V on its vertex axis
    y2 = rnd(N, K * C, T * rep, V, seed=4)
    for _, yl in layouts(y2):
        close(nv.agg_reduce(yl.to(d), A.to(d), rep), pr.agg_reduce(y2, A, rep))
    # an adjacency handed over as the transposed VIEW of a (K, W, V) tensor is read in place (adjoint passes)
    At = A.transpose(1, 2).contiguous().to(d)
    assert not At.transpose(1, 2).is_contiguous() or V == 1 or W == 1
    close(nv.agg_expand(x.to(d), At.transpose(1, 2), rep), pr.agg_expand(x, A, rep))
    close(nv.agg_reduce(y2.to(d), At.transpose(1, 2), rep), pr.agg_reduce(y2, A, rep))


@pytest.mark.parametrize("N,C,T,V", [(2, 32, 64, 11), (3, 256, 4, 1), (2, 3, 64, 25), (1, 7, 5, 3), (64, 32, 64, 11)])
def test_rowsum_and_pointwise(N, C, T, V):
    d = dev()
    x, y = rnd(N, C, T, V, seed=1) + 3.0, rnd(N, C, T, V, seed=2)
    noise = rnd(N, 1, T, V, seed=3)
    shift = rnd(C, seed=4) + 3.0
    vec = [rnd(C, seed=10 + i) for i in range(5)]
    for (_, xl), (_, yl) in zip(layouts(x), layouts(y)):
        xd, yd = xl.to(d), yl.to(d)
        close(nv.rowsum(xd), pr.rowsum(x), 1e-5)
        close(nv.rowsum(xd, None, True, shift.to(d)), pr.rowsum(x.double(), None, True, shift.double()), 1e-5)
        close(nv.rowsum(xd, yd, True, shift.to(d)), pr.rowsum(x.double(), y.double(), True, shift.double()), 2e-5)
        close(nv.rowsum(yd, noise.to(d), True)[1], pr.rowsum(y.double(), noise.double(), True)[1], 2e-5)
        for act in (nv.ACT_LRELU, nv.ACT_TANH, nv.ACT_NONE):
            ref_out = torch.tanh(y) if act == nv.ACT_TANH else y
            close(nv.act_bwd(xd, ref_out.to(d), act), pr.act_bwd(x, ref_out, act))
            close(nv.affine_act(xd, *[v.to(d) for v in vec[:2]], yd, vec[2].to(d), vec[3].to(d), noise.to(d),
                                vec[4].to(d), act),
                  pr.affine_act(x, vec[0], vec[1], y, vec[2], vec[3], noise, vec[4], act))
        close(nv.affine_act(xd, vec[0].to(d)), pr.affine_act(x, vec[0]))


@pytest.mark.parametrize("N,C,T,V", [(2, 32, 64, 11), (3, 256, 4, 1), (2, 3, 64, 25), (1, 7, 5, 3), (64, 32, 64, 11), (64, 512, 4, 1)])
def test_rowsum_destinations(N, C, T, V):
    """accumulate / second destination (the gradient sink of two biases that share one gradient), both the
    one-launch (small input) and the two-launch form"""
    d = dev()
    x = rnd(N, C, T, V, seed=1)
    ref = pr.rowsum(x.double())[0]
    o1, o2 = rnd(C, seed=2), rnd(C, seed=3)
    a, b = o1.to(d), o2.to(d)
    nv.rowsum(x.to(d), out=a, accumulate=True, out2=b)
    close(a, o1.double() + ref, 1e-5)
    close(b, o2.double() + ref, 1e-5)
    nv.rowsum(x.to(d), out=a, accumulate=False, out2=b)
    close(a, ref, 1e-5)
    close(b, ref, 1e-5)


@pytest.mark.parametrize("N,C,T,V", [(2, 32, 64, 11), (64, 512, 4, 1), (2, 3, 64, 25), (1, 7, 5, 3), (64, 32, 64, 25), (3, 1, 1, 1)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_coefficients(N, C, T, V, training):
    """kg_bn_fwd / kg_bn_bwd against torch.nn.BatchNorm2d itself (normalised output through kg_affine_act, running
    statistics, input / gamma / beta gradients) and against their torch definitions."""
    d = dev()
    x = rnd(N, C, T, V, seed=1) * 2.0 + 5.0          # a large mean: the one-pass E[x^2]-mean^2 form would lose digits
    g = rnd(N, C, T, V, seed=2)
    gamma, beta = rnd(C, seed=3) + 1.5, rnd(C, seed=4)
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
        bn.running_mean.copy_(rnd(C, seed=5)); bn.running_var.copy_(rnd(C, seed=6).abs() + 0.5)
    rm0, rv0 = bn.running_mean.clone().float(), bn.running_var.clone().float()
    bn.train(training)
    xr = x.double().requires_grad_(True)
    yr = bn(xr)
    yr.backward(g.double())
    for (_, xl), (_, gl) in zip(layouts(x), layouts(g)):
        rm, rv, nbt = rm0.clone().to(d), rv0.clone().to(d), torch.zeros((), dtype=torch.int64, device=d)
        coef = nv.bn_fwd(xl.to(d), gamma.to(d), beta.to(d), rm, rv, nbt, training, 0.1, 1e-5)
        y = nv.affine_act(xl.to(d), coef[0], coef[1])
        close(y, yr, 2e-5)
        close(rm, bn.running_mean, 1e-5)
        close(rv, bn.running_var, 1e-5)
        assert int(nbt.item()) == (1 if training else 0)
        k = nv.bn_bwd(gl.to(d), xl.to(d), gamma.to(d), coef[2].contiguous(), coef[3].contiguous(), training)
        dx = nv.affine_act(gl.to(d), k[0], k[2], xl.to(d), k[1]) if training else nv.affine_act(gl.to(d), k[0])
        close(dx, xr.grad, 5e-5)
        close(k[3], bn.weight.grad, 5e-5)
        close(k[4], bn.bias.grad, 2e-5)
    # torch definitions used by the CPU host-logic tests
    rm, rv, nbt = rm0.clone().double(), rv0.clone().double(), torch.zeros((), dtype=torch.int64)
    cr = pr.bn_fwd(x.double(), gamma.double(), beta.double(), rm, rv, nbt, training, 0.1, 1e-5)
    close(coef, cr, 2e-5)
    close(k, pr.bn_bwd(g.double(), x.double(), gamma.double(), cr[2], cr[3], training), 5e-5)


def test_adam_matches_torch():
    d = dev()
    p0, g = rnd(10007, seed=1), rnd(10007, seed=2)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2e-4, betas=(0.5, 0.999))
    p, m, v = p0.clone().to(d), torch.zeros(10007, device=d), torch.zeros(10007, device=d)
    step = torch.zeros(1, dtype=torch.int32, device=d)
    for it in range(3):
        p_ref.grad = g * (it + 1)
        opt.step()
        step += 1
        nv.adam_step(p, (g * (it + 1) * 2).to(d), m, v, 2e-4, 0.5, 0.999, 1e-8, step, 0.5)
    close(p, p_ref, 1e-6)
    # buffers that are not 16-byte aligned take the scalar kernel
    big = [torch.zeros(10008, device=d) for _ in range(3)]
    p2, m2, v2 = (b[1:] for b in big)
    p2.copy_(p0)
    step.zero_()
    for it in range(3):
        step += 1
        gg = torch.zeros(10008, device=d)
        gg[1:] = (g * (it + 1) * 2).to(d)
        nv.adam_step(p2, gg[1:], m2, v2, 2e-4, 0.5, 0.999, 1e-8, step, 0.5)
    close(p2, p_ref, 1e-6)
    # kg_adam_step_fused: the launch also clears the gradient it has consumed - aligned and unaligned buffers
    for off in (0, 1):
        bufs = [torch.zeros(10007 + off, device=d) for _ in range(4)]
        p3, m3, v3, g3 = (b[off:] for b in bufs)
        p3.copy_(p0)
        step.zero_()
        for it in range(3):
            g3.copy_((g * (it + 1) * 2).to(d))
            step += 1
            nv.adam_step(p3, g3, m3, v3, 2e-4, 0.5, 0.999, 1e-8, step, 0.5, zero_grad=True)
            assert float(g3.abs().max()) == 0.0
        close(p3, p_ref, 1e-6)


def test_error_paths():
    d = dev()
    x = torch.zeros(1, 4, 3, 30, device=d)
    with pytest.raises(RuntimeError, match="V="):
        nv.agg_expand(x, torch.zeros(3, 30, 30, device=d))
    with pytest.raises(TypeError):
        nv.rowsum(x.half())


def _nbr_table(A):
    """neighbour table of an adjacency's non-zero pattern (what disc_trunk.BlockGeom builds from the graph tables)"""
    K, V, W = A.shape
    tab = torch.full((K, W, nv.AGGCONV_P), -1, dtype=torch.int32)
    pcount = [0, 0, 0]
    for k in range(K):
        for w in range(W):
            vs = torch.nonzero(A[k, :, w]).flatten().tolist()
            pcount[k] = max(pcount[k], len(vs))
            tab[k, w, :len(vs)] = torch.tensor(vs, dtype=torch.int32)
    return tab, pcount


AGGCONV_CASES = [
    # dataset, level, dw_s, N, Cin, M, T, const-channel offset of the weight view
    ("ntu", 0, True, 3, 3, 32, 64, 60), ("ntu", 1, False, 4, 32, 64, 64, 0), ("ntu", 1, True, 5, 64, 128, 64, 0),
    ("ntu", 2, False, 6, 128, 256, 32, 0), ("h36m", 0, True, 3, 2, 32, 32, 10), ("h36m", 1, False, 4, 32, 64, 32, 0),
    ("ntu", 0, False, 2, 20, 70, 16, 0), ("ntu", 3, False, 9, 40, 33, 8, 0),
]


@pytest.mark.parametrize("ds,lvl,dw_s,N,Cin,M,T,cc", AGGCONV_CASES)
def test_aggconv_fused_gcn(ds, lvl, dw_s, N, Cin, M, T, cc):
    """kg_aggconv (aggregation fused into the gcn contraction) on the real adjacency patterns, with live non-unit
    edge importances, against expand + conv of the oracle; the label-bias add of block 0, the aggregated planes as a
    side product, NCHW and channel-major inputs, a weight addressed inside its parent (block 0's data channels)."""
    import numpy as np
    from kinetic_gan_amd.graph import build_graph
    d = dev()
    g = build_graph(ds)
    A = torch.tensor(np.asarray(g.As[lvl]), dtype=torch.float32)
    if dw_s:
        A = A[:, :, torch.as_tensor(np.asarray(g.map[lvl + 1][:, 1]))]
    A = (A * (0.5 + torch.rand(A.shape, generator=torch.Generator().manual_seed(5)))).contiguous()
    K, V, W = A.shape
    nbr, pcount = _nbr_table(A)
    x = rnd(N, Cin, T, V, seed=2)
    ctot = Cin + cc
    wfull = rnd(K * M, ctot, 1, 1, seed=1) / (K * Cin) ** 0.5
    wview = wfull.reshape(-1)[cc:]
    wv = WView(sT=M * ctot, sO=ctot, sI=1)
    add = rnd(N, M, 1, W, seed=4)
    for name, xl in layouts(x):
        for use_add in (False, True):
            out, xa = nv.aggconv(xl.to(d), A.to(d), nbr.to(d), pcount, wview.to(d), wv, M,
                                 add=add.to(d) if use_add else None, add_tstride=0, want_xa=True)
            ro, rxa = pr.aggconv(x, A, nbr, pcount, wview, wv, M, add=add if use_add else None, add_tstride=0, want_xa=True)
            close(out, ro)
            close(xa, rxa)
    out, xa = nv.aggconv(x.to(d), A.transpose(1, 2).contiguous().transpose(1, 2).to(d), nbr.to(d), pcount, wview.to(d), wv, M)
    assert xa is None
    close(out, pr.aggconv(x, A, nbr, pcount, wview, wv, M)[0])


@pytest.mark.parametrize("N,Cin,M,T,V,stride", [(3, 64, 32, 64, 11, 1), (2, 512, 256, 16, 5, 2), (4, 128, 64, 32, 5, 2), (2, 40, 70, 9, 7, 1)])
def test_conv_mask_epilogue(N, Cin, M, T, V, stride):
    """kg_conv with the LeakyReLU-derivative mask in its epilogue (g * act'(out) of the consumer folded into the
    producing launch): transposed 1x1 residual conv with an add (the backward's last launch of a block), direct and
    split-K plans, channel-major operands."""
    d = dev()
    t_out = T // stride
    gm = rnd(N, Cin, t_out, V, seed=3)
    w = rnd(Cin, M, 1, 1, seed=1) / Cin ** 0.5            # forward weight (Cin_out, M_in): contraction over dim 0 here
    add = rnd(N, M, T, V, seed=5)
    ref_out = rnd(N, M, T, V, seed=6)                      # the consumer's activation output (sign decides the slope)
    inv = torch.arange(V, dtype=torch.int32)
    for name, gl in layouts(gm):
        g = Group(gl.to(d), w.to(d), WView(0, 1, M), Cin, 1, TAP_TIME, stride, True, inv.to(d))
        cm = layouts(ref_out)[1][1]
        out = nv.conv([g], N, M, T, V, add=layouts(add)[1][1].to(d), mask=cm.to(d), slope=0.2)
        ref = pr.conv([cpu_group(g)], N, M, T, V, add=add, mask=ref_out, slope=0.2)
        close(out, ref)


def test_wgrad_many_layers_one_call():
    """kg_wgrad_many: the weight gradients of several layers (different shapes, tap modes, strides, one to three
    operand pairs, accumulate on / off, a destination addressed inside a parent weight) in shared launches against
    one kg_wgrad definition per layer."""
    d = dev()
    specs = [  # N, Cin, M, T, V, taps, mode, stride, npairs
        (3, 32, 64, 64, 11, 3, TAP_TIME, 1, 2), (2, 64, 128, 32, 5, 3, TAP_TIME, 2, 3), (4, 3, 32, 16, 11, 3, TAP_CHANBLOCK, 1, 1),
        (2, 128, 256, 16, 5, 3, TAP_CHANBLOCK, 1, 2), (5, 256, 512, 8, 1, 1, TAP_TIME, 2, 1), (2, 40, 70, 9, 7, 3, TAP_TIME, 1, 2),
        (2, 512, 512, 4, 1, 3, TAP_TIME, 2, 2), (3, 17, 33, 12, 7, 1, TAP_TIME, 1, 1), (2, 64, 64, 32, 11, 3, TAP_TIME, 1, 3),
        (2, 96, 48, 8, 5, 3, TAP_CHANBLOCK, 1, 1), (2, 20, 24, 8, 5, 1, TAP_TIME, 1, 1), (1, 8, 8, 4, 2, 3, TAP_TIME, 1, 1),
    ]
    jobs, refs = [], []
    for i, (N, Cin, M, T, V, taps, mode, stride, npairs) in enumerate(specs):
        t_out = T // stride
        wnum = M * Cin * taps + (7 if i == 2 else 0)
        wv = WView(sT=1, sO=Cin * taps, sI=taps) if mode == TAP_TIME else WView(sT=M * Cin, sO=Cin, sI=1)
        base = rnd(wnum, seed=100 + i)
        dst = base.clone().to(d)
        out = dst[7:] if i == 2 else dst
        acc = i % 2 == 0
        prs = []
        for q in range(npairs):
            n = N + q
            x = rnd(n, Cin * (taps if mode == TAP_CHANBLOCK else 1), T, V, seed=200 + 10 * i + q)
            g = rnd(n, M, t_out, V, seed=300 + 10 * i + q)
            prs.append((g, x))
        cm = lambda t: t.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)
        jobs.append(dict(g=cm(prs[0][0]).to(d), x=cm(prs[0][1]).to(d), Cin=Cin, taps=taps, tap_mode=mode, t_stride=stride,
                         vmap=None, wv=wv, out=out, accumulate=acc, extra=[(cm(g).to(d), cm(x).to(d)) for g, x in prs[1:]]))
        ref = base.clone()
        ro = ref[7:] if i == 2 else ref
        pr.wgrad(prs[0][0], prs[0][1], Cin, taps, mode, stride, None, ro.numel(), wv, out=ro, accumulate=acc, extra=prs[1:])
        refs.append((dst, ref))
    nv.wgrad_many(jobs)
    for dst, ref in refs:
        close(dst, ref, tol=5e-5)


def test_wgrad_many_more_layers_than_one_launch_holds():
    """kg_wgrad_many takes 20 layers per launch (kernel-argument table): 23 layers go out as two tile launches and two
    slab reductions, every layer still equal to its one-layer definition; the first layers of the call (the coarser
    workgroups of the pass's first half, round 4) and the last ones alike."""
    d = dev()
    jobs, refs = [], []
    for i in range(23):
        N, Cin, M, T, V = 2 + i % 3, (8, 32, 40, 64, 130)[i % 5], (16, 64, 33, 128)[i % 4], (8, 16, 12)[i % 3], (5, 11, 1, 7)[i % 4]
        taps, stride = (3, 1)[i % 2], (1, 2)[(i // 2) % 2]
        if T % stride:
            stride = 1
        wv = WView(sT=1, sO=Cin * taps, sI=taps)
        x, g = rnd(N, Cin, T, V, seed=500 + i), rnd(N, M, T // stride, V, seed=600 + i)
        base = rnd(M * Cin * taps, seed=700 + i)
        dst = base.clone().to(d)
        cm = lambda t: t.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)
        jobs.append(dict(g=cm(g).to(d), x=cm(x).to(d), Cin=Cin, taps=taps, tap_mode=TAP_TIME, t_stride=stride, vmap=None,
                         wv=wv, out=dst, accumulate=True, extra=[]))
        ref = base.clone()
        pr.wgrad(g, x, Cin, taps, TAP_TIME, stride, None, ref.numel(), wv, out=ref, accumulate=True)
        refs.append((dst, ref))
    nv.wgrad_many(jobs)
    for dst, ref in refs:
        close(dst, ref, tol=5e-5)


@pytest.mark.parametrize("split", [2, 3, 4, 8])
@pytest.mark.parametrize("tile", [4, 3, 9])
@pytest.mark.parametrize("fast", ["1", "0"])
def test_ksplit_completion_forms_are_deterministic(monkeypatch, fast, tile, split):
    """a K-split launch sums its partial slabs in a fixed order: bit-identical run after run, in both forms - completed by the
    last workgroup of every tile to arrive (ticket counters, up to 4 splits by default) and by kg_conv_splitk_epilogue - and
    the two forms agree to rounding (bias0 + bias1 are added in a different order).  tile: 32x64 / 64x64 / 32x32 with the
    waves splitting K too"""
    d = dev()
    monkeypatch.setenv("KG_CONV_FAST", fast)        # full-slice instantiation / general instantiation of the tap GEMM
    N, Cin, M, T, V = 4, 512, 500, 8, 3
    x = rnd(N, Cin, T, V, seed=2)
    w = rnd(M, Cin, 3, 1, seed=1) / (3 * Cin) ** 0.5
    bias, bias1 = rnd(M, seed=3).to(d), rnd(M, seed=5).to(d)
    xr = rnd(N, M, T // 2, V, seed=4)
    msk = rnd(N, M, T // 2, V, seed=6)
    g = Group(layouts(x)[1][1].to(d), w.to(d), WView(1, Cin * 3, 3), Cin, 3, TAP_TIME, 2, False, None)
    kw = dict(bias0=bias, bias1=bias1, add=layouts(xr)[1][1].to(d), act=nv.ACT_LRELU, mask=layouts(msk)[1][1].to(d))
    ref = pr.conv([cpu_group(g)], N, M, T // 2, V, bias0=bias.cpu(), bias1=bias1.cpu(), add=xr, act=nv.ACT_LRELU, mask=msk)
    monkeypatch.setenv("KG_CONV_PLAN", "%d,%d" % (tile, split))
    monkeypatch.setenv("KG_CONV_INKERNEL_MAX", "8")
    nv.last_conv_plan = []
    try:
        got = {}
        for form in ("1", "0"):
            monkeypatch.setenv("KG_CONV_INKERNEL", form)
            outs = [nv.conv([g], N, M, T // 2, V, **kw) for _ in range(3)]
            assert nv.last_conv_plan == [tile, split], nv.last_conv_plan       # the launch really is K-split
            for o in outs:
                close(o, ref)
                assert torch.equal(o, outs[0])
            got[form] = outs[0]
        close(got["1"], got["0"], 2e-6)
        assert not (nv._sync_buffer(d) != 0).any()          # every ticket counter is back at zero
    finally:
        nv.last_conv_plan = None


@pytest.mark.parametrize("N,C,T,V", [(64, 3, 64, 25), (5, 2, 32, 16), (3, 7, 9, 5)])
def test_gradient_penalty_kernels(N, C, T, V):
    """kg_gp_fwd / kg_gp_bwd against torch's norm / mean and their autograd gradient (kinetic-gan.py:112-113), on
    NCHW and channel-major gradients, with one all-zero sample (torch.norm's zero-gradient convention)."""
    d = dev()
    g = rnd(N, C, T, V, seed=1) * 0.05
    g[1] = 0
    gout = torch.tensor(10.0)
    gr = g.clone().requires_grad_(True)
    ref = ((gr.reshape(N, -1).norm(2, dim=1) - 1) ** 2).mean()
    (ref * gout).backward()
    for name, gl in layouts(g):
        nrm, gp = nv.gp_fwd(gl.to(d))
        close(gp, ref, tol=1e-5)
        close(nrm, g.reshape(N, -1).norm(2, dim=1), tol=1e-5)
        close(nv.gp_bwd(gl.to(d), nrm, gout.to(d)), gr.grad, tol=1e-5)


def test_agg_outer_deferred_sums_one_launch():
    """kg_agg_outer with the slab sums of several launches deferred to ONE kg_agg_outer_sum_many launch (the six
    blocks of a backward pass), writing into slices of one packed gradient buffer: same results as one by one."""
    d = dev()
    cases = [(3, 32, 3, 25, 11, 16), (4, 64, 3, 11, 11, 8), (2, 128, 3, 11, 5, 8), (5, 256, 3, 5, 5, 4), (3, 512, 3, 5, 1, 4), (6, 512, 3, 1, 1, 2)]
    total = sum(K * V * W for _, _, K, V, W, _ in cases)
    packed = torch.zeros(total, device=d)
    jobs, refs, off = [], [], 0
    for i, (N, C, K, V, W, T) in enumerate(cases):
        x = rnd(N, C, T, V, seed=10 + i)
        y = rnd(N, K * C, T, W, seed=20 + i)
        xl, yl = layouts(x)[1][1].to(d), layouts(y)[1][1].to(d)
        out = packed[off:off + K * V * W].view(K, V, W)
        nv.agg_outer(xl, yl, K, 1, out=out, defer=jobs)
        refs.append((out, pr.agg_outer(x, y, K, 1), nv.agg_outer(xl, yl, K, 1)))
        off += K * V * W
    assert len(jobs) == len(cases)
    nv.agg_outer_finish(jobs)
    for out, ref, single in refs:
        close(out, ref, 5e-5)
        close(out, single, 5e-6)         # (the shared launch deals ONE workgroup budget to its jobs: other slab boundaries)


def test_rowsum_many_one_launch():
    """kg_rowsum_many: the per-channel sums of several tensors (small single-pass ones and long ones that need the
    finishing launch), second destinations, accumulate on / off - against torch sums."""
    d = dev()
    shapes = [(64, 32, 64, 11), (3, 512, 4, 1), (128, 64, 64, 11), (2, 3, 64, 25), (5, 256, 16, 5), (7, 70, 9, 7)]
    jobs, refs = [], []
    for i, (N, C, T, V) in enumerate(shapes):
        x = rnd(N, C, T, V, seed=30 + i)
        base = rnd(C, seed=60 + i)
        out = base.clone().to(d)
        out2 = base.clone().to(d) if i % 2 == 0 else None
        acc = i % 3 != 0
        jobs.append(dict(x=layouts(x)[i % 2][1].to(d), out=out, out2=out2, accumulate=acc))
        ref = x.double().sum((0, 2, 3)) + (base.double() if acc else 0)
        refs.append((out, out2, ref))
    nv.rowsum_many(jobs)
    for out, out2, ref in refs:
        close(out, ref, 1e-5)
        if out2 is not None:
            close(out2, ref, 1e-5)


def test_rowsum_product_row_alone():
    """want_second = 2 (the NoiseInjection.weight gradient, generator.py:16-19): sum over (n,t,v) of x*y with y
    broadcast over channels, alone in a (1, C) destination - single launch and as a job of kg_rowsum_many next to
    plain sums, short (one pass) and long (finishing launch) rows."""
    d = dev()
    jobs, refs = [], []
    for i, (N, C, T, V) in enumerate([(2, 3, 64, 25), (64, 32, 64, 11), (5, 256, 4, 5), (3, 70, 9, 7)]):
        x = rnd(N, C, T, V, seed=80 + i)
        y = rnd(N, 1, T, V, seed=90 + i) if i != 3 else rnd(N, C, T, V, seed=90 + i)
        ref = (x.double() * y.double()).sum((0, 2, 3))
        xd = layouts(x)[i % 2][1].to(d)
        close(nv.rowsum(xd, y.to(d), 2).view(-1), ref, 1e-5)
        base = rnd(C, seed=70 + i)
        out = base.clone().to(d)
        jobs.append(dict(x=xd, y=y.to(d), out=out, accumulate=True))
        refs.append((out, ref + base.double()))
        plain = torch.zeros(C, device=d)
        jobs.append(dict(x=xd, out=plain))
        refs.append((plain, x.double().sum((0, 2, 3))))
    nv.rowsum_many(jobs)
    for out, ref in refs:
        close(out, ref, 1e-5)


@pytest.mark.parametrize("shapes", [[(4, 3, 32, 25), (4, 3, 32, 25)], [(6, 256, 4, 5)], [(64, 32, 16, 11), (64, 32, 16, 11)],
                                    [(2, 3, 64, 25), (2, 70, 9, 7), (8, 5, 300, 3), (4, 1, 1, 1)]])
def test_bn_fwd_many_equals_per_group_launches(shapes):
    """kg_bn_fwd_many (chunked partials + last-arriver merge, several layers and two stacked batches per launch)
    against torch.nn.BatchNorm2d applied to the stacked batches one after the other: coefficients per batch, running
    statistics after both updates, the batch counter; and the ticket counters are left at zero (a second call on
    the same buffers gives the same coefficients)."""
    d = dev()
    jobs, refs = [], []
    for i, (N, C, T, V) in enumerate(shapes):
        x = rnd(N, C, T, V, seed=300 + i) * 2.0 + 3.0
        gamma, beta = rnd(C, seed=310 + i), rnd(C, seed=320 + i)
        bn = torch.nn.BatchNorm2d(C, momentum=0.1, eps=1e-5).double()
        bn.weight.data.copy_(gamma)
        bn.bias.data.copy_(beta)
        bn.train()
        h = N // 2
        per = []
        for q in range(2):
            xq = x[q * h:(q + 1) * h].double()
            bn(xq)
            mean = xq.mean((0, 2, 3))
            var = xq.var((0, 2, 3), unbiased=False)
            rstd = torch.rsqrt(var + 1e-5)
            per.append(torch.stack([gamma.double() * rstd, beta.double() - mean * gamma.double() * rstd, mean, rstd]))
        rm = torch.zeros(C, device=d)
        rv = torch.ones(C, device=d)
        nbt = torch.zeros((), dtype=torch.int64, device=d)
        jobs.append(dict(x=layouts(x)[i % 2][1].to(d), gamma=gamma.to(d), beta=beta.to(d), running_mean=rm, running_var=rv,
                         num_batches_tracked=nbt, momentum=0.1, eps=1e-5, groups=2))
        refs.append((torch.stack(per), bn.running_mean.clone(), bn.running_var.clone(), rm, rv, nbt))
    first = nv.bn_fwd_many(jobs)
    for coef, (want, wrm, wrv, rm, rv, nbt) in zip(first, refs):
        close(coef, want, 2e-5)
        close(rm, wrm, 2e-5)
        close(rv, wrv, 2e-5)
        assert int(nbt.item()) == 2
    again = nv.bn_fwd_many(jobs)
    for a, b in zip(first, again):
        assert torch.equal(a, b)


def test_bn_fwd_cumulative_moving_average_on_device():
    """BatchNorm2d(momentum=None) - torch's cumulative moving average, factor 1 / num_batches_tracked - with the factor
    formed on the device from the live counter (kg_bn_fwd, momentum < 0: no host read of the counter, capturable):
    three successive batches against torch.nn.BatchNorm2d(momentum=None)."""
    d = dev()
    N, C, T, V = 6, 5, 7, 11
    bn = torch.nn.BatchNorm2d(C, momentum=None, eps=1e-5).double().train()
    rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    for i in range(3):
        x = rnd(N, C, T, V, seed=500 + i) * (1.0 + i) + 0.5 * i
        bn(x.double())
        coef = nv.bn_fwd(x.to(d), None, None, rm, rv, nbt, True, -1.0, 1e-5)
        close(coef[2], x.double().mean((0, 2, 3)), 2e-5)
        close(rm, bn.running_mean, 2e-5)
        close(rv, bn.running_var, 2e-5)
        assert int(nbt.item()) == i + 1 == int(bn.num_batches_tracked)


def test_affine_act_per_batch_coefficients():
    """kg_affine_act with groups = 2: the two stacked batches read their own scale / shift vectors (the (groups, 4, C)
    layout of kg_bn_fwd_many) in ONE launch - against two launches, one per batch."""
    d = dev()
    N, C, T, V = 6, 5, 7, 11
    x, r, noise = rnd(N, C, T, V, seed=1), rnd(N, C, T, V, seed=2), rnd(N, 1, T, V, seed=3)
    ct, cr, nw = rnd(2, 4, C, seed=4).to(d), rnd(2, 4, C, seed=5).to(d), rnd(C, seed=6).to(d)
    xd, rd, nd = layouts(x)[1][1].to(d), layouts(r)[0][1].to(d), noise.to(d)
    got = nv.affine_act(xd, ct[0, 0], ct[0, 1], rd, cr[0, 0], cr[0, 1], nd, nw, nv.ACT_LRELU, 0.2, groups=2, coef_gs=4 * C)
    h = N // 2
    for q in range(2):
        want = nv.affine_act(xd[q * h:(q + 1) * h], ct[q, 0], ct[q, 1], rd[q * h:(q + 1) * h], cr[q, 0], cr[q, 1],
                             nd[q * h:(q + 1) * h], nw, nv.ACT_LRELU, 0.2)
        assert torch.equal(got[q * h:(q + 1) * h], want)
    ref = pr.affine_act(x.double(), ct[0, 0].cpu().double(), ct[0, 1].cpu().double(), r.double(), cr[0, 0].cpu().double(),
                        cr[0, 1].cpu().double(), noise.double(), nw.cpu().double(), nv.ACT_LRELU, 0.2)
    close(got[:h], ref[:h], 1e-5)


@pytest.mark.parametrize("N,C,T,V", [(3, 64, 16, 5), (2, 96, 8, 1), (5, 32, 32, 11), (64, 512, 8, 1)])
def test_transposed_stride2_tcn_as_two_parity_launches(N, C, T, V):
    """disc_trunk._tcn_transposed: the input gradient of a stride-2 temporal conv as two launches, one per frame
    parity, each writing every other frame (KgConvArgs.o_tstride) - against the single strided transposed launch
    (which multiplies zeros for half of its (frame, tap) pairs) and the definition."""
    from kinetic_gan_amd import disc_trunk, ops
    d = dev()
    st = ops.ConvSpec(M=C, Cin=C, taps=3, tap_mode=TAP_TIME, t_stride=2, T_in=T, V_in=V, T_out=T // 2, V_out=V,
                      wv=WView(sT=1, sO=C * 3, sI=3), w_shape=(C, C, 3, 1))
    gm = rnd(N, C, T // 2, V, seed=1)
    wt = rnd(C, C, 3, 1, seed=2) * 0.1
    gmd, wtd = layouts(gm)[1][1].to(d), wt.to(d)
    one = nv.conv([Group(gmd, wtd, WView(st.wv.sT, st.wv.sI, st.wv.sO), st.M, 3, TAP_TIME, 2, True, None)], N, C, T, V)
    two = disc_trunk._tcn_transposed(gmd, wtd, st)
    ref = torch.nn.functional.conv_transpose2d(gm.double(), wt.double(), stride=(2, 1), padding=(1, 0), output_padding=(1, 0))
    close(one, ref, 2e-5)
    close(two, ref, 2e-5)
    assert tuple(two.shape) == (N, C, T, V)


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("M,Cin,T,V,N", [(96, 80, 8, 1, 5), (512, 512, 4, 1, 8), (33, 70, 6, 5, 3)])
def test_conv_wave_ksplit_tile(M, Cin, T, V, N, transposed, monkeypatch):
    """K32x32 (plan tile 9: the four waves of a workgroup split K, private weight tiles, partial tiles added in LDS),
    with and without an additional split across workgroups, on launches that use every epilogue feature at once - two
    K-slice groups, biases, residual add, LeakyReLU, the derivative mask, ragged rows / channels / columns - against
    the 32x128 tile and the definition; plus the strided-output form (o_tstride) the frame-parity launches use."""
    d = dev()
    x = rnd(N, Cin, T, V, seed=1)
    x2 = rnd(N, 24, T, V, seed=2)
    addt = plane(rnd(N, M, T, V, seed=3).to(d), d)
    maskt = plane(rnd(N, M, T, V, seed=4).to(d), d)
    b0, b1 = rnd(M, seed=5).to(d), rnd(M, seed=6).to(d)
    if transposed:
        w = rnd(Cin, M, 3, 1, seed=7) * 0.1        # (Cout_fwd = Cin here, Cin_fwd = M): rows of the result = M
        g0 = Group(layouts(x)[1][1].to(d), w.to(d), WView(1, 3, M * 3), Cin, 3, TAP_TIME, 1, True, None)
    else:
        w = rnd(M, Cin, 3, 1, seed=7) * 0.1
        g0 = Group(layouts(x)[1][1].to(d), w.to(d), WView(1, Cin * 3, 3), Cin, 3, TAP_TIME, 1, False, None)
    if transposed:       # both groups store their weights in the same orientation (channel index slowest)
        w2 = rnd(24, M, 1, 1, seed=8) * 0.1
        g1 = Group(layouts(x2)[0][1].to(d), w2.to(d), WView(0, 1, M), 24, 1, TAP_TIME, 1, False, None)
    else:
        w2 = rnd(M, 24, 1, 1, seed=8) * 0.1
        g1 = Group(layouts(x2)[0][1].to(d), w2.to(d), WView(0, 24, 1), 24, 1, TAP_TIME, 1, False, None)
    kw = dict(bias0=b0, bias1=b1, add=addt, act=nv.ACT_LRELU, mask=maskt)
    ref = pr.conv([g0, g1], N, M, T, V, **kw)
    outs = {}
    nv.last_conv_plan = []
    try:
        for plan in ("2,1", "9,1", "9,2"):
            monkeypatch.setenv("KG_CONV_PLAN", plan)
            outs[plan] = nv.conv([g0, g1], N, M, T, V, **kw)
            assert nv.last_conv_plan[0] == int(plan.split(",")[0]), (plan, nv.last_conv_plan)
            close(outs[plan], ref, 2e-5)
        # every other frame of a twice-as-long destination, starting at frame 1
        big = nv.new_plane(N, M, 2 * T, V, d).zero_()
        nv.conv([g0, g1], N, M, T, V, out=big, out_t0=1, out_tstride=2, **kw)
        assert torch.equal(big[:, :, 1::2], outs["9,2"]) and float(big[:, :, 0::2].abs().max()) == 0.0
    finally:
        nv.last_conv_plan = None


def test_bn_bwd_many_equals_single_launches():
    """kg_bn_bwd_many (chunked partial sums + last-arriver, two layers in one launch) against kg_bn_bwd per layer and
    the definition; long 3-channel planes (many chunks per channel) and short 256-channel ones (one chunk)."""
    d = dev()
    jobs, singles, refs = [], [], []
    for i, (N, C, T, V, training) in enumerate([(64, 3, 64, 25, True), (8, 256, 4, 5, True), (5, 32, 32, 11, False)]):
        x, g = rnd(N, C, T, V, seed=400 + i) * 1.5 + 0.5, rnd(N, C, T, V, seed=410 + i)
        gamma = rnd(C, seed=420 + i)
        mean = x.mean((0, 2, 3))
        rstd = torch.rsqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
        xd, gd = layouts(x)[1][1].to(d), layouts(g)[i % 2][1].to(d)
        jobs.append(dict(g=gd, x=xd, gamma=gamma.to(d), mean=mean.to(d), rstd=rstd.to(d), training=training))
        singles.append(nv.bn_bwd(gd, xd, gamma.to(d), mean.to(d), rstd.to(d), training))
        refs.append(pr.bn_bwd(g.double(), x.double(), gamma.double(), mean.double(), rstd.double(), training))
    for rounds in range(2):          # the ticket counters are left at zero: a second call gives the same result
        got = nv.bn_bwd_many(jobs)
        for k, s1, ref in zip(got, singles, refs):
            close(k, ref, 2e-5)
            close(k, s1, 2e-5)


GEN_CASES = [  # ds, lvl (output level), up_s, N, C, Cr, Tc, rep, K
    ("ntu", 0, True, 6, 3, 3, 32, 2, 3),        # G6: 11 -> 25 vertices, 3 channels, identity residual source
    ("ntu", 1, True, 4, 32, 32, 8, 2, 3),       # G4: 5 -> 11
    ("ntu", 1, False, 4, 3, 3, 16, 2, 3),       # G5: no spatial up-sampling
    ("ntu", 2, True, 3, 128, 128, 4, 1, 3),     # G2: 1 -> 5, no frame repeat
    ("ntu", 3, False, 5, 256, 256, 1, 4, 1),    # G1: single vertex, single partition, 1 -> 4 frames
    ("h36m", 0, True, 2, 2, 0, 16, 2, 3),       # no residual branch
    ("h36m", 1, True, 3, 0, 40, 8, 1, 3),       # residual branch alone
]


@pytest.mark.parametrize("ds,lvl,up_s,N,C,Cr,Tc,rep,K", GEN_CASES)
def test_gen_expand_fold_adjfinish(ds, lvl, up_s, N, C, Cr, Tc, rep, K):
    """kg_gen_expand / kg_gen_fold / kg_gen_adj_finish (the generator block on the coarse grid) against their
    definitions: values, the adjoint identity <expand(y), g> == <y, fold(g)>, and d edge_importance against autograd
    through the definition."""
    from kinetic_gan_amd.graph import build_graph
    d = dev()
    g = build_graph(ds)
    V = g.num_node[lvl]
    U = torch.as_tensor(g.upsample_matrix(lvl), dtype=torch.float32) if up_s else None
    Vc = U.shape[0] if up_s else V
    A = (torch.as_tensor(g.As[lvl], dtype=torch.float32)[:K] * (0.5 + torch.rand(K, V, V, generator=torch.Generator().manual_seed(1))))
    y = rnd(N, K * C, Tc, Vc, seed=2) if C else None
    rs = rnd(N, Cr, Tc, Vc, seed=3) if Cr else None
    rb = rnd(Cr, seed=4) if Cr else None
    to = lambda t: None if t is None else t.to(d)
    yl = None if y is None else layouts(y)[1][1].to(d)
    z, r = nv.gen_expand(yl, to(A) if C else None, to(U), rep, C, rs=to(rs), rbias=to(rb))
    zr, rr = pr.gen_expand(y, A if C else None, U, rep, C, rs=rs, rbias=rb)
    if C:
        close(z, zr)
    if Cr:
        close(r, rr)
    gz = rnd(N, C, Tc * rep, V, seed=5) if C else None
    gr = rnd(N, Cr, Tc * rep, V, seed=6) if Cr else None
    gy, grs, zf = nv.gen_fold(to(gz), to(A) if C else None, to(U), rep, K, gr=to(gr), want_zf=bool(C))
    gyr, grsr, zfr = pr.gen_fold(gz, A if C else None, U, rep, K, gr=gr, want_zf=bool(C))
    if C:
        close(gy, gyr)
        close(zf, zfr)
        lhs = (zr.double() * gz.double()).sum()
        rhs = (y.double() * gy.cpu().double()).sum()
        assert abs(lhs - rhs) <= 1e-4 * abs(lhs) + 1e-6
    if Cr:
        close(grs, grsr)
    if C:
        # the same with A * importance and U (A * importance) precomputed for the launch (kg_gen_adj_prepare)
        imp0 = 0.5 + torch.rand(K, V, V, generator=torch.Generator().manual_seed(9))
        A00 = torch.as_tensor(g.As[lvl], dtype=torch.float32)[:K].contiguous()
        ae, bb = torch.empty(K, V, V, device=d), torch.empty(K, Vc, V, device=d)
        nv.gen_adj_prepare([dict(a=to(A00), imp=to(imp0), u=to(U), aeff=ae, b=bb)])
        close(ae, A00 * imp0)
        close(bb, pr._gen_b(A00 * imp0, U)[0])
        z2, _ = nv.gen_expand(yl, None, to(U), rep, C, B=bb)
        close(z2, pr.gen_expand(y, A00 * imp0, U, rep, C)[0])
        gy2, _, _ = nv.gen_fold(to(gz), None, to(U), rep, K, B=bb)
        close(gy2, pr.gen_fold(gz, A00 * imp0, U, rep, K)[0])
        # d edge_importance: autograd through the definition vs outer product (kg_agg_outer) + kg_gen_adj_finish
        A0 = torch.as_tensor(g.As[lvl], dtype=torch.float32)[:K]
        imp = (0.5 + torch.rand(K, V, V, generator=torch.Generator().manual_seed(7))).requires_grad_(True)
        zz, _ = pr.gen_expand(y, A0 * imp, U, rep, C)
        (zz * gz).sum().backward()
        dbt = nv.agg_outer(zf, yl, K, 1)                      # (K, V, Vc)
        out = torch.full((K, V, V), 0.25, device=d)
        nv.gen_adj_finish([dict(dbt=dbt.contiguous(), u=to(U), a=to(A0), out=out, accumulate=True)])
        close(out - 0.25, imp.grad, 1e-4)


@pytest.mark.parametrize("M", [2, 3, 5, 33])
def test_conv_few_rows_with_residual_and_mask(M):
    """Tiles with <= 4 valid rows (the generator's image channels, M = 2 / 3): the upper half-wave of a 32-row MFMA tile
    has NO valid row, and its clamped residual / mask loads must stay inside the (M-channel) operands - an earlier
    clamp read up to four channel rows past their end (a memory fault whenever that memory was unmapped).  The
    operands are allocated exactly (no slack behind them) and the values compared with the definition."""
    d = dev()
    N, Cin, T, V = 64, 6, 16, 7
    x = layouts(rnd(N, Cin, T, V, seed=1))[1][1].to(d)
    w = rnd(Cin, M, seed=2).to(d)                      # transposed orientation: W(m, c) = w[c, m]
    add = rnd(N, M, T, V, seed=3).permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3).to(d)
    mask = rnd(N, M, T, V, seed=4).permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3).to(d)
    grp = Group(x, w, WView(0, 1, M), Cin, 1)
    out = nv.conv([grp], N, M, T, V, add=add, mask=mask, slope=0.2)
    ref = pr.conv([cpu_group(grp)], N, M, T, V, add=add.cpu(), mask=mask.cpu(), slope=0.2)
    close(out, ref)


def test_conv_tiny_channel_kernel_features():
    """The tiny-channel streaming kernel (M <= 16, <= 48 contraction terms: the generator's image-channel convs) takes
    every kg_conv feature: two K-slice groups, temporal taps with stride, vertex gather, transposed taps, channel-block
    taps, row-blocked weights, biases, residual add with frame stride, tanh / LeakyReLU, mask, output frame stride."""
    d = dev()
    N, T, V, W = 5, 12, 7, 4
    keep = torch.tensor([0, 2, 3, 6], dtype=torch.int32)
    z = layouts(rnd(N, 6, T, W, seed=1))[1][1]
    x = layouts(rnd(N, 5, T, V, seed=2))[1][1]
    wt, wr = rnd(3, 6, 3, 1, seed=3), rnd(3, 5, seed=4)
    b0, b1 = rnd(3, seed=5), rnd(3, seed=6)
    add = layouts(rnd(N, 3, T, W, seed=7))[1][1]
    g0 = Group(z.to(d), wt.to(d), WView(1, 18, 3), 6, 3, TAP_TIME, 2, False, None)
    g1 = Group(x.to(d), wr.to(d), WView(0, 5, 1), 5, 1, TAP_TIME, 2, False, keep.to(d))
    plan = []
    nv.last_conv_plan = plan
    try:
        out = nv.conv([g0, g1], N, 3, T // 2, W, bias0=b0.to(d), bias1=b1.to(d), add=add.to(d), add_tstride=2, act=nv.ACT_TANH)
    finally:
        nv.last_conv_plan = None
    ref = pr.conv([cpu_group(g0), cpu_group(g1)], N, 3, T // 2, W, bias0=b0, bias1=b1, add=add, add_tstride=2, act=nv.ACT_TANH)
    close(out, ref)
    # transposed stride-2 taps with a mask epilogue, written to every other frame of a larger tensor
    gm = layouts(rnd(N, 3, T // 2, W, seed=8))[1][1]
    mask = layouts(rnd(N, 6, T, W, seed=9))[1][1]
    gt = Group(gm.to(d), wt.to(d), WView(1, 3, 18), 3, 3, TAP_TIME, 2, True, None)
    out = nv.conv([gt], N, 6, T, W, mask=mask.to(d), slope=0.2)
    close(out, pr.conv([cpu_group(gt)], N, 6, T, W, mask=mask, slope=0.2))
    # channel-block taps with a second row block of weights (w_MB / w_sMB)
    xa = layouts(rnd(N, 9, T, V, seed=10))[1][1]
    wg = rnd(3 * 4 * 3 + 50, seed=11)
    gc = Group(xa.to(d), wg.to(d), WView(12, 3, 1, 40, 2), 3, 3, TAP_CHANBLOCK)
    big = nv.new_plane(N, 4, 2 * T, V, d, zero=True)
    nv.conv([gc], N, 4, T, V, out=big, out_t0=1, out_tstride=2, act=nv.ACT_LRELU)
    refb = torch.zeros(N, 4, 2 * T, V)
    pr.conv([cpu_group(gc)], N, 4, T, V, out=refb, out_t0=1, out_tstride=2, act=nv.ACT_LRELU)
    close(big, refb)


@pytest.mark.parametrize("N,C,T,V", [(192, 512, 4, 1), (5, 70, 3, 2), (64, 512, 2, 1)])
def test_head_kernels(N, C, T, V):
    """kg_head_fwd / kg_head_bwd / kg_head_wgrad: average pool + Linear(latent, 1), the top gradient with the LeakyReLU
    derivative of the last block, and the Linear's gradients (also with accumulation into existing values)."""
    d = dev()
    h = layouts(rnd(N, C, T, V, seed=1))[1][1]
    w, b, gv = rnd(1, C, seed=2), rnd(1, seed=3), rnd(N, 1, seed=4)
    close(nv.head_fwd(h.to(d), w.to(d), b.to(d)), pr.head_fwd(h, w, b))
    for masked in (True, False):
        close(nv.head_bwd(gv.to(d), w.to(d), h.to(d), masked=masked), pr.head_bwd(gv, w, h, masked=masked))
    dw, db = torch.full((C,), 0.5, device=d), torch.full((1,), 0.25, device=d)
    nv.head_wgrad(h.to(d), gv.to(d), dw, db, accumulate=True)
    rw, rb = torch.full((C,), 0.5), torch.full((1,), 0.25)
    pr.head_wgrad(h, gv, rw, rb, accumulate=True)
    close(dw, rw)
    close(db, rb)
    nv.head_wgrad(h.to(d), gv.to(d), dw, None, accumulate=False)
    pr.head_wgrad(h, gv, rw, None, accumulate=False)
    close(dw, rw)


@pytest.mark.parametrize("ds,N,L,T", [("ntu", 128, 60, 64), ("h36m", 9, 10, 32), ("ntu", 7, 120, 4)])
def test_label_bias_kernels(ds, N, L, T):
    """kg_label_bias_fwd / _bwd against the definition: the bias of block 0's label channels and its gradients w.r.t. the
    embedding, the label columns of the gcn weight (written in place, other columns untouched) and the adjacency."""
    from kinetic_gan_amd.graph import build_graph
    d = dev()
    g = build_graph(ds)
    keep = torch.as_tensor(g.keep(0))
    K, C, Cd = 3, 32, 3
    J, cin = L, L + Cd
    ak = (torch.as_tensor(g.As[0], dtype=torch.float32) * (0.5 + torch.rand(3, g.num_node[0], g.num_node[0], generator=torch.Generator().manual_seed(1))))[:, :, keep].contiguous()
    W = ak.shape[2]
    labels = torch.randint(0, L, (N,), generator=torch.Generator().manual_seed(2))
    emb, wg = rnd(L, J, seed=3), rnd(K * C, cin, 1, 1, seed=4) * 0.1
    to = lambda t: t.to(d)
    zl = nv.label_bias_fwd(to(labels), to(emb), to(wg), K, C, cin, J, to(ak))
    close(zl, pr.label_bias_fwd(labels, emb, wg, K, C, cin, J, ak))
    gz = layouts(rnd(N, C, T, W, seed=5))[1][1]
    demb, dw, dak = torch.full((L, J), 0.5, device=d), torch.full((K * C * cin,), 0.25, device=d), torch.full(tuple(ak.shape), 2.0, device=d)
    nv.label_bias_bwd(to(gz), to(labels), to(emb), to(wg), K, C, cin, J, to(ak), demb=demb, dw=dw, dak=dak)
    rdemb, rdw, rdak = torch.full((L, J), 0.5), torch.full((K * C * cin,), 0.25), torch.full(tuple(ak.shape), 2.0)
    pr.label_bias_bwd(gz, labels, emb, wg, K, C, cin, J, ak, demb=rdemb, dw=rdw, dak=rdak)
    close(demb, rdemb, 1e-4)
    close(dw, rdw, 1e-4)
    close(dak, rdak, 1e-4)
    assert torch.equal(dw.view(K * C, cin)[:, J:].cpu(), torch.full((K * C, Cd), 0.25))      # data columns untouched
    # against autograd through the definition
    e2, w2, a2 = emb.clone().requires_grad_(True), wg.clone().requires_grad_(True), ak.clone().requires_grad_(True)
    (pr.label_bias_fwd(labels, e2, w2, K, C, cin, J, a2) * gz.sum(2, keepdim=True)).sum().backward()
    close(demb - 0.5, e2.grad, 1e-4)
    close(dak - 2.0, a2.grad, 1e-4)
    close((dw - 0.25).view(K * C, cin)[:, :J], w2.grad.view(K * C, cin)[:, :J], 1e-4)
    # a label outside [0, L) must not read past the table: its sample comes back as NaN, the others are untouched
    # (nn.Embedding raises there, discriminator.py:57; round-3 ADVICE)
    bad = labels.clone()
    bad[0], bad[N - 1] = L, -1
    zb = nv.label_bias_fwd(to(bad), to(emb), to(wg), K, C, cin, J, to(ak)).cpu()
    assert torch.isnan(zb[0]).all() and torch.isnan(zb[N - 1]).all()
    assert torch.equal(zb[1:N - 1], zl.cpu()[1:N - 1])


@pytest.mark.parametrize("N,D,L,J", [(64, 512, 60, 60), (64, 512, 120, 120), (7, 512, 10, 10), (33, 37, 6, 5), (128, 96, 8, 0)])
def test_mapping_network_kernels(N, D, L, J):
    """kg_linear_fwd / kg_linear_bwd / kg_embed_bwd against their definitions (oracle/prim_ref.py) at the mapping
    network's sizes (generator.py:22-37,80-85: 572 = 512 + 60 columns for NTU-60 - 4-float vector loads; 632 for NTU-120;
    522 for H36M - 2-float loads; an odd size - scalar loads; J = 0 - a layer behind the first), forward, the one-launch
    backward with every output combination, accumulation into existing gradients, the embedding gradient."""
    d = dev()
    Din = D + J
    gen = torch.Generator().manual_seed(N + D)
    x = torch.randn(N, D, generator=gen)
    w = torch.randn(Din, Din, generator=gen) * 0.2
    b = torch.randn(Din, generator=gen)
    emb = torch.randn(L, J, generator=gen) if J else None
    labels = torch.randint(0, L, (N,), generator=gen) if J else None
    to = lambda t: None if t is None else t.to(d)
    y = nv.linear_fwd(to(x), to(w), to(b), nv.ACT_LRELU, 0.2, emb=to(emb), labels=to(labels))
    yr = pr.linear_fwd(x, w, b, 1, 0.2, emb=emb, labels=labels)
    close(y, yr)
    close(nv.linear_fwd(to(x), to(w), None, nv.ACT_NONE, 0.2, emb=to(emb), labels=to(labels)), pr.linear_fwd(x, w, None, 0, 0.2, emb=emb, labels=labels))
    g = torch.randn(N, Din, generator=gen)
    for cols, acc in ((None, False), (J, True), (0, True)):
        dw, db = torch.full((Din, Din), 0.5, device=d), torch.full((Din,), 0.25, device=d)
        rdw, rdb = torch.full((Din, Din), 0.5), torch.full((Din,), 0.25)
        gx = nv.linear_bwd(to(g), to(yr), to(x), to(w), nv.ACT_LRELU, 0.2, emb=to(emb), labels=to(labels), gx_cols=cols, dw=dw, db=db, accumulate=acc)
        rgx = pr.linear_bwd(g, yr, x, w, 1, 0.2, emb=emb, labels=labels, gx_cols=cols, dw=rdw, db=rdb, accumulate=acc)
        if rgx is None:
            assert gx is None
        else:
            close(gx, rgx, 1e-4)
        close(dw, rdw, 1e-4)
        close(db, rdb, 1e-4)
    # input gradient alone (no parameter gradients), through a strided view of g
    g2 = torch.randn(N, 2 * Din, generator=gen)
    gx = nv.linear_bwd(to(g2)[:, :Din] if Din % 4 == 0 else to(g2[:, :Din].contiguous()), to(yr), to(x), to(w), nv.ACT_LRELU, 0.2, emb=to(emb), labels=to(labels))
    close(gx, pr.linear_bwd(g2[:, :Din], yr, x, w, 1, 0.2, emb=emb, labels=labels), 1e-4)
    if J:
        gxe = torch.randn(N, Din, generator=gen)
        demb, rdemb = torch.full((L, J), 2.0, device=d), torch.full((L, J), 2.0)
        nv.embed_bwd(to(gxe), to(labels), demb, accumulate=True)
        pr.embed_bwd(gxe, labels, rdemb, accumulate=True)
        close(demb, rdemb, 1e-4)
        nv.embed_bwd(to(gxe), to(labels), demb, accumulate=False)
        pr.embed_bwd(gxe, labels, rdemb, accumulate=False)
        close(demb, rdemb, 1e-4)
        # a label outside [0, L): that sample is NaN, the others are untouched (nn.Embedding raises, generator.py:80)
        bad = labels.clone()
        bad[N // 2] = L
        yb = nv.linear_fwd(to(x), to(w), to(b), nv.ACT_LRELU, 0.2, emb=to(emb), labels=to(bad)).cpu()
        assert torch.isnan(yb[N // 2]).all()
        keep = torch.arange(N) != N // 2
        assert torch.equal(yb[keep], y.cpu()[keep])


def test_mix3_and_masked_adjacency_kernels():
    d = dev()
    n, C, T, V = 5, 3, 8, 25
    real, fake, alpha = rnd(n, C, T, V, seed=1), layouts(rnd(n, C, T, V, seed=2))[1][1], torch.rand(n, 1, 1, 1)
    out = nv.mix3(real.to(d), fake.to(d), alpha.to(d))
    assert out.is_contiguous()
    close(out, pr.mix3(real, fake, alpha))
    A = rnd(300, seed=3)
    imp = rnd(300, seed=4)
    sel = torch.randperm(300, generator=torch.Generator().manual_seed(5))[:170].contiguous()
    for s_ in (sel, None):
        sd = None if s_ is None else s_.to(d)
        close(nv.masked_adj_fwd(A.to(d), imp.to(d), sd), pr.masked_adj_fwd(A, imp, s_))
        g = rnd(170 if s_ is not None else 300, seed=6)
        for acc in (True, False):
            dimp, ref = torch.full((300,), 0.5, device=d), torch.full((300,), 0.5)
            nv.masked_adj_bwd(g.to(d), A.to(d), sd, dimp, acc)
            pr.masked_adj_bwd(g, A, s_, ref, acc)
            close(dimp, ref)


@pytest.mark.parametrize("form", ["auto", "frame", "stream", "mfma"])
@pytest.mark.parametrize("N,C,T,V,W,s,drop,K", [(5, 64, 32, 5, 11, 2, True, 3), (3, 128, 16, 5, 5, 2, False, 3), (4, 512, 8, 1, 1, 2, False, 1),
                                                (2, 7, 9, 3, 5, 1, True, 3), (40, 32, 64, 11, 25, 1, True, 3), (9, 16, 33, 4, 9, 3, False, 3)])
def test_agg_reduce_residual_and_mask_epilogue(N, C, T, V, W, s, drop, K, form, monkeypatch):
    """kg_agg_reduce with its epilogue - the residual branch's input gradient of a down-sampling block added at the
    frames / vertices it exists on, times the block input's LeakyReLU derivative - in every kernel form (frame per
    thread, stream, matrix cores; ragged last tiles, single-partition level, frame strides 1 / 2 / 3), with either
    operand alone, against the definition."""
    d = dev()
    if form != "auto":
        monkeypatch.setenv("KG_AGG_STREAM", "1" if form == "stream" else "0")
        monkeypatch.setenv("KG_AGG_MFMA", "1" if form == "mfma" else "0")
    keep = torch.arange(0, W, 2) if drop else None
    inv, Vr = None, W
    if drop:
        inv = torch.full((W,), -1, dtype=torch.int32)
        inv[keep] = torch.arange(len(keep), dtype=torch.int32)
        Vr = len(keep)
    Tr = (T + s - 1) // s
    y = layouts(rnd(N, K * C, T, V, seed=1))[1][1]
    A = rnd(K, W, V, seed=4).transpose(1, 2)                 # the adjoint pass hands A^T as a view
    res = layouts(rnd(N, C, Tr, Vr, seed=2))[1][1]
    mask = layouts(rnd(N, C, T, W, seed=3))[1][1]
    invd = None if inv is None else inv.to(d)
    ref = pr.agg_reduce(y, A, 1, res=res, res_tstride=s, res_inv=inv, mask=mask)
    close(nv.agg_reduce(y.to(d), A.to(d), 1, res=res.to(d), res_tstride=s, res_inv=invd, mask=mask.to(d)), ref)
    close(nv.agg_reduce(y.to(d), A.to(d), 1, mask=mask.to(d), slope=0.1), pr.agg_reduce(y, A, 1, mask=mask, slope=0.1))
    close(nv.agg_reduce(y.to(d), A.to(d), 1, res=res.to(d), res_tstride=s, res_inv=invd), pr.agg_reduce(y, A, 1, res=res, res_tstride=s, res_inv=inv))
    close(nv.agg_reduce(y.to(d), A.to(d), 1), pr.agg_reduce(y, A, 1))


@pytest.mark.parametrize("N,C,T,V,bn_t,res,act", [(64, 3, 64, 25, False, "identity", "tanh"), (64, 32, 16, 11, False, "bn", "lrelu"),
                                                   (5, 64, 8, 5, True, "bn", "lrelu"), (7, 512, 1, 1, False, "none", "lrelu"),
                                                   (3, 3, 32, 11, True, "bn", "lrelu")])
def test_gen_tail_backward_kernels(N, C, T, V, bn_t, res, act):
    """kg_gen_tail_stats / kg_gen_tail_apply (g * act'(out), both BatchNorm backward passes, noise-weight and affine
    gradients added into existing buffers) against kg_act_bwd + kg_bn_bwd + the definitions; twice on the same ticket
    counters."""
    d = dev()
    a = nv.ACT_TANH if act == "tanh" else nv.ACT_LRELU
    g, out = layouts(rnd(N, C, T, V, seed=1))[1][1], layouts(torch.tanh(rnd(N, C, T, V, seed=2)))[1][1]
    u, r = layouts(rnd(N, C, T, V, seed=3) * 1.5 + 0.3)[1][1], layouts(rnd(N, C, T, V, seed=4))[1][1]
    noise = rnd(N, 1, T, V, seed=5)

    def stats(x):
        return rnd(C, seed=6) + 1.5, x.mean((0, 2, 3)), torch.rsqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
    st, sr = stats(u), stats(r)
    to = lambda t: None if t is None else t.to(d)
    for rounds in range(2):
        names = ["nw"] + (["gamma_t", "beta_t"] if bn_t else []) + (["gamma_r", "beta_r"] if res == "bn" else [])
        sinks = {k: torch.full((C,), 0.25, device=d) for k in names}
        rsinks = {k: torch.full((C,), 0.25) for k in names}
        du, dr = nv.gen_tail_bwd(to(g), to(out), a, u=to(u) if bn_t else None, bn_t=tuple(map(to, st)) if bn_t else None,
                                 r=to(r) if res != "none" else None, bn_r=tuple(map(to, sr)) if res == "bn" else None,
                                 noise=to(noise), sinks=sinks)
        rdu, rdr = pr.gen_tail_bwd(g, out, a, u=u if bn_t else None, bn_t=st if bn_t else None, r=r if res != "none" else None,
                                   bn_r=sr if res == "bn" else None, noise=noise, sinks=rsinks)
        close(du, rdu, 1e-4)
        if res != "none":
            close(dr, rdr, 1e-4)
        else:
            assert dr is None
        for k in names:
            close(sinks[k], rsinks[k], 1e-4)



@pytest.mark.parametrize("M,stride,N", [(64, 2, 6), (128, 2, 6), (64, 1, 6), (64, 1, 280)])
def test_conv_many_equals_single_launches(M, stride, N, monkeypatch, kernel_path):
    """kg_conv_many: the backward pass's independent contractions on one gradient gm - the transposed temporal conv (two
    frame-parity problems with strided output for stride 2, the second with two K-slice groups) and the residual
    branch's small dense product - in ONE launch; bit-identical to one kg_conv launch per problem, equal to the
    definition, also when the call falls back to single launches (a job with ragged channels / KG_CONV_MANY=0)."""
    d = dev()
    N, T, V, Cr = 6, 32, 11, 32
    gm = plane(rnd(N, M, T // stride, V, seed=1).to(d), d)
    wt = (rnd(M, M, 3, 1, seed=2) / (3 * M) ** 0.5).to(d)
    wr = (rnd(M, Cr, 1, 1, seed=3) / Cr ** 0.5).to(d)

    def jobs(cr=Cr, wres=wr):
        gz = nv.new_plane(N, M, T, V, d).zero_()
        if stride == 2:
            flat, wv = wt.reshape(-1), WView(0, 3, M * 3)
            js = [dict(groups=[Group(gm, flat[1:], wv, M, 1)], N=N, M=M, T_out=T // 2, V_out=V, out=gz, out_t0=0, out_tstride=2),
                  dict(groups=[Group(gm, flat[2:], wv, M, 1), Group(gm[:, :, 1:], flat[0:], wv, M, 1)], N=N, M=M, T_out=T // 2,
                       V_out=V, out=gz, out_t0=1, out_tstride=2)]
        else:
            js = [dict(groups=[Group(gm, wt, WView(1, 3, M * 3), M, 3, TAP_TIME, 1, True, None)], N=N, M=M, T_out=T, V_out=V, out=gz)]
        js.append(dict(groups=[Group(gm, wres, WView(0, 1, cr), M, 1)], N=N, M=cr, T_out=T // stride, V_out=V))
        return js, gz

    js, gz = jobs()
    nv.last_conv_plan = []
    try:
        outs = nv.conv_many(js)
        if not kernel_path.startswith(("ring", "bs")):      # (the shared launch does not use the ring / bf16-split forms: either way is right there)
            assert (nv.last_conv_plan[0] >= 0) == (kernel_path == "default"), nv.last_conv_plan     # really ONE launch
    finally:
        nv.last_conv_plan = None
    js1, gz1 = jobs()
    singles = [nv.conv(**j) for j in js1]
    close(gz, gz1, 5e-6)            # (bit-identical unless a single launch takes the wave-level K-split tile: other summation order)
    close(outs[-1], singles[-1], 5e-6)
    ref_gz = pr.conv([Group(gm.cpu(), wt.cpu(), WView(1, 3, M * 3), M, 3, TAP_TIME, stride, True, None)], N, M, T, V)
    close(gz, ref_gz)
    close(outs[-1], pr.conv([Group(gm.cpu(), wr.cpu(), WView(0, 1, Cr), M, 1)], N, Cr, T // stride, V))
    monkeypatch.setenv("KG_CONV_MANY", "0")
    js2, gz2 = jobs()
    outs2 = nv.conv_many(js2)
    close(gz2, gz, 5e-6)
    close(outs2[-1], outs[-1], 5e-6)
    monkeypatch.delenv("KG_CONV_MANY")
    wr5 = (rnd(M, 40, 1, 1, seed=4) / 40 ** 0.5).to(d)        # 40 rows: fine; contraction depth M: full slices -> still merged
    js3, gz3 = jobs(40, wr5)
    outs3 = nv.conv_many(js3)
    close(gz3, gz, 5e-6)
    close(outs3[-1], pr.conv([Group(gm.cpu(), wr5.cpu(), WView(0, 1, 40), M, 1)], N, 40, T // stride, V))


def test_integration_md_stub_values():
    """The binding INTEGRATION.md section B prints for a reference maintainer (tgcn.py:63-66 through kg_agg_reduce), executed
    as written: its graph_aggregate against the einsum it replaces (round-5 VERDICT: the printed struct had gone stale)."""
    from test_abi_cpu import integration_md_stub
    ns = integration_md_stub(nv.LIB_PATH)
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    for (n, c, t, v, w, k) in [(3, 8, 16, 25, 25, 3), (2, 5, 7, 11, 5, 3), (4, 16, 4, 5, 5, 1)]:
        y = torch.randn(n, k * c, t, v, generator=g).to(d)
        A = torch.rand(k, v, w, generator=g).to(d)
        out = ns["graph_aggregate"](y, A)
        ref = torch.einsum("nkctv,kvw->nctw", y.view(n, k, c, t, v), A)
        assert (out - ref).abs().max().item() <= TOL * ref.abs().max().item()



GENBLOCK_CASES = [  # ds, lvl (output level), up_s, N, Cin, C, Tc, rep, res, bn_t, act
    ("ntu", 2, False, 6, 128, 64, 4, 2, "conv", True, "lrelu"),       # G3
    ("ntu", 1, True, 4, 64, 32, 8, 2, "conv", False, "lrelu"),        # G4: 5 -> 11 vertices
    ("ntu", 1, False, 4, 32, 3, 16, 2, "conv", True, "lrelu"),        # G5: 3 output channels (VALU contractions)
    ("ntu", 0, True, 6, 3, 3, 32, 2, "identity", False, "tanh"),      # G6: self-finishing, identity residual
    ("ntu", 2, True, 4, 256, 128, 4, 1, "conv", False, "lrelu"),      # G2: 4 columns per sample, no frame repeat
    ("h36m", 0, True, 4, 2, 2, 16, 2, "identity", False, "tanh"),
    ("h36m", 1, False, 4, 32, 2, 8, 2, "conv", True, "lrelu"),
    ("ntu", 1, False, 2, 32, 32, 8, 1, "none", True, "lrelu"),        # no residual branch
    ("ntu", 1, False, 2, 48, 48, 4, 3, "identity", True, "lrelu"),    # identity residual behind a BatchNorm tcn, rep 3
]


@pytest.mark.parametrize("ds,lvl,up_s,N,Cin,C,Tc,rep,res,bn_t,act", GENBLOCK_CASES)
def test_genblock_fused_forward_backward(ds, lvl, up_s, N, Cin, C, Tc, rep, res, bn_t, act):
    """kg_genblock_fwd / kg_genblock_bwd (one generator block per launch, generator.py:168-182) against the staged
    definitions composed in oracle/prim_ref.py: every tape tensor, the BatchNorm coefficients and running statistics of
    two stacked batches, the self-finished output; the pending-tail input form; backward outputs, the previous block's
    tail coefficients and the parameter-gradient adds - twice on the same ticket counter."""
    from kinetic_gan_amd.graph import build_graph
    d = dev()
    gr = build_graph(ds)
    V = gr.num_node[lvl]
    U = torch.as_tensor(gr.upsample_matrix(lvl), dtype=torch.float32).contiguous() if up_s else None
    Vc = U.shape[0] if up_s else V
    K = 3
    Kp = 1 if V == 1 else K
    T = Tc * rep
    a_ = nv.ACT_TANH if act == "tanh" else nv.ACT_LRELU
    dims = nv.GenBlockDims(Cin=Cin, C=C, K=K, Kp=Kp, Tc=Tc, Vc=Vc, T=T, V=V, rep=rep, res_kind={"none": 0, "identity": 1, "conv": 2}[res],
                           bn_t=bn_t, act=a_)
    A = torch.as_tensor(gr.As[lvl], dtype=torch.float32)[:K] * (0.5 + torch.rand(K, V, V, generator=torch.Generator().manual_seed(1)))
    B = pr._gen_b(A, U)[0][:Kp].contiguous()
    wg = rnd(K * C, Cin, 1, 1, seed=2) / Cin ** 0.5
    wr = rnd(C, Cin, 1, 1, seed=3) / Cin ** 0.5 if res == "conv" else None
    br = rnd(C, seed=4) if res == "conv" else None
    wt = rnd(C, C, 3, 1, seed=5) / (3 * C) ** 0.5
    bt = rnd(C, seed=6)
    nw = rnd(1, C, 1, 1, seed=7) * 0.3
    noise = rnd(N, 1, T, V, seed=8)
    to = lambda t: None if t is None else t.to(d)
    assert nv.genblock_supported(dims, N, to(wg), to(wr), to(wt)) and nv.genblock_supported(dims, N, to(wg), to(wr), to(wt), backward=True)

    def bn_layer(seed, dv):
        g_ = torch.Generator().manual_seed(seed)
        return dict(gamma=(torch.rand(C, generator=g_) + 0.5).to(dv), beta=torch.randn(C, generator=g_).to(dv),
                    running_mean=torch.randn(C, generator=g_).to(dv), running_var=(torch.rand(C, generator=g_) + 0.5).to(dv),
                    num_batches_tracked=torch.tensor(3, dtype=torch.int64, device=dv), momentum=0.1, eps=1e-5)

    x = rnd(N, Cin, Tc, Vc, seed=9)
    # the same input as the pending tail of a previous block: act(pu * s + b + pr * s' + b' + pnw * pnoise), two batches
    pu, prr = rnd(N, Cin, Tc, Vc, seed=10), rnd(N, Cin, Tc, Vc, seed=11)
    pct, pcr = rnd(2, 4, Cin, seed=12) * 0.5, rnd(2, 4, Cin, seed=13) * 0.5
    pnoise, pnw = rnd(N, 1, Tc, Vc, seed=14), rnd(1, Cin, 1, 1, seed=15) * 0.2
    for mode in ("x", "pend"):
        for rounds in range(2):
            bts, brs = (bn_layer(20, d), bn_layer(20, "cpu")) if bn_t else (None, None), (bn_layer(21, d), bn_layer(21, "cpu")) if res == "conv" else (None, None)
            kw = dict(wg=wg, wr=wr, br=br, wt=wt, bt=bt, B=B, U=U, groups=2, noise=noise, nw=nw)
            kd = {k: (to(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
            if mode == "x":
                got = nv.genblock_fwd(dims, x=plane(x.to(d), d), bn_t=bts[0], bn_r=brs[0], **kd)
                ref = pr.genblock_fwd(dims, x=x, bn_t=bts[1], bn_r=brs[1], **kw)
            else:
                pend = dict(u=pu, r=prr, ct=pct, cr=pcr, noise=pnoise, nw=pnw, act=nv.ACT_LRELU)
                pend_d = {k: (plane(v.to(d), d) if k in ("u", "r") else to(v) if torch.is_tensor(v) else v) for k, v in pend.items()}
                got = nv.genblock_fwd(dims, pend=pend_d, bn_t=bts[0], bn_r=brs[0], **kd)
                ref = pr.genblock_fwd(dims, pend=pend, bn_t=bts[1], bn_r=brs[1], **kw)
            for k in ("x", "yc", "z", "r", "u", "ct", "cr", "out"):
                assert (got[k] is None) == (ref[k] is None), k
                if ref[k] is not None:
                    close(got[k], ref[k], 1e-4 if k in ("ct", "cr") else TOL * 2, f"{mode}:{k}")
            for bl in (bts, brs):
                if bl[0] is not None:
                    close(bl[0]["running_mean"], bl[1]["running_mean"], 1e-5)
                    close(bl[0]["running_var"], bl[1]["running_var"], 1e-5)
                    assert int(bl[0]["num_batches_tracked"]) == int(bl[1]["num_batches_tracked"]) == 5
    # ---- backward
    g = rnd(N, C, T, V, seed=30)
    out = torch.tanh(rnd(N, C, T, V, seed=31))
    u, r = rnd(N, C, T, V, seed=32) * 1.5 + 0.3, rnd(N, C, T, V, seed=33)
    coef = rnd(6, C, seed=34)
    pg = torch.Generator().manual_seed(35)
    prev_x = torch.tanh(rnd(N, Cin, Tc, Vc, seed=36))

    def stats(t, seed):
        return rnd(Cin, seed=seed) + 1.5, t.mean((0, 2, 3)), torch.rsqrt(t.var((0, 2, 3), unbiased=False) + 1e-5)
    for with_prev in (False, True):
        for rounds in range(2):
            names = ["nw", "gamma_t", "beta_t", "gamma_r", "beta_r"]
            sinks = {k: torch.full((Cin,), 0.25, device=d) for k in names}
            rsinks = {k: torch.full((Cin,), 0.25) for k in names}
            prev = prev_d = None
            if with_prev:
                prev = dict(x=prev_x, u=pu, r=prr, noise=pnoise, act=nv.ACT_LRELU, bn_t=stats(pu, 40), bn_r=stats(prr, 41), sinks=rsinks)
                prev_d = dict(x=plane(prev_x.to(d), d), u=plane(pu.to(d), d), r=plane(prr.to(d), d), noise=to(pnoise), act=nv.ACT_LRELU,
                              bn_t=tuple(map(to, prev["bn_t"])), bn_r=tuple(map(to, prev["bn_r"])), sinks=sinks)
            kw = dict(g=g, out=out, u=u if bn_t else None, r=r if res == "conv" else None, coef=coef, wg=wg, wr=wr, wt=wt, B=B, U=U)
            kd = {k: (plane(v.to(d), d) if k in ("g", "out", "u", "r") and v is not None else to(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
            got = nv.genblock_bwd(dims, prev=prev_d, **kd)
            ref = pr.genblock_bwd(dims, prev=prev, **kw)
            for k in ("du", "dr", "gyc", "zf", "gx", "pcoef"):
                assert (got[k] is None) == (ref[k] is None), k
                if ref[k] is not None:
                    close(got[k], ref[k], 1e-4)
            if with_prev:
                for k in names:
                    close(sinks[k], rsinks[k], 1e-4)
