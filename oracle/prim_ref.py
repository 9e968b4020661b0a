"""ORACLE (test infrastructure): plain-torch definition of every libkgan_hip.so entry point.

Same signatures as kinetic_gan_amd._native, arithmetic in float64-free fp32 torch ops written
straight from the formulas in include/kgan_hip.h.  Two uses, both in tests/ only:
  * `-m gpu` tests compare each HIP kernel against these on the same seeded inputs;
  * CPU tests install them in place of the native functions (``install``) to exercise the
    autograd composition (ops.py, modules, WGAN-GP step incl. double backward) without a GPU.
The product never imports this file.
"""
from __future__ import annotations

import torch

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
TAP_TIME, TAP_CHANBLOCK = 0, 1


def _weights(w, wv, taps, M, Cin):
    flat = w.contiguous().reshape(-1)
    d = torch.arange(taps, device=w.device).view(-1, 1, 1)
    m = torch.arange(M, device=w.device).view(1, -1, 1)
    c = torch.arange(Cin, device=w.device).view(1, 1, -1)
    mb = min(wv.MB, 1 << 30)
    idx = d * wv.sT + (m // mb) * wv.sMB + (m % mb) * wv.sO + c * wv.sI
    need = int(idx.max()) + 1
    if need > flat.numel():
        # a row block that lives in ANOTHER parameter of the same flat buffer (w_sMB = distance between the two
        # parameters, kgan_hip.h): address the underlying storage like the kernel's raw pointer does
        flat = torch.as_strided(w.detach(), (need,), (1,), w.storage_offset())
    return flat[idx]                       # (taps, M, Cin)


def _gather_src(x, Cin, d, taps, tap_mode, t_stride, transposed, vmap, T_out, V_out):
    """X_src (N, Cin, T_out, V_out) for tap d, zero where the source is outside the frame range."""
    n, _, T_in, V_in = x.shape
    dev = x.device
    shift = d - (taps - 1) // 2 if tap_mode == TAP_TIME else 0
    choff = d * Cin if tap_mode == TAP_CHANBLOCK else 0
    to = torch.arange(T_out, device=dev)
    if not transposed:
        ti = to * t_stride + shift
        ok_t = (ti >= 0) & (ti < T_in)
    else:
        num = to - shift
        ok_t = (num >= 0) & (num % t_stride == 0)
        ti = torch.div(num, t_stride, rounding_mode="floor")
        ok_t = ok_t & (ti >= 0) & (ti < T_in)
    ti = ti.clamp(0, T_in - 1)
    if vmap is None:
        vi = torch.arange(V_out, device=dev)
        ok_v = torch.ones(V_out, dtype=torch.bool, device=dev)
    else:
        vi = vmap.long()
        ok_v = vi >= 0
        vi = vi.clamp(0, V_in - 1)
    xs = x[:, choff:choff + Cin][:, :, ti][:, :, :, vi]
    mask = (ok_t.view(-1, 1) & ok_v.view(1, -1)).to(x.dtype)
    return xs * mask


def conv(groups, N, M, T_out, V_out, bias0=None, bias1=None, add=None, add_tstride=1,
         act=ACT_NONE, slope=0.2, mask=None, out=None, out_t0=0, out_tstride=1, wpack=None):
    # (wpack: kg_conv's packed weights - a layout of the same weights, nothing to restate)
    dest = out
    out = torch.zeros(N, M, T_out, V_out, dtype=torch.float32, device=groups[0].x.device)
    for g in groups:
        W = _weights(g.w, g.wv, g.taps, M, g.Cin)
        for d in range(g.taps):
            xs = _gather_src(g.x, g.Cin, d, g.taps, g.tap_mode, g.t_stride, g.transposed, g.vmap, T_out, V_out)
            out = out + torch.einsum("mc,nctv->nmtv", W[d], xs)
    if bias0 is not None:
        out = out + bias0.view(1, -1, 1, 1)
    if bias1 is not None:
        out = out + bias1.view(1, -1, 1, 1)
    if add is not None:
        # add_tstride == 0: one frame broadcast over the output frames
        out = out + (add[:, :, :1] if add_tstride == 0 else add[:, :, ::add_tstride][:, :, :T_out])
    out = _act(out, act, slope)
    if mask is not None:
        out = out * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))
    if dest is not None:        # output frame `to` goes to frame out_t0 + to * out_tstride of the caller's tensor
        dest[:, :, out_t0::out_tstride][:, :, :T_out] = out
        return dest
    return out


def _act(v, act, slope):
    if act == ACT_LRELU:
        return torch.where(v > 0, v, v * slope)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def wgrad_reduce_many(jobs):
    """the emulated wgrad finishes immediately: nothing is ever deferred"""
    assert not jobs


def wgrad_many(jobs):
    """kg_wgrad_many: every job is an independent kg_wgrad"""
    dws = [j["out"].data_ptr() for j in jobs]
    assert len(set(dws)) == len(dws), "two jobs write the same dw"
    for j in jobs:
        wgrad(j["g"], j["x"], j["Cin"], j["taps"], j["tap_mode"], j["t_stride"], j.get("vmap"), j["out"].numel(),
              j["wv"], out=j["out"], accumulate=j.get("accumulate", False), extra=j.get("extra", ()))


def wgrad(g, x, Cin, taps, tap_mode, t_stride, vmap, w_numel, wv, out=None, accumulate=False, extra=(), defer=None):
    if extra:       # further operand pairs of the same layer: their products are summed into the same gradient
        total = wgrad(g, x, Cin, taps, tap_mode, t_stride, vmap, w_numel, wv)
        for ge, xe in extra:
            total = total + wgrad(ge, xe, Cin, taps, tap_mode, t_stride, vmap, w_numel, wv)
        if out is not None:
            if accumulate:
                out += total        # positions the weight view does not address stay untouched: they are 0 in total
            else:
                out.copy_(total)
            return out
        return total
    n, M, T_out, V_out = g.shape
    dw = torch.zeros(w_numel, dtype=torch.float32, device=g.device)
    d_ = torch.arange(taps, device=g.device).view(-1, 1, 1)
    m_ = torch.arange(M, device=g.device).view(1, -1, 1)
    c_ = torch.arange(Cin, device=g.device).view(1, 1, -1)
    idx = d_ * wv.sT + m_ * wv.sO + c_ * wv.sI
    vals = torch.stack([torch.einsum("nmtv,nctv->mc", g,
                                     _gather_src(x, Cin, d, taps, tap_mode, t_stride, False, vmap, T_out, V_out))
                        for d in range(taps)])
    dw[idx.reshape(-1)] = vals.reshape(-1)
    if out is not None:
        sel = idx.reshape(-1)
        if accumulate:
            out[sel] += dw[sel]
        else:
            out[sel] = dw[sel]
        return out
    return dw


def agg_expand(x, A, rep=1):
    n, c, t, v = x.shape
    k = A.shape[0]
    xr = x.repeat_interleave(rep, dim=2) if rep > 1 else x
    out = torch.einsum("nctv,kvw->nkctw", xr, A)
    return out.reshape(n, k * c, t * rep, A.shape[2])


def aggconv_supported(V, W, pcount, ncols):
    span = (127 // W + 2) * V
    return span <= 384 and pcount[0] <= 1 and pcount[1] <= 4 and pcount[2] <= 1 and ncols >= 8192


def aggconv(x, A, nbr, pcount, w, wv, M, add=None, add_tstride=1, want_xa=False):
    """kg_aggconv: sum_k W_k (x A_k) (+ add).  The neighbour table must list every non-zero of A (it is how the
    kernel finds them) and pcount must bound its rows."""
    k, v, wd = A.shape
    nz = (A != 0)
    tab = torch.zeros_like(nz)
    for kk in range(k):
        for ww in range(wd):
            ent = [int(e) for e in nbr[kk, ww].tolist() if e >= 0]
            assert len(ent) <= pcount[kk], (kk, ww, ent, pcount)
            for e in ent:
                tab[kk, e, ww] = True
    assert not bool((nz & ~tab).any()), "neighbour table misses non-zeros of the adjacency"
    xa = agg_expand(x, A, 1)
    cin = x.shape[1]
    flat = w.contiguous().reshape(-1)
    d = torch.arange(k, device=w.device).view(-1, 1, 1)
    m = torch.arange(M, device=w.device).view(1, -1, 1)
    c = torch.arange(cin, device=w.device).view(1, 1, -1)
    Wk = flat[d * wv.sT + m * wv.sO + c * wv.sI]            # (K, M, Cin)
    out = torch.einsum("kmc,nkctw->nmtw", Wk, xa.reshape(x.shape[0], k, cin, x.shape[2], wd))
    if add is not None:
        out = out + (add[:, :, :1] if add_tstride == 0 else add[:, :, ::add_tstride][:, :, :out.shape[2]])
    return out, (xa if want_xa else None)


def agg_reduce(y, A, fold=1, res=None, res_tstride=1, res_inv=None, mask=None, slope=0.2):
    n, kc, tin, v = y.shape
    k = A.shape[0]
    c = kc // k
    out = torch.einsum("nkctv,kvw->nctw", y.reshape(n, k, c, tin, v), A)
    if fold > 1:
        out = out.reshape(n, c, tin // fold, fold, A.shape[2]).sum(3)
    if res is not None or mask is not None:     # kg_agg_reduce's epilogue = kg_scatter_add_act on the aggregate
        assert fold == 1
        if res is not None:
            out = _scatter_add_act(out, res, res_tstride, res_inv, mask=mask, slope=slope, inplace=False)
        else:
            out = out * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))
    return out


def agg_outer_finish(jobs):
    """the recorded problems are computed now (their destinations hold NaN until then, as a reader that came too
    early would notice)"""
    for j in jobs:
        j()
    jobs.clear()


def agg_outer(x, y, K, rep=1, out=None, defer=None):
    n, c, t, v = x.shape
    if defer is not None:
        if out is None:
            out = torch.empty((K, v, y.shape[3]), dtype=x.dtype, device=x.device)
        out.fill_(float("nan"))
        defer.append(lambda: agg_outer(x, y, K, rep, out))
        return out
    xr = x.repeat_interleave(rep, dim=2) if rep > 1 else x
    res = torch.einsum("nctv,nkctw->kvw", xr, y.reshape(n, K, c, t * rep, y.shape[3]))
    if out is not None:
        out.copy_(res)
        return out
    return res


def rowsum(x, y=None, second=False, shift=None, out=None, accumulate=False, out2=None):
    s0 = x.sum((0, 2, 3))
    if not second:
        res = s0.view(1, -1)
    else:
        sh = 0 if shift is None else shift.reshape(1, -1, 1, 1)
        s1 = ((x - sh) ** 2 if y is None else x * (y - sh)).sum((0, 2, 3))
        res = torch.stack([s0, s1]) if int(second) == 1 else s1.view(1, -1)      # 2: the product row alone
    if out2 is not None:
        if accumulate:
            out2.view(res.shape).add_(res)
        else:
            out2.view(res.shape).copy_(res)
    if out is not None:
        if accumulate:
            out.view(res.shape).add_(res)
        else:
            out.view(res.shape).copy_(res)
        return out
    return res


def rowsum_many(jobs):
    dsts = [j["out"].data_ptr() for j in jobs] + [j["out2"].data_ptr() for j in jobs if j.get("out2") is not None]
    assert len(set(dsts)) == len(dsts), "two jobs write the same destination"
    for j in jobs:
        rowsum(j["x"], j.get("y"), 2 if j.get("y") is not None else False, out=j["out"],
               accumulate=j.get("accumulate", False), out2=j.get("out2"))


def bn_fwd(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps):
    """kg_bn_fwd: (4, C) = [scale, shift, mean, rstd]; torch.nn.BatchNorm2d's running-statistics update."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    if training:
        mean = x.mean((0, 2, 3))
        var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean((0, 2, 3))
        if momentum < 0:      # torch's momentum=None: cumulative moving average
            momentum = 1.0 / float(int(num_batches_tracked) + 1 if num_batches_tracked is not None else 1)
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1 - momentum).add_(var * (n / max(n - 1, 1)), alpha=momentum)
        if num_batches_tracked is not None:
            num_batches_tracked.add_(1)
    else:
        mean, var = running_mean, running_var
    rstd = torch.rsqrt(var + eps)
    scale = rstd if gamma is None else gamma * rstd
    shift = -mean * scale if beta is None else beta - mean * scale
    return torch.stack([scale, shift, mean, rstd])


def bn_fwd_many(jobs):
    """kg_bn_fwd_many: per job the (groups, 4, C) coefficients of `groups` batches stacked along N, running
    statistics updated batch by batch."""
    res = []
    for j in jobs:
        g = int(j.get("groups", 1))
        h = j["x"].shape[0] // g
        res.append(torch.stack([bn_fwd(j["x"][q * h:(q + 1) * h], j.get("gamma"), j.get("beta"), j.get("running_mean"),
                                       j.get("running_var"), j.get("num_batches_tracked"), True, j["momentum"], j["eps"])
                                for q in range(g)]))
    return res


def bn_bwd(g, x, gamma, mean, rstd, training):
    """kg_bn_bwd: (5, C) = [a, b, c, dgamma, dbeta] with dL/dx = a*g + b*x + c."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    s0 = g.sum((0, 2, 3))
    q = (g * (x - mean.view(1, -1, 1, 1))).sum((0, 2, 3)) * rstd
    a = rstd if gamma is None else gamma * rstd
    if training:
        b = -a * rstd * q / n
        c = -a * s0 / n - b * mean
    else:
        b, c = torch.zeros_like(a), torch.zeros_like(a)
    return torch.stack([a, b, c, q, s0])


def bn_bwd_many(jobs):
    return [bn_bwd(j["g"], j["x"], j.get("gamma"), j["mean"], j["rstd"], j["training"]) for j in jobs]


def act_bwd(g, ref, act, slope=0.2):
    if act == ACT_LRELU:
        return g * torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, slope))
    if act == ACT_TANH:
        return g * (1 - ref * ref)
    return g.clone()


def affine_act(x, sx=None, bx=None, r=None, sr=None, br=None, noise=None, nw=None, act=ACT_NONE, slope=0.2, out=None,
               groups=1, coef_gs=0):
    if groups > 1:
        # batch q's coefficient vectors sit q * coef_gs floats behind the first batch's, in the same buffer
        def of(t, q):
            return None if t is None else torch.as_strided(t, (t.numel(),), (1,), t.storage_offset() + q * coef_gs)
        h = x.shape[0] // groups
        res = torch.cat([_affine_act(x[q * h:(q + 1) * h], of(sx, q), of(bx, q), None if r is None else r[q * h:(q + 1) * h],
                                     of(sr, q), of(br, q), None if noise is None else noise[q * h:(q + 1) * h], nw, act,
                                     slope) for q in range(groups)], 0)
    else:
        res = _affine_act(x, sx, bx, r, sr, br, noise, nw, act, slope)
    if out is not None:
        out.copy_(res)
        return out
    return res


def _affine_act(x, sx=None, bx=None, r=None, sr=None, br=None, noise=None, nw=None, act=ACT_NONE, slope=0.2):
    def vec(t):
        return t.reshape(1, -1, 1, 1)
    v = x
    if sx is not None:
        v = v * vec(sx)
    if bx is not None:
        v = v + vec(bx)
    if r is not None:
        v = v + (r * vec(sr) if sr is not None else r)
    if br is not None:
        v = v + vec(br)
    if noise is not None and nw is not None:
        v = v + vec(nw) * noise
    return _act(v, act, slope)


def gp_fwd(g):
    nrm = g.reshape(g.shape[0], -1).norm(2, dim=1)
    return nrm, ((nrm - 1) ** 2).mean()


def gp_bwd(g, nrm, gout):
    coef = torch.where(nrm > 0, (2.0 / g.shape[0]) * (1 - 1 / nrm.clamp_min(1e-38)), torch.zeros_like(nrm)) * gout.reshape(())
    return g * coef.view(-1, 1, 1, 1)


def adam_step(p, g, m, v, lr, b1, b2, eps, step_t, grad_scale=1.0, zero_grad=False):
    """kg_adam_step / kg_adam_step_fused (zero_grad: the launch clears the gradient it has consumed)"""
    t = float(step_t.item())
    gi = g * grad_scale
    if zero_grad:
        g.zero_()
    m.mul_(b1).add_(gi, alpha=1 - b1)
    v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
    denom = v.sqrt() / (1 - b2 ** t) ** 0.5 + eps
    p.addcdiv_(m, denom, value=-(lr / (1 - b1 ** t)))


def _gen_b(A, U):
    """(B (K, Vc, V) = U A_k, U (Vc, V)) with U = identity when absent"""
    if A is None and U is None:
        return None, None
    v = A.shape[1] if A is not None else U.shape[1]
    Um = U if U is not None else torch.eye(v, dtype=torch.float32, device=(A if A is not None else U).device)
    return (torch.einsum("cv,kvw->kcw", Um, A) if A is not None else None), Um


def gen_adj_prepare(jobs):
    """kg_gen_adj_prepare: aeff = a * imp, b = u aeff (u None: b = aeff)"""
    for j in jobs:
        ae = j["a"] * j["imp"] if j.get("imp") is not None else j["a"].clone()
        j["aeff"].copy_(ae)
        j["b"].copy_(ae if j.get("u") is None else torch.einsum("cv,kvw->kcw", j["u"], ae))


def gen_expand(y, A, U, rep, C_out, rs=None, rbias=None, B=None):
    """kg_gen_expand: z = sum_k y_k (U A_k) repeated over `rep` frames, r = rs U + rbias likewise."""
    Bc, Um = _gen_b(A, U)
    B = B if B is not None else Bc
    z = r = None
    if y is not None:
        n, kc, tc, vc = y.shape
        k = kc // C_out
        z = torch.einsum("nkctv,kvw->nctw", y.reshape(n, k, C_out, tc, vc), B).repeat_interleave(rep, dim=2)
    if rs is not None:
        r = rs if Um is None else torch.einsum("nctv,vw->nctw", rs, Um)
        if rbias is not None:
            r = r + rbias.view(1, -1, 1, 1)
        r = r.repeat_interleave(rep, dim=2)
    return z, r


def gen_fold(gz, A, U, rep, K, gr=None, want_zf=False, y_out=None, rs_out=None, B=None):
    """kg_gen_fold: the adjoint of gen_expand (+ gz summed over the repeated frames)."""
    Bc, Um = _gen_b(A, U)
    B = B if B is not None else Bc
    gy = grs = zf = None
    if gz is not None:
        n, c, tf, v = gz.shape
        f = gz.reshape(n, c, tf // rep, rep, v).sum(3)
        gy = torch.einsum("nctw,kvw->nkctv", f, B).reshape(n, K * c, tf // rep, B.shape[1])
        if y_out is not None:
            y_out.copy_(gy)
            gy = y_out
        if want_zf:
            zf = gz if rep == 1 else f
    if gr is not None:
        n, c, tf, v = gr.shape
        grs = gr.reshape(n, c, tf // rep, rep, v).sum(3)
        if Um is not None:
            grs = torch.einsum("nctw,vw->nctv", grs, Um)
        if rs_out is not None:
            rs_out.copy_(grs)
            grs = rs_out
    return gy, grs, zf


def _tail_coefs(gp, u, bn_t, r, bn_r, noise, sinks):
    """(6, C) = [a_t, b_t, c_t, a_r, b_r, c_r] of kg_gen_tail_stats; parameter gradients ADDED into the sinks."""
    c = gp.shape[1]
    one, zero = torch.ones(c, dtype=gp.dtype, device=gp.device), torch.zeros(c, dtype=gp.dtype, device=gp.device)
    rows = []
    for x, stats, kg, kb in ((u, bn_t, "gamma_t", "beta_t"), (r, bn_r, "gamma_r", "beta_r")):
        if stats is None:
            rows += [one, zero, zero]
            continue
        gam, mean, rstd = stats
        k = bn_bwd(gp, x, gam, mean, rstd, True)
        if sinks.get(kg) is not None:
            sinks[kg].add_(k[3])
        if sinks.get(kb) is not None:
            sinks[kb].add_(k[4])
        rows += [k[0], k[1], k[2]]
    if noise is not None and sinks.get("nw") is not None:
        sinks["nw"].add_((gp * noise).sum((0, 2, 3)))
    return torch.stack(rows)


def gen_tail_bwd(g, out, act, u=None, bn_t=None, r=None, bn_r=None, noise=None, sinks=None, slope=0.2, coef=None, stats_only=False):
    """kg_gen_tail_stats + kg_gen_tail_apply (``coef`` given: apply only; ``stats_only``: the (6, C) coefficients)"""
    sinks = sinks or {}
    gp = act_bwd(g, out, act, slope)
    if coef is None:
        coef = _tail_coefs(gp, u, bn_t, r, bn_r, noise, sinks)
    if stats_only:
        return coef

    def ap(x, k0):
        return gp * coef[k0].view(1, -1, 1, 1) + x * coef[k0 + 1].view(1, -1, 1, 1) + coef[k0 + 2].view(1, -1, 1, 1)

    du = ap(u, 0) if bn_t is not None else gp
    dr = None
    if r is not None:
        dr = ap(r, 3) if bn_r is not None else gp
    return du, dr


# ---- fused generator block (kg_genblock_fwd / kg_genblock_bwd): the staged entry points composed -------------------------

def _gb_lds_bytes(d, backward):
    """the eligibility rule of kg_genblock.hip (make_layout): per-sample working set <= 150 KB of LDS, contraction shapes"""
    def path(M, K, N, at):
        if (at or K % 16 == 0) and (M >= 17 or N >= 64):
            return True, True
        return (M <= 32 and K * (4 if M <= 4 else 16 if M <= 16 else 32) <= 2048), False
    Nc, Nf, ZP = d.Tc * d.Vc, d.T * d.V, (d.T + 2) * d.V
    Mg = d.Kp * d.C
    Mh = Mg + (d.C if d.res_kind == 2 else 0)
    ok0, m0 = path(d.Cin, Mh, Nc, True) if backward else path(Mh, d.Cin, Nc, False)
    ok1, _ = path(d.C, 3 * d.C, Nf, backward)
    if not (ok0 and ok1) or (not backward and m0 and d.Cin % 4):
        return -1
    r4 = lambda v: (v + 3) & ~3
    if not backward:
        tot = r4(max(d.Cin * Nc + Mh * Nc, d.C * Nf)) + r4(d.C * ZP) + r4(d.C * Nf if d.res_kind else 0)
    else:
        tot = (r4(d.C * ZP) + r4(d.C * Nf if d.res_kind else 0) + r4(d.C * Nf) + r4(Mh * Nc) + r4(d.C * Nc if d.res_kind == 1 else 0)
               + r4(d.Cin * Nc))
    tot += r4(d.Kp * d.Vc * d.V) + r4(d.Vc * d.V) + r4(d.C) + 2048
    return tot * 4 if tot * 4 <= 150 * 1024 else -1


def genblock_supported(d, n, wg, wr, wt, backward=False):
    return _gb_lds_bytes(d, backward) >= 0


def _gb_head_weight(d, wg, wr):
    Mg = d.Kp * d.C
    w = wg.reshape(-1, d.Cin)[:Mg]
    return torch.cat([w, wr.reshape(d.C, d.Cin)]) if d.res_kind == 2 else w


def genblock_fwd(d, *, x=None, pend=None, wg, wr=None, br=None, wt, bt=None, B, U=None, bn_t=None, bn_r=None, groups=1,
                 noise=None, nw=None, slope=0.2):
    if x is None:
        ct, cr = pend.get("ct"), pend.get("cr")
        x = affine_act(pend["u"], ct[0, 0] if ct is not None else None, ct[0, 1] if ct is not None else None, pend.get("r"),
                       cr[0, 0] if cr is not None else None, cr[0, 1] if cr is not None else None, pend.get("noise"),
                       None if pend.get("nw") is None else pend["nw"].reshape(-1), pend["act"], slope,
                       groups=groups if (ct is not None or cr is not None) else 1, coef_gs=4 * d.Cin)
    Mg = d.Kp * d.C
    yc = torch.einsum("mc,nctv->nmtv", _gb_head_weight(d, wg, wr), x)
    rs = yc[:, Mg:] if d.res_kind == 2 else (x if d.res_kind == 1 else None)
    z, r = gen_expand(yc[:, :Mg], None, U, d.rep, d.C, rs=rs, rbias=br if d.res_kind == 2 else None, B=B)
    u = torch.nn.functional.conv2d(z, wt.reshape(d.C, d.C, 3, 1), bt, padding=(1, 0))
    jobs = []
    if bn_t is not None:
        jobs.append(dict(bn_t, x=u, groups=groups))
    if bn_r is not None:
        jobs.append(dict(bn_r, x=r, groups=groups))
    coefs = bn_fwd_many(jobs) if jobs else []
    ct = coefs[0] if bn_t is not None else None
    cr = coefs[-1] if bn_r is not None else None
    out = None
    if ct is None and cr is None:
        out = affine_act(u, None, None, r, None, None, noise, None if nw is None else nw.reshape(-1), d.act, slope)
    return dict(x=x, yc=yc, z=z, r=r, u=u, ct=ct, cr=cr, out=out)


def genblock_bwd(d, *, g, out, u=None, r=None, coef, wg, wr=None, wt, B, U=None, prev=None, slope=0.2):
    gp = act_bwd(g, out, d.act, slope)

    def ap(x, k0):
        return gp * coef[k0].view(1, -1, 1, 1) + x * coef[k0 + 1].view(1, -1, 1, 1) + coef[k0 + 2].view(1, -1, 1, 1)

    du = ap(u, 0) if d.bn_t else gp
    dr = (ap(r, 3) if d.res_kind == 2 else gp) if d.res_kind != 0 else None
    gz = torch.nn.functional.conv_transpose2d(du, wt.reshape(d.C, d.C, 3, 1), padding=(1, 0))
    gy, grs, zf = gen_fold(gz, None, U, d.rep, d.Kp, gr=dr, want_zf=True, B=B)
    gyc = torch.cat([gy, grs], 1) if d.res_kind == 2 else gy
    gx = torch.einsum("mc,nmtv->nctv", _gb_head_weight(d, wg, wr), gyc)
    if d.res_kind == 1:
        gx = gx + grs
    pcoef = None
    if prev is not None:
        gpp = act_bwd(gx, prev["x"], prev["act"], slope)
        pcoef = _tail_coefs(gpp, prev.get("u"), prev.get("bn_t"), prev.get("r"), prev.get("bn_r"), prev.get("noise"),
                            prev.get("sinks") or {})
    return dict(du=du, dr=dr, gyc=gyc, zf=zf.contiguous() if zf is gz else zf, gx=gx, pcoef=pcoef)


def gen_adj_finish(jobs):
    """kg_gen_adj_finish: out[k,v,w] (+)= a[k,v,w] * sum_vc u[vc,v] dbt[k,w,vc] for k < Kd (0 beyond)."""
    for j in jobs:
        dbt, out = j["dbt"], j["out"]
        kd = dbt.shape[0]
        u = j.get("u")
        d = dbt.transpose(1, 2) if u is None else torch.einsum("cv,kwc->kvw", u, dbt)
        full = torch.zeros_like(out)
        full[:kd] = d
        if j.get("a") is not None:
            full = full * j["a"]
        if j.get("accumulate", False):
            out.add_(full)
        else:
            out.copy_(full)


def head_fwd(h, w, b):
    v = h.mean((2, 3)) @ w.reshape(-1)
    return v + b.reshape(()) if b is not None else v


def head_bwd(gv, w, h, masked=True, slope=0.2):
    n, c, t, v = h.shape
    g = (gv.reshape(-1, 1) * w.reshape(1, -1) / float(t * v)).view(n, c, 1, 1).expand(n, c, t, v)
    if masked:
        g = g * torch.where(h > 0, torch.ones_like(h), torch.full_like(h, slope))
    return g.contiguous()


def head_wgrad(x, gv, dw, db, accumulate=True):
    d = gv.reshape(-1) @ x.mean((2, 3))
    s = gv.sum().reshape(1)
    if accumulate:
        dw.add_(d)
        if db is not None:
            db.add_(s)
    else:
        dw.copy_(d)
        if db is not None:
            db.copy_(s)


def _label_p(emb, wg, K, C_out, cin, J):
    Wc = wg.reshape(K, C_out, cin)[:, :, :J]
    return torch.einsum("kcj,lj->lkc", Wc, emb)                     # (L, K, C)


def label_bias_fwd(labels, emb, wg, K, C_out, cin, J, ak):
    P = _label_p(emb, wg, K, C_out, cin, J)
    S = ak.sum(1)                                                   # (K, W)
    table = torch.einsum("lkc,kw->lcw", P, S)
    return table[labels].unsqueeze(2).contiguous()                  # (N, C, 1, W)


def label_bias_bwd(gz, labels, emb, wg, K, C_out, cin, J, ak, demb=None, dw=None, dak=None, accumulate=True, dak_accumulate=True):
    L = emb.shape[0]
    gzl = gz.sum(2)                                                 # (N, C, W)
    dT = torch.zeros(L, C_out, gz.shape[3], dtype=gz.dtype, device=gz.device).index_add_(0, labels, gzl)
    S = ak.sum(1)
    P = _label_p(emb, wg, K, C_out, cin, J)
    Q = torch.einsum("lcw,kw->lkc", dT, S)
    Wc = wg.reshape(K, C_out, cin)[:, :, :J]
    if demb is not None:
        d = torch.einsum("kcj,lkc->lj", Wc, Q)
        demb.copy_(demb + d if accumulate else d)
    if dw is not None:
        d = torch.einsum("lj,lkc->kcj", emb, Q)
        view = dw.reshape(K, C_out, cin)[:, :, :J]
        view.copy_(view + d if accumulate else d)
    if dak is not None:
        dS = torch.einsum("lcw,lkc->kw", dT, P)
        d = dS.unsqueeze(1).expand_as(dak)
        dak.copy_(dak + d if dak_accumulate else d)


def mix3(real, fake, alpha):
    al = alpha.reshape(-1, 1, 1, 1)
    return torch.cat((real, fake, al * real + (1 - al) * fake), 0).contiguous()


def _scatter_add_act(a, b, t_stride, inv_vmap, mask=None, slope=0.2, shape=None, inplace=True):
    """(a + scatter(b)) * lrelu'(mask): the epilogue of kg_agg_reduce (kgan_hip.h)"""
    n, c, t, v = tuple(a.shape) if a is not None else shape
    out = torch.zeros(n, c, t, v, dtype=b.dtype, device=b.device) if a is None else a.clone()
    tb = min(b.shape[2], (t + t_stride - 1) // t_stride)
    if inv_vmap is None:
        out[:, :, 0:tb * t_stride:t_stride] += b[:, :, :tb]
    else:
        vs = torch.nonzero(inv_vmap >= 0).reshape(-1)
        out[:, :, 0:tb * t_stride:t_stride][:, :, :, vs] += b[:, :, :tb][:, :, :, inv_vmap[vs].long()]
    if mask is not None:
        out = out * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))
    if a is not None and inplace:
        a.copy_(out)
        return a
    return out


def masked_adj_fwd(A_all, imp_all, sel):
    ae = A_all * imp_all if imp_all is not None else A_all.clone()
    return ae if sel is None else ae.index_select(0, sel)


def masked_adj_bwd(g, A_all, sel, dimp, accumulate):
    g = g.reshape(-1)
    if sel is None:
        d = g * A_all
        dimp.copy_(dimp + d if accumulate else d)
    else:
        d = g * A_all[sel]
        if accumulate:
            dimp.index_add_(0, sel, d)
        else:
            dimp.index_copy_(0, sel, d)


def conv_many(jobs):
    """kg_conv_many: the jobs one by one (kgan_hip.h)"""
    return [conv(**j) for j in jobs]


def _lin_in(x, emb, labels):
    return x if emb is None else torch.cat((emb[labels], x), 1)


def _lin_act(v, act, slope):
    return torch.nn.functional.leaky_relu(v, slope) if act == 1 else v


def linear_fwd(x, w, b, act=1, slope=0.2, emb=None, labels=None):
    """kg_linear_fwd: act(cat(emb[labels], x) @ w.T + b)   (generator.py:22-37 one Linear + LeakyReLU; :80-82)"""
    return _lin_act(torch.nn.functional.linear(_lin_in(x, emb, labels), w, b), act, slope)


def linear_bwd(g, y, x, w, act=1, slope=0.2, emb=None, labels=None, gx_cols=None, dw=None, db=None, accumulate=False):
    """kg_linear_bwd: g' = g * act'(y); gx = (g' @ w)[:, :gx_cols]; dw (+)= g'.T @ xin; db (+)= g'.sum(0)"""
    gp = g * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope)) if act == 1 else g
    xin = _lin_in(x, emb, labels)
    cols = w.shape[1] if gx_cols is None else gx_cols
    gx = (gp @ w)[:, :cols].contiguous() if cols > 0 else None
    if dw is not None:
        r = (gp.t() @ xin).reshape(dw.shape)
        dw.copy_(dw + r if accumulate else r)
    if db is not None:
        r = gp.sum(0).reshape(db.shape)
        db.copy_(db + r if accumulate else r)
    return gx


def embed_bwd(gx, labels, demb, accumulate=False):
    """kg_embed_bwd: demb[l] (+)= sum_{n: labels[n] = l} gx[n, :J]"""
    r = torch.zeros_like(demb).index_add_(0, labels, gx[:, :demb.shape[1]])
    demb.copy_(demb + r if accumulate else r)


NAMES = ["genblock_supported", "genblock_fwd", "genblock_bwd", "linear_fwd", "linear_bwd", "embed_bwd", "gen_tail_bwd", "head_fwd", "head_bwd", "head_wgrad", "label_bias_fwd", "label_bias_bwd", "mix3", "masked_adj_fwd", "masked_adj_bwd",
         "gen_expand", "gen_fold", "gen_adj_finish", "gen_adj_prepare", "conv", "conv_many", "wgrad", "wgrad_many", "wgrad_reduce_many", "aggconv", "aggconv_supported", "agg_expand", "agg_reduce", "agg_outer", "agg_outer_finish", "rowsum", "rowsum_many", "bn_fwd", "bn_fwd_many", "bn_bwd", "bn_bwd_many", "act_bwd", "affine_act", "gp_fwd", "gp_bwd",
         "adam_step"]


def install(native_module):
    """Swap the native entry points of kinetic_gan_amd._native for these emulations (tests only).
    Returns a callable that restores the originals."""
    saved = {k: getattr(native_module, k) for k in NAMES}
    g = globals()
    for k in NAMES:
        setattr(native_module, k, g[k])

    def restore():
        for k, f in saved.items():
            setattr(native_module, k, f)
    return restore
