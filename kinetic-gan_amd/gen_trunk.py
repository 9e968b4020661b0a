"""Hand-scheduled forward / backward of the generator's seven st_gcn blocks (generator.py:89-95, 168-182).

Evaluated block by block through ``ops.py`` a generator block is 7-9 launches forward and ~11 backward plus autograd's
accumulation adds, for 5 % of the iteration's flops (pure launch latency).  Here the seven blocks are ONE autograd node
(``GenTrunkFn``, first order only: the generator is differentiated once, kinetic-gan.py:167-173) whose passes are plain
launch sequences over the C ABI, and every block runs CONTRACT-FIRST ON ITS INPUT GRID - a 1x1 conv commutes with
upsample_s (generator.py:185-200) and the nearest frame repeat (generator.py:172):

  FWD per block   yc  = [W_gcn; W_res] x                          ONE kg_conv on (N, *, Tc, Vc): the weight's second
                                                                  row block is the residual conv (KgConvGroup.w_MB)
                  z,r = kg_gen_expand(yc; U A_k, U, rep, b_res)   gcn aggregation + both up-samplings + residual branch
                  u   = W_tcn * z + b                             kg_conv, 3 temporal taps (1 tap at T = 1)
                  coef= kg_bn_fwd_many(u, r)                      BatchNorm statistics of both branches, all stacked batches
                  out = kg_affine_act(u, r, noise)                BN(u) + BN(r) + w_noise * noise, LeakyReLU / tanh
  BWD per block   du, dr = kg_gen_tail_stats / _apply(g, out, u, r)  act', both BatchNorm backward passes, noise-weight
                                                                  and affine gradients: two launches
                  gz  = W_tcn^T * du                              kg_conv transposed
                  gyc = kg_gen_fold(gz, dr)                       adjoint of kg_gen_expand (+ frame-folded gz)
                  gx  = [W_gcn; W_res]^T gyc (+ identity branch)  ONE kg_conv with two K-slice groups
                  parameter gradients: weight / bias / noise-weight products are deferred into the pass's shared
                  kg_wgrad_many / kg_rowsum_many launches (ops parameter sink), the adjacency outer products into one
                  kg_agg_outer_many launch and kg_gen_adj_finish (d A_k = U^T d B_k, d importance = A * d A).

Round 6 (gen_trunk.FUSED, default on; KG_GEN_FUSED=0 keeps the staged form): for the blocks whose per-sample working set fits
LDS (the last four; kg_genblock.hip) the FWD sequence is ONE launch (the block's normalise + noise + activation rides at the front of the next
block's launch: "pending tail") and the BWD sequence is ONE launch that also takes the tail statistics of the block before
it; the tensors the deferred parameter-gradient launches read are written as the staged form writes them.

The up-sampled input, the 3*C_out-plane conv output at the output resolution and the separate residual-conv launch
never exist; the gcn / residual weight gradients contract over the COARSE columns (2-5x fewer).  With two batches
stacked along N (``Generator.synthesis_pair``) every forward launch covers both, BatchNorm statistics are taken per
batch in order, and the backward pass touches the differentiated batch only.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
from torch.autograd import Function

from . import _native as nv
from . import ops
from ._native import ACT_LRELU, ACT_TANH, TAP_TIME, Group, WView

SLOPE = 0.2
# The blocks whose per-sample working set fits LDS run as ONE launch per block and direction (kg_genblock_fwd / kg_genblock_bwd,
# round 6): with the block geometry fixed at compile time (the NTU / Human3.6M generators' last four / two blocks) a launch
# takes 13-30 us against 31-45 us for the staged five-launch sequence of the same block, the iteration 3.17 ms against 3.25 ms
# (profiles/r06_genblock_*.log).  KG_GEN_FUSED=0 / gen_trunk.FUSED = False: every block through the staged sequence (A/B, tests).
FUSED = os.environ.get("KG_GEN_FUSED", "1") != "0"
# A fused launch puts ONE sample on a workgroup: every workgroup streams the block's weights and its contractions have
# Tc * Vc columns.  Blocks with fewer input-grid columns than this (the 512- / 256-channel blocks at T <= 4, V = 1: 1-4
# columns against 0.7-1.8 MB of weights) keep the row-split staged form.
FUSED_MIN_COLS = int(os.environ.get("KG_GEN_FUSED_MINCOLS", "16"))


class GenBlockGeom:
    """Static description of one generator block for one input geometry."""

    def __init__(self, blk, Tc: int, Vc: int, device):
        self.cin, self.cout, self.K = blk.in_channels, blk.out_channels, blk.gcn.kernel_size
        self.res = blk.res_kind
        self.bn_t = len(blk.tcn) > 1
        self.act = ACT_TANH if blk.tan else ACT_LRELU
        self.Tc, self.Vc = Tc, Vc
        self.T = blk.up_t
        self.ok = self.T >= Tc and self.T % Tc == 0 and self.K <= 3
        self.rep = self.T // Tc if self.ok else 1
        if blk.up_s:
            self.U = torch.as_tensor(blk.graph.upsample_matrix(blk.lvl), dtype=torch.float32, device=device).contiguous()
            self.V = self.U.shape[1]
            self.ok = self.ok and self.U.shape[0] == Vc
        else:
            self.U, self.V = None, Vc
        a_lvl = np.asarray(blk.graph.As[blk.lvl])
        self.ok = self.ok and a_lvl.shape[1] == self.V
        # single-vertex level: A = [[1], [0], [0]] - only the first partition's rows of the gcn weight take part
        self.Kp = 1 if bool(getattr(blk.gcn, "single_partition", False)) and self.V == 1 else self.K
        C = self.cout
        self.Mg = self.Kp * C
        self.Mh = self.Mg + (C if self.res == "conv" else 0)
        self.dims = nv.GenBlockDims(Cin=self.cin, C=C, K=self.K, Kp=self.Kp, Tc=Tc, Vc=Vc, T=self.T, V=self.V, rep=self.rep,
                                    res_kind={"none": 0, "identity": 1, "conv": 2}[self.res], bn_t=self.bn_t, act=self.act)
        self.A_fixed = torch.as_tensor(a_lvl, dtype=torch.float32, device=device).contiguous()
        # parameter-gradient geometries (ops.ConvSpec: what kg_wgrad_many needs)
        self.spec_g = ops.ConvSpec(M=self.Mg, Cin=self.cin, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=Tc, V_in=Vc,
                                   T_out=Tc, V_out=Vc, wv=WView(sT=0, sO=self.cin, sI=1), w_shape=(self.Mg * self.cin,))
        self.spec_r = ops.ConvSpec(M=C, Cin=self.cin, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=Tc, V_in=Vc,
                                   T_out=Tc, V_out=Vc, wv=WView(sT=0, sO=self.cin, sI=1), w_shape=(C * self.cin,))
        if self.T == 1:      # 3 taps over ONE frame: the outer taps read the zero padding, only the centre tap acts
            self.spec_t = ops.ConvSpec(M=C, Cin=C, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=1, V_in=self.V, T_out=1,
                                       V_out=self.V, wv=WView(sT=0, sO=3 * C, sI=3), w_shape=(3 * C * C - 1,))
        else:
            self.spec_t = ops.ConvSpec(M=C, Cin=C, taps=3, tap_mode=TAP_TIME, t_stride=1, T_in=self.T, V_in=self.V,
                                       T_out=self.T, V_out=self.V, wv=WView(sT=1, sO=3 * C, sI=3), w_shape=(3 * C * C,))


class GenTrunkMeta:
    def __init__(self, G, device):
        self.geoms: List[GenBlockGeom] = []
        t, v = 1, G.graph.num_node[G.st_gcn_networks[0].lvl]
        self.ok = True
        for blk in G.st_gcn_networks:
            g = GenBlockGeom(blk, t, v, device)
            self.ok = self.ok and g.ok
            self.geoms.append(g)
            t, v = g.T, g.V
        self.nb = len(self.geoms)
        # parameter layout of GenTrunkFn: per block [wg, wt, bt, nw] + [gam_t, bet_t] + [wr, br, gam_r, bet_r]
        self.poff, off = [], 0
        for g in self.geoms:
            self.poff.append(off)
            off += 4 + (2 if g.bn_t else 0) + (4 if g.res == "conv" else 0)
        self.nparams = off
        # packed per-step adjacency products (GenTrunkFn.forward): per block A_eff (K, V, V) then B = U A_eff (K, Vc, V)
        self.adj_off, off = [], 0
        for g in self.geoms:
            na, nb_ = g.K * g.V * g.V, g.K * g.Vc * g.V
            self.adj_off.append((off, off + na))
            off += na + nb_
        self.adj_numel = off

    def block_params(self, params, i):
        g = self.geoms[i]
        p = params[self.poff[i]:]
        d = dict(wg=p[0], wt=p[1], bt=p[2], nw=p[3], gam_t=None, bet_t=None, wr=None, br=None, gam_r=None, bet_r=None)
        q = 4
        if g.bn_t:
            d["gam_t"], d["bet_t"] = p[q], p[q + 1]
            q += 2
        if g.res == "conv":
            d["wr"], d["br"], d["gam_r"], d["bet_r"] = p[q:q + 4]
        return d


def collect_params(G):
    """Flat parameter list in GenTrunkMeta's layout + the BatchNorm modules per block (None where absent)."""
    params, bns = [], []
    for blk in G.st_gcn_networks:
        params += [blk.gcn.conv.weight, blk.tcn[0].weight, blk.tcn[0].bias, blk.noise.weight]
        bn_t = blk.tcn[1] if len(blk.tcn) > 1 else None
        bn_r = blk.residual[1] if blk.res_kind == "conv" else None
        if bn_t is not None:
            params += [bn_t.weight, bn_t.bias]
        if bn_r is not None:
            params += [blk.residual[0].weight, blk.residual[0].bias, bn_r.weight, bn_r.bias]
        bns.append((bn_t, bn_r))
    return params, bns


def trunk_supported(G, bns) -> bool:
    """The trunk runs training-mode BatchNorm with running statistics and a plain momentum (what the reference's
    generator has, generator.py:142,160); anything else takes the block-wise path."""
    for bn_t, bn_r in bns:
        for b in (bn_t, bn_r):
            if b is not None and (b.running_mean is None or b.momentum is None or not b.affine):
                return False
    return True


def _rowblock_delta(wg: torch.Tensor, wr: torch.Tensor) -> Optional[int]:
    """Distance (elements) from the gcn weight to the residual conv's weight when both live in one buffer (the flat
    parameter buffer of wgan_gp.FlatParams) and the residual one comes later: its rows are then addressed as a second
    row block of ONE weight operand (KgConvGroup.w_sMB); None otherwise."""
    if not (wg.is_contiguous() and wr.is_contiguous()):
        return None
    if wg.untyped_storage().data_ptr() != wr.untyped_storage().data_ptr():
        return None
    d = wr.storage_offset() - wg.storage_offset()
    return d if 0 < d < (1 << 27) else None


def _head_conv(g: GenBlockGeom, x, wg, wr):
    """yc (N, Mh, Tc, Vc) = [W_gcn[:Mg]; W_res] x on the block's input grid."""
    n = x.shape[0]
    if g.res != "conv":
        return nv.conv([Group(x, wg, WView(0, g.cin, 1), g.cin, 1)], n, g.Mg, g.Tc, g.Vc)
    delta = _rowblock_delta(wg, wr)
    if delta is not None:
        return nv.conv([Group(x, wg, WView(0, g.cin, 1, delta, g.Mg), g.cin, 1)], n, g.Mh, g.Tc, g.Vc)
    yc = nv.new_plane(n, g.Mh, g.Tc, g.Vc, x.device)
    nv.conv([Group(x, wg, WView(0, g.cin, 1), g.cin, 1)], n, g.Mg, g.Tc, g.Vc, out=yc[:, :g.Mg])
    nv.conv([Group(x, wr, WView(0, g.cin, 1), g.cin, 1)], n, g.cout, g.Tc, g.Vc, out=yc[:, g.Mg:])
    return yc


def _tcn_weight(g: GenBlockGeom, wt):
    return wt.reshape(-1)[1:] if g.T == 1 else wt


def _bn_job(bn, gamma, beta):
    return dict(gamma=gamma, beta=beta, running_mean=bn.running_mean, running_var=bn.running_var,
                num_batches_tracked=bn.num_batches_tracked, momentum=bn.momentum, eps=bn.eps)


def _finish_tail(pend, groups: int):
    """out = act(BN_t(u) + BN_r(r) + w_noise noise) of a block whose BatchNorm coefficients are known (kg_affine_act)"""
    ct, cr, C = pend["ct"], pend["cr"], pend["u"].shape[1]
    return nv.affine_act(pend["u"], ct[0, 0] if ct is not None else None, ct[0, 1] if ct is not None else None, pend["r"],
                         cr[0, 0] if cr is not None else None, cr[0, 1] if cr is not None else None,
                         pend["noise"], pend["nw"].reshape(-1), pend["act"], SLOPE,
                         groups=groups if (ct is not None or cr is not None) else 1, coef_gs=4 * C)


def _fusable(g: GenBlockGeom, n: int, p, backward: bool = False) -> bool:
    return (FUSED and g.T > 1 and g.Tc * g.Vc >= FUSED_MIN_COLS and
            nv.genblock_supported(g.dims, n, p["wg"], p["wr"] if g.res == "conv" else None, p["wt"], backward=backward))


def fwd_pass(meta: GenTrunkMeta, w, noise, adjs, params, bns, groups: int, keep: bool):
    """x = w.view(N, lat, 1, 1) through the seven blocks.  Returns (out, tape); tape[i] = dict of what the backward
    pass of block i reads (``keep``)."""
    x = w.view(w.shape[0], w.shape[1], 1, 1)
    tape = []
    pend = None         # the previous block's tail, not applied yet (fused blocks: its coefficients are known at launch end)
    nall = w.shape[0]
    for i, g in enumerate(meta.geoms):
        p = meta.block_params(params, i)
        bn_t, bn_r = bns[i]
        n, C = nall, g.cout
        if _fusable(g, n, p):
            res = nv.genblock_fwd(g.dims, x=x if pend is None else None, pend=pend, wg=p["wg"], wr=p["wr"], br=p["br"],
                                  wt=p["wt"], bt=p["bt"], B=adjs[i], U=g.U,
                                  bn_t=_bn_job(bn_t, p["gam_t"], p["bet_t"]) if bn_t is not None else None,
                                  bn_r=_bn_job(bn_r, p["gam_r"], p["bet_r"]) if bn_r is not None else None,
                                  groups=groups, noise=noise[i], nw=p["nw"], slope=SLOPE)
            if pend is not None and keep:
                tape[i - 1]["out"] = res["x"]
            if keep:
                tape.append(dict(x=res["x"], yc=res["yc"], z=res["z"], u=res["u"], r=res["r"], out=res["out"], ct=res["ct"], cr=res["cr"]))
            if res["out"] is not None:
                x, pend = res["out"], None
            else:
                x, pend = None, dict(u=res["u"], r=res["r"], ct=res["ct"], cr=res["cr"], noise=noise[i], nw=p["nw"], act=g.act)
            continue
        if pend is not None:
            x = _finish_tail(pend, groups)
            if keep:
                tape[i - 1]["out"] = x
            pend = None
        yc = _head_conv(g, x, p["wg"], p["wr"])
        rs = yc[:, g.Mg:] if g.res == "conv" else (x if g.res == "identity" else None)
        z, r = nv.gen_expand(yc[:, :g.Mg], None, g.U, g.rep, C, rs=rs, rbias=p["br"] if g.res == "conv" else None, B=adjs[i])
        st = g.spec_t
        u = nv.conv([Group(z, _tcn_weight(g, p["wt"]), st.wv, C, st.taps, TAP_TIME, 1, False, None)], n, C, g.T, g.V,
                    bias0=p["bt"])
        jobs = []
        if bn_t is not None:
            jobs.append(dict(_bn_job(bn_t, p["gam_t"], p["bet_t"]), x=u, groups=groups))
        if bn_r is not None:
            jobs.append(dict(_bn_job(bn_r, p["gam_r"], p["bet_r"]), x=r, groups=groups))
        coefs = nv.bn_fwd_many(jobs) if jobs else []
        ct = coefs[0] if bn_t is not None else None          # (groups, 4, C): scale, shift, mean, rstd
        cr = coefs[-1] if bn_r is not None else None
        out = _finish_tail(dict(u=u, r=r, ct=ct, cr=cr, noise=noise[i], nw=p["nw"], act=g.act), groups)
        if keep:
            tape.append(dict(x=x, yc=yc, z=z, u=u, r=r, out=out, ct=ct, cr=cr))
        x = out
    if pend is not None:
        x = _finish_tail(pend, groups)
        if keep:
            tape[-1]["out"] = x
    return x, tape


def _tail_operands(geo: GenBlockGeom, tp, p, sl, noise_i):
    """what the tail's backward of one block reads: (kwargs of nv.gen_tail_bwd / the `prev` dict of nv.genblock_bwd)"""
    ct, cr = tp["ct"], tp["cr"]
    sinks = dict(nw=ops._sink_of(p["nw"]))
    bn_t = bn_r = None
    if ct is not None:
        bn_t = (p["gam_t"], ct[-1, 2], ct[-1, 3])
        sinks.update(gamma_t=ops._sink_of(p["gam_t"]), beta_t=ops._sink_of(p["bet_t"]))
    if cr is not None:
        bn_r = (p["gam_r"], cr[-1, 2], cr[-1, 3])
        sinks.update(gamma_r=ops._sink_of(p["gam_r"]), beta_r=ops._sink_of(p["bet_r"]))
    r = tp["r"][sl] if tp["r"] is not None else None
    return dict(u=tp["u"][sl] if ct is not None else None, bn_t=bn_t, r=r, bn_r=bn_r, noise=noise_i[sl], sinks=sinks)


def bwd_pass(meta: GenTrunkMeta, tape, g, noise, adjs, params, lo: int, hi: int, imp_sinks, need_gx0: bool = True):
    """Backward over samples [lo, hi) of the taped batch (their BatchNorm batch = the LAST stacked group).  All
    parameter gradients go to the flat-bucket sinks.  Returns d out / d w (hi - lo, lat)."""
    outer_jobs, adj_jobs = [], []
    sl = slice(lo, hi)
    coef_next = None        # tail coefficients of block i, taken by block i + 1's fused backward launch on its way out
    for i in range(meta.nb - 1, -1, -1):
        geo = meta.geoms[i]
        tp = tape[i]
        p = meta.block_params(params, i)
        C, n = geo.cout, hi - lo
        x, yc, z, out = tp["x"][sl], tp["yc"][sl], tp["z"][sl], tp["out"][sl]
        tail = _tail_operands(geo, tp, p, sl, noise[i])
        st = geo.spec_t
        wt_sink = ops._sink_of(p["wt"])
        wg_sink = ops._sink_of(p["wg"])
        if _fusable(geo, n, p, backward=True):
            # ONE launch: tail apply, tcn^T, fold, head^T, and the tail statistics of block i - 1
            coef = coef_next if coef_next is not None else nv.gen_tail_bwd(g, out, geo.act, slope=SLOPE, stats_only=True, **tail)
            prev = None
            if i > 0:
                pg = meta.geoms[i - 1]
                prev = _tail_operands(pg, tape[i - 1], meta.block_params(params, i - 1), sl, noise[i - 1])
                prev.update(x=x, act=pg.act)
            res = nv.genblock_bwd(geo.dims, g=g, out=out, u=tail["u"], r=tail["r"] if tail["bn_r"] is not None else None, coef=coef,
                                  wg=p["wg"], wr=p["wr"], wt=p["wt"], B=adjs[i], U=geo.U, prev=prev, slope=SLOPE)
            coef_next = res["pcoef"]
            du, dr, gyc, zf = res["du"], res["dr"], res["gyc"], res["zf"]
            ops._wgrad_into(wt_sink, z, du, st)
            ops._rowsum_into([ops._sink_of(p["bt"])], du)
            ops._wgrad_into(wg_sink[:geo.Mg * geo.cin], x, gyc[:, :geo.Mg], geo.spec_g)
            if geo.res == "conv":
                ops._wgrad_into(ops._sink_of(p["wr"]), x, gyc[:, geo.Mg:], geo.spec_r)
                ops._rowsum_into([ops._sink_of(p["br"])], dr)
            dbt = nv.agg_outer(zf, yc[:, :geo.Mg], geo.Kp, 1, defer=outer_jobs)
            adj_jobs.append(dict(dbt=dbt, u=geo.U, a=geo.A_fixed, out=imp_sinks[i].view(geo.K, geo.V, geo.V), accumulate=True))
            g = res["gx"]
            continue
        # tail: g * act'(out), both BatchNorm backward passes, the noise weight's gradient - two launches
        # (kg_gen_tail_stats / kg_gen_tail_apply; one when the fused launch of block i + 1 has taken the statistics); the
        # affine parameters' and the noise weight's gradients are added into their bucket slices by the kernel
        du, dr = nv.gen_tail_bwd(g, out, geo.act, slope=SLOPE, coef=coef_next, **tail)
        coef_next = None
        # temporal conv
        ops._wgrad_into(wt_sink[1:] if geo.T == 1 else wt_sink, z, du, st)
        ops._rowsum_into([ops._sink_of(p["bt"])], du)
        gz = nv.conv([Group(du, _tcn_weight(geo, p["wt"]), WView(st.wv.sT, st.wv.sI, st.wv.sO), C, st.taps, TAP_TIME, 1, True, None)],
                     n, C, geo.T, geo.V)
        # head: back to the block's input grid
        gyc = nv.new_plane(n, geo.Mh, geo.Tc, geo.Vc, g.device)
        b_i = adjs[i]                    # (Kp, Vc, V) = U (A[lvl] * importance), kg_gen_adj_prepare
        if geo.res == "conv":
            _, _, zf = nv.gen_fold(gz, None, geo.U, geo.rep, geo.Kp, gr=dr, want_zf=True, y_out=gyc[:, :geo.Mg], rs_out=gyc[:, geo.Mg:], B=b_i)
            gid = None
        elif geo.res == "identity":
            _, gid, zf = nv.gen_fold(gz, None, geo.U, geo.rep, geo.Kp, gr=dr, want_zf=True, y_out=gyc, B=b_i)
        else:
            _, _, zf = nv.gen_fold(gz, None, geo.U, geo.rep, geo.Kp, want_zf=True, y_out=gyc, B=b_i)
            gid = None
        ops._wgrad_into(wg_sink[:geo.Mg * geo.cin], x, gyc[:, :geo.Mg], geo.spec_g)
        if geo.res == "conv":
            ops._wgrad_into(ops._sink_of(p["wr"]), x, gyc[:, geo.Mg:], geo.spec_r)
            ops._rowsum_into([ops._sink_of(p["br"])], dr)
        # adjacency: d B_k^T (Kp, V, Vc) = sum zf[c,(.,w)] yc[k C + c,(.,vc)], finished for all blocks at the end
        dbt = nv.agg_outer(zf, yc[:, :geo.Mg], geo.Kp, 1, defer=outer_jobs)
        adj_jobs.append(dict(dbt=dbt, u=geo.U, a=geo.A_fixed, out=imp_sinks[i].view(geo.K, geo.V, geo.V), accumulate=True))
        if i == 0 and not need_gx0:
            g = None
            break
        grp = [Group(gyc[:, :geo.Mg], p["wg"], WView(0, 1, geo.cin), geo.Mg, 1)]
        if geo.res == "conv":
            grp.append(Group(gyc[:, geo.Mg:], p["wr"], WView(0, 1, geo.cin), C, 1))
        out0 = None
        if i == 0 and geo.Tc == 1 and geo.Vc == 1:
            # d out / d w leaves the trunk as an (n, lat) matrix (the mapping network's backward reads rows): written
            # sample-major here - as a channel-major plane its reshape below was a 146 KB copy launch per iteration
            out0 = torch.empty((n, geo.cin, 1, 1), dtype=torch.float32, device=gyc.device)
        g = nv.conv(grp, n, geo.cin, geo.Tc, geo.Vc, add=gid, out=out0)
    nv.agg_outer_finish(outer_jobs)
    nv.gen_adj_finish(adj_jobs)
    return None if g is None else g.reshape(g.shape[0], -1)


class GenTrunkFn(Function):
    """out = the seven generator blocks on mapped latents.  Arguments: cfg = (meta, bns, groups, A_all), w_all (N, lat)
    - all stacked batches -, w_b (n, lat) | None - the LAST batch with history when w_all carries none
    (``Generator.synthesis_pair``; None: w_all itself is differentiated), the seven noise planes, the seven
    edge_importance parameters, the block parameters (GenTrunkMeta layout).  Returns out, or (out of all batches -
    non-differentiable -, its differentiated tail) with w_b."""

    @staticmethod
    def forward(ctx, cfg, w_all, w_b, *rest):
        meta, bns, groups = cfg
        nb = meta.nb
        noise, imps, params = list(rest[:nb]), list(rest[nb:2 * nb]), list(rest[2 * nb:])
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad[1:])
        with torch.no_grad():
            # A[lvl] * importance and U (A[lvl] * importance) of all seven blocks: one launch
            pack = torch.empty(meta.adj_numel, dtype=torch.float32, device=w_all.device)
            adjs, jobs = [], []
            for g, imp, (oa, ob) in zip(meta.geoms, imps, meta.adj_off):
                k, v, vc = g.K, g.V, g.Vc
                ae = pack[oa:oa + k * v * v].view(k, v, v)
                b = pack[ob:ob + k * vc * v].view(k, vc, v)
                jobs.append(dict(a=g.A_fixed, imp=imp.detach(), u=g.U, aeff=ae, b=b))
                adjs.append(b[:g.Kp])
            nv.gen_adj_prepare(jobs)
            out, tape = fwd_pass(meta, w_all.detach(), [t.detach() for t in noise], adjs, [p.detach() for p in params], bns,
                                 groups, keep=need)
        ctx.meta, ctx.tape, ctx.adjs, ctx.noise = meta, tape, adjs, noise
        ctx.n_all = w_all.shape[0]
        ctx.n_b = w_b.shape[0] if w_b is not None else ctx.n_all
        ctx.paired = w_b is not None
        ctx.imp_sinks = [ops._sink_of(p) for p in imps]
        ctx.save_for_backward(*params)
        if w_b is None:
            return out
        ctx.mark_non_differentiable(out)
        return out, out[ctx.n_all - ctx.n_b:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        meta = ctx.meta
        nret = 3 + 2 * meta.nb + meta.nparams
        g = gs[-1]
        if g is None:
            return (None,) * nret
        params = list(ctx.saved_tensors)
        lo = ctx.n_all - ctx.n_b
        need_w = ctx.needs_input_grad[2] if ctx.paired else ctx.needs_input_grad[1]
        with torch.no_grad():
            gw = bwd_pass(meta, ctx.tape, nv.as_plane(g), [t.detach() for t in ctx.noise], ctx.adjs, params, lo, ctx.n_all,
                          ctx.imp_sinks, need_gx0=need_w)
        ctx.tape = None
        if ctx.paired:
            return (None, None, gw) + (None,) * (nret - 3)
        return (None, gw, None) + (None,) * (nret - 3)


def all_sinks_registered(G) -> bool:
    return all(ops._sink_of(p) is not None for p in G.parameters() if p.requires_grad)
