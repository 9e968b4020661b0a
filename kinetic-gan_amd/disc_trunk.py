"""Hand-scheduled forward / backward / double backward of the discriminator's six st_gcn blocks.

``Discriminator.forward`` (discriminator.py:52-74 of the reference) is a chain of six blocks; evaluated block by
block through ``ops.py`` every block boundary costs autograd bookkeeping launches (accumulation adds of the
twice-used block input, masks, index copies).  Here the whole chain is ONE autograd node whose three passes are
straight launch sequences over the C ABI:

  FWD  x -> h          per block:  xa = x A_k[:, :, keep]  ->  z = W_gcn xa (+ label bias, block 0)
                                   -> out = lrelu(W_tcn * z + W_res x[keep] + b)      at the kept (t, v) only
  BWD  g -> gx         per block:  gm = g * lrelu'(out) -> gz = W_tcn^T * gm -> gxa = W_gcn^T gz
                                   -> gx = gxa A_k^T + W_res^T gm                     (+ parameter gradients)
  DBL  h -> gg         the adjoint of BWD w.r.t. g, i.e. FWD linearised: lrelu -> multiplication by lrelu'(out),
                       no biases; needed because the WGAN-GP penalty (kinetic-gan.py:94-114) differentiates
                       THROUGH the backward pass w.r.t. the weights and edge_importance.

``DiscTrunkFn`` (FWD, its backward = BWD) and ``DiscTrunkBwdFn`` (BWD as a differentiable forward, its backward
= DBL) close the family: that is all three derivatives kinetic-gan.py:137-174 ever takes of D.  The trunk accepts
one or two batches ("parts": the critic step runs D on the real+fake batch and on the interpolates between two
optimiser steps) and runs them as ONE launch sequence over the concatenated batch; a backward pass touches only
the samples of the part(s) whose gradient arrived.
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch
from torch.autograd import Function

from . import _native as nv
from . import ops
from ._native import ACT_LRELU, ACT_NONE, TAP_TIME, Group, WView

SLOPE = 0.2


class BlockGeom:
    """Static description of one block for one input geometry (built by Discriminator._trunk_meta)."""

    def __init__(self, blk, T, V, device, const_channels=0):
        p = blk._plan(T, V, device)
        self.res = blk.res_kind
        self.cin, self.cout, self.K = blk.in_channels, blk.out_channels, blk.gcn.kernel_size
        self.dw_s = blk.dw_s
        self.keep_l = p["keep"]
        self.inv_keep = None
        if blk.dw_s:
            inv = torch.full((V,), -1, dtype=torch.int32)
            inv[p["keep"].cpu()] = torch.arange(len(p["keep"]), dtype=torch.int32)
            self.inv_keep = inv.to(device)
        self.T, self.V = T, V
        self.W = len(p["keep"]) if blk.dw_s else V
        self.stride, self.t_out = p["stride"], p["t_out"]
        self.spec_t, self.spec_r = p["spec_t"], p["spec_r"]
        self.cc = const_channels
        if const_channels:
            # the data channels sit BEHIND the constant ones in the weight's input dimension: same strides as the
            # parent weight, base pointer moved by `cc` (no slice copy; the gradient goes to the parent's slice)
            cd = self.cin - const_channels
            self.spec_g = ops.ConvSpec(M=self.cout, Cin=cd, taps=self.K, tap_mode=nv.TAP_CHANBLOCK, t_stride=1,
                                       T_in=T, V_in=self.W, T_out=T, V_out=self.W,
                                       wv=WView(sT=self.cout * self.cin, sO=self.cin, sI=1),
                                       w_shape=(self.K * self.cout * self.cin - const_channels,))
        else:
            self.spec_g = p["spec_g"]
        self.nparam = 5 if self.res == "conv" else 3
        # fixed sparsity pattern of the block's (kept-column) adjacency as a neighbour table for kg_aggconv
        import numpy as np
        pat = np.asarray(blk.graph.As[blk.lvl]) != 0
        if blk.dw_s:
            pat = pat[:, :, self.keep_l.cpu().numpy()]
        tab = np.full((self.K, self.W, nv.AGGCONV_P), -1, dtype=np.int32)
        self.pcount = [0, 0, 0]
        self.fusable = True
        for k in range(self.K):
            for w in range(self.W):
                vs = np.nonzero(pat[k, :, w])[0]
                self.pcount[k] = max(self.pcount[k], len(vs))
                if len(vs) > nv.AGGCONV_P:
                    self.fusable = False
                    vs = vs[:nv.AGGCONV_P]
                tab[k, w, :len(vs)] = vs
        self.nbr = torch.as_tensor(tab, device=device)
        # Partitions without any non-zero in the kept columns contribute nothing (and receive no gradient): at the
        # single-vertex level A = [[1], [0], [0]], so the gcn there is ONE 1x1 conv with the first C_out
        # rows of its weight instead of three - 2/3 of that block's gcn flops in every pass are multiplications by 0.
        self.single = self.K == 3 and self.pcount[0] > 0 and self.pcount[1] == 0 and self.pcount[2] == 0 and not const_channels
        if self.single:
            self.spec_g1 = ops.ConvSpec(M=self.cout, Cin=self.cin, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=T, V_in=self.W,
                                        T_out=T, V_out=self.W, wv=WView(sT=0, sO=self.cin, sI=1),
                                        w_shape=(self.cout * self.cin,))

    def fused_gcn(self, ncols: int) -> bool:
        return self.fusable and nv.aggconv_supported(self.V, self.W, self.pcount, ncols)


class TrunkMeta:
    def __init__(self, geoms: List[BlockGeom], A_list=None, head: bool = False, label_bias: bool = False):
        """``head``: the node also holds global average pool + Linear(latent, 1) (discriminator.py:68-72): its outputs
        are validities, two more parameters (fcn.weight, fcn.bias) follow the blocks'.  ``label_bias``: the node computes
        block 0's label bias itself from the class labels (its 4th argument) and label_emb.weight (the last parameter)."""
        self.geoms = geoms
        self.nb = len(geoms)
        self.head, self.lb = bool(head), bool(label_bias) and geoms[0].cc > 0
        self.poff = []
        off = 0
        for g in geoms:
            self.poff.append(off)
            off += g.nparam
        self.nparams = off
        self.any_single = any(g.single for g in geoms)
        # Packed adjacencies: the six masked, kept-column adjacencies A[lvl] * importance [:, :, keep] live in ONE flat
        # tensor `ak_all` (block i at ak_off[i], shape ak_shape[i]); `sel` maps its elements into the flat
        # concatenation of the full adjacencies (`A_all`, same order as the importance parameters).
        self.ak_off, self.ak_shape, self.a_off = [], [], []
        if A_list is not None:
            dev = A_list[0].device
            a_parts, sel_parts = [], []
            aoff = koff = 0
            for g, A in zip(geoms, A_list):
                K, V, _ = A.shape
                idx = torch.arange(A.numel(), device=dev).view(K, V, V)
                if g.dw_s:
                    idx = idx[:, :, g.keep_l]
                sel_parts.append((idx + aoff).reshape(-1))
                a_parts.append(A.reshape(-1))
                self.a_off.append(aoff)
                self.ak_off.append(koff)
                self.ak_shape.append((K, V, g.W))
                aoff += A.numel()
                koff += K * V * g.W
            self.A_all = torch.cat(a_parts).contiguous()
            self.sel = torch.cat(sel_parts).contiguous()
            self.A_sel = self.A_all[self.sel].contiguous()
            self.ak_numel = koff

    def ak_views(self, ak_all):
        return [ak_all[o:o + s[0] * s[1] * s[2]].view(s) for o, s in zip(self.ak_off, self.ak_shape)]


def _pack(params):
    """One flat tensor over the given parameters: a zero-copy view when they sit back to back in one buffer (the
    flat parameter buffer of wgan_gp.FlatParams keeps a ParameterList's entries adjacent), else a concatenation."""
    p0 = params[0]
    off = p0.storage_offset()
    ok = all(p.is_contiguous() and p.untyped_storage().data_ptr() == p0.untyped_storage().data_ptr() for p in params)
    if ok:
        for p in params:
            if p.storage_offset() != off:
                ok = False
                break
            off += p.numel()
    total = sum(p.numel() for p in params)
    if ok:
        return torch.as_strided(p0.detach(), (total,), (1,), p0.storage_offset())
    return torch.cat([p.detach().reshape(-1) for p in params])


class AdjacencyPack:
    """The `meta` of MaskedAdjacencyFn for a plain list of full adjacencies (the generator: no kept columns)."""

    def __init__(self, A_list):
        self.A_all = torch.cat([A.reshape(-1) for A in A_list]).contiguous()
        self.sel = None
        self.split = True
        self.ak_off, self.ak_shape = [], []
        off = 0
        for A in A_list:
            self.ak_off.append(off)
            self.ak_shape.append(tuple(A.shape))
            off += A.numel()

    def ak_views(self, ak_all):
        return [ak_all[o:o + s[0] * s[1] * s[2]].view(s) for o, s in zip(self.ak_off, self.ak_shape)]


class MaskedAdjacencyFn(Function):
    """ak_all = (A_all * importance_all)[sel]: the effective adjacencies of all six blocks (generator.py:92-93 /
    discriminator.py:63-64: ``self.A[lvl] * importance``) and their kept-column restriction in two launches; the
    backward pass (d importance = A * scatter(d ak)) in three, instead of a mul + an advanced-index gather per
    block whose autograd backward sorts indices."""

    @staticmethod
    def forward(ctx, meta, *importances):
        """Returns the packed tensor, or - for a meta with ``split`` set (the generator, whose blocks take their
        adjacency one by one) - the per-block views of it as separate outputs (their gradients are concatenated in
        one launch instead of being scattered into seven zero-filled copies by autograd's view backward)."""
        imp_all = _pack(importances)
        ctx.meta = meta
        ctx.shapes = [tuple(p.shape) for p in importances]
        ctx.sinks = [ops._sink_of(p) for p in importances]
        ctx.split = bool(getattr(meta, "split", False))
        ak = nv.masked_adj_fwd(meta.A_all, imp_all, meta.sel)          # one launch (was mul + index_select)
        if ctx.split:
            ctx.set_materialize_grads(False)
            return tuple(meta.ak_views(ak))
        return ak

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        meta = ctx.meta
        ops.flush_outer()      # adjacency gradients recorded by AggReduce.backward(lazy_outer=True): computed now
        if ctx.split:
            if all(g is None for g in gs):
                return (None,) * (1 + len(ctx.shapes))
            g = torch.cat([(g if g is not None else torch.zeros(s, device=meta.A_all.device)).reshape(-1)
                           for g, s in zip(gs, meta.ak_shape)])
        else:
            g = gs[0]
        if g is None:
            return (None,) * (1 + len(ctx.shapes))
        sinks = ctx.sinks
        if all(v is not None for v in sinks):
            adjacent = all(sinks[i + 1].data_ptr() == sinks[i].data_ptr() + 4 * sinks[i].numel() for i in range(len(sinks) - 1))
            if adjacent:        # the importance gradients are one contiguous slice of the flat bucket: ONE launch adds
                flat = torch.as_strided(sinks[0], (meta.A_all.numel(),), (1,), sinks[0].storage_offset())
                nv.masked_adj_bwd(g, meta.A_all, meta.sel, flat, accumulate=True)
                return (None,) * (1 + len(ctx.shapes))
        dimp = torch.zeros_like(meta.A_all) if meta.sel is not None else torch.empty_like(meta.A_all)
        nv.masked_adj_bwd(g, meta.A_all, meta.sel, dimp, accumulate=False)
        outs, off = [], 0
        for shp in ctx.shapes:
            n = shp[0] * shp[1] * shp[2]
            outs.append(dimp[off:off + n].view(shp))
            off += n
        return (None,) + tuple(outs)


def _wg_view(geom: BlockGeom, wg: torch.Tensor) -> torch.Tensor:
    return wg.reshape(-1)[geom.cc:] if geom.cc else wg


def _sub(t: Optional[torch.Tensor], lo: int, hi: int):
    return None if t is None else t[lo:hi]


# ---- the three passes (plain launch sequences; no autograd inside) -----------------------------------------------------

def _gcn(geom: BlockGeom, xa, wg, add=None, add_tstride=1):
    sp = geom.spec_g
    grp = Group(xa, _wg_view(geom, wg), sp.wv, sp.Cin, sp.taps, sp.tap_mode, 1, False, None)
    return nv.conv([grp], xa.shape[0], sp.M, sp.T_out, sp.V_out, add=add, add_tstride=add_tstride)


def _tail(geom: BlockGeom, z, x, wt, bt, wr, br, linear: bool, mask=None):
    """lrelu(tcn(z) + residual(x) + biases) at the kept frames / vertices; ``linear``: no biases, no activation,
    the result times lrelu'(mask) (the double backward's linearised block)."""
    st, sr = geom.spec_t, geom.spec_r
    groups = [Group(z, wt, st.wv, st.Cin, 3, TAP_TIME, st.t_stride, False, None)]
    add = None
    if geom.res == "conv":
        groups.append(Group(x, wr, sr.wv, sr.Cin, 1, TAP_TIME, sr.t_stride, False, sr.vmap))
    elif geom.res == "identity":
        add = x[:, :, :, geom.keep_l] if geom.dw_s else x
    return nv.conv(groups, z.shape[0], st.M, st.T_out, st.V_out,
                   bias0=None if linear else bt, bias1=None if (linear or geom.res != "conv") else br,
                   add=add, add_tstride=st.t_stride,
                   act=ACT_NONE if linear else ACT_LRELU, slope=SLOPE, mask=mask)


def _tcn_transposed_jobs(gm, wt, st):
    """gz = W_tcn^T * gm (input gradient of the temporal conv) as a list of kg_conv problems + the tensor they fill.  With
    frame stride 2 an even input frame 2u is reached by the centre tap alone (from output frame u) and an odd one 2u+1
    by the two outer taps (from u and u+1): the strided transposed conv as ONE problem multiplies zeros for half of its
    (frame, tap) pairs, so it is two problems, one per frame parity, each writing every other frame of gz
    (KgConvArgs.o_tstride): half the MFMA work.  The problems are independent of each other (and of the residual
    branch's product, which reads the same gm): the caller hands them to ONE kg_conv_many launch."""
    n = gm.shape[0]
    gz = nv.new_plane(n, st.Cin, st.T_in, st.V_in, gm.device)
    if not (st.t_stride == 2 and st.T_in % 2 == 0 and st.T_in >= 4 and wt.is_contiguous()):
        return [dict(groups=[Group(gm, wt, WView(st.wv.sT, st.wv.sI, st.wv.sO), st.M, 3, TAP_TIME, st.t_stride, True, None)],
                     N=n, M=st.Cin, T_out=st.T_in, V_out=st.V_in, out=gz)], gz
    flat = wt.reshape(-1)
    wv = WView(0, st.wv.sI, st.wv.sO)          # rows = input channels, contraction over the output channels
    half = st.T_in // 2                          # = frames of gm
    tap = [flat[d * st.wv.sT:] for d in range(3)]
    return [dict(groups=[Group(gm, tap[1], wv, st.M, 1)], N=n, M=st.Cin, T_out=half, V_out=st.V_in, out=gz, out_t0=0, out_tstride=2),
            dict(groups=[Group(gm, tap[2], wv, st.M, 1), Group(gm[:, :, 1:], tap[0], wv, st.M, 1)], N=n, M=st.Cin, T_out=half,
                 V_out=st.V_in, out=gz, out_t0=1, out_tstride=2)], gz


def _tcn_transposed(gm, wt, st):
    jobs, gz = _tcn_transposed_jobs(gm, wt, st)
    nv.conv_many(jobs)
    return gz


def _agg_gcn(g: BlockGeom, x, ak, wg, add, want_xa: bool):
    """z = sum_k W_k (x A_k) (+ per-sample bias): ONE fused launch where the geometry allows, else expand + conv.
    Returns (z, xa | None)."""
    tstride = 0 if add is not None else 1
    if g.single:
        xa = nv.agg_expand(x, ak[:1], 1)
        sp = g.spec_g1
        return nv.conv([Group(xa, wg, sp.wv, sp.Cin, 1)], x.shape[0], sp.M, sp.T_out, sp.V_out, add=add, add_tstride=tstride), xa
    # (round 4, tools/time_aggconv.py: with 256 output channels - four 64-row tiles, each repeating the aggregation - and the
    # aggregated planes written out for the weight gradient, the fused launch is 4-10 us SLOWER than expand + conv at every
    # batch size: 43.1 / 70.8 / 80.5 us against 38.8 / 60.9 / 76.0 us at 64 / 128 / 192 samples; block 3 takes the pair then)
    if g.fused_gcn(x.shape[0] * x.shape[2] * g.W) and not (want_xa and g.cout >= 256):
        sp = g.spec_g
        return nv.aggconv(x, ak, g.nbr, g.pcount, _wg_view(g, wg), WView(sp.wv.sT, sp.wv.sO, sp.wv.sI), sp.M,
                          add=add, add_tstride=tstride, want_xa=want_xa)
    xa = nv.agg_expand(x, ak, 1)
    return _gcn(g, xa, wg, add=add, add_tstride=tstride), xa


def fwd_pass(meta: TrunkMeta, x, zl, aks, params, want_xa: bool = True):
    """Returns (h, tape); tape[i] = (x_i, xa_i | None, z_i, out_i)."""
    tape = []
    for i, g in enumerate(meta.geoms):
        wg, wt, bt = params[meta.poff[i]:meta.poff[i] + 3]
        wr, br = (params[meta.poff[i] + 3], params[meta.poff[i] + 4]) if g.res == "conv" else (None, None)
        # block 0: per-sample label bias, broadcast over the frames
        z, xa = _agg_gcn(g, x, aks[i], wg, zl if g.cc else None, want_xa)
        out = _tail(g, z, x, wt, bt, wr, br, linear=False)
        # the tape lives on the autograd context as a plain attribute: it must not hold the very tensor OBJECT the
        # Function returns (output -> grad_fn -> ctx -> tape -> output would be a reference cycle that keeps the whole
        # upstream graph, e.g. the generator's, alive until the cyclic collector runs), hence the alias
        tape.append((x, xa, z, out.detach()))
        x = out
    return x, tape


def _rows(t, r):
    return t if (r is None or t is None) else t[r[0]:r[1]]


def bwd_pass(meta: TrunkMeta, tape, g, aks, params, need_gx0: bool, want_params: bool, keep: bool,
             use_sink: bool = True, prow=None, krow=None, xrow=None, top_masked: bool = False, lb=None):
    """BWD over the samples the tape slices cover.  Returns (gx0 | None, gzl | None, dAk list | None, param grads
    list | None, tape2 | None); tape2[i] = (gm_i, gz_i, gxa_i) for the double backward.  Parameter gradients go to
    the flat-bucket sink where one is registered (returned entry None), else they are returned.
    Sample sub-ranges (lo, hi) of the pass (the merged critic backward, DiscTrunkFn): ``prow`` - the samples whose
    parameter / adjacency / label-bias gradients are wanted, ``krow`` - the samples kept in tape2, ``xrow`` - the
    samples whose input gradient gx0 is computed.
    ``top_masked``: g already carries the last block's LeakyReLU derivative (kg_head_bwd).  ``lb`` (a trunk that
    computes the label bias itself): dict(labels (of the pass's samples), emb, J) - the label channels' three gradients
    are taken here from block 0's gz (kg_label_bias_bwd); the returned `gzl` is then (d emb | None)."""
    nb = meta.nb
    dak = None
    if want_params:
        # every block writes its slice (a single-partition block only its first partition: the others stay zero)
        dak = (torch.zeros if meta.any_single else torch.empty)(meta.ak_numel, dtype=torch.float32, device=g.device)
    dviews = meta.ak_views(dak) if dak is not None else None
    pgr = [None] * meta.nparams if want_params else None
    tape2 = [None] * nb
    gzl = None
    outer_jobs = []          # the six adjacency-gradient slab sums finish in one launch
    masked = bool(top_masked)          # g already multiplied by lrelu'(out_i) by the launch that produced it
    lb_job = None
    for i in range(nb - 1, -1, -1):
        geo = meta.geoms[i]
        x, xa, z, out = tape[i]
        wg, wt, bt = params[meta.poff[i]:meta.poff[i] + 3]
        wr = params[meta.poff[i] + 3] if geo.res == "conv" else None
        br = params[meta.poff[i] + 4] if geo.res == "conv" else None
        st, sr, sg = geo.spec_t, geo.spec_r, geo.spec_g
        gm = g if masked else nv.act_bwd(g, out, ACT_LRELU, SLOPE)
        masked = False
        need_gx = i > 0 or need_gx0
        # the contractions that read gm and nothing else of this block - the transposed temporal conv (one or two frame
        # parities) and the small dense product of a down-sampling residual branch - share ONE launch (kg_conv_many)
        jobs, gz = _tcn_transposed_jobs(gm, wt, st)
        sel = xrow if i == 0 else None          # the trunk's input gradient: only for the samples that want it
        gm_s = _rows(gm, sel)
        res_scatter = need_gx and geo.res == "conv" and (sr.t_stride > 1 or sr.inv_vmap is not None)
        if res_scatter:
            jobs.append(dict(groups=[Group(gm_s, wr, WView(0, sr.wv.sI, sr.wv.sO), sr.M, 1)], N=gm_s.shape[0], M=sr.Cin,
                             T_out=gm_s.shape[2], V_out=gm_s.shape[3]))
        outs = nv.conv_many(jobs)
        rs = outs[-1] if res_scatter else None
        gxa = None
        ak_i = aks[i][:1] if geo.single else aks[i]
        if need_gx or want_params:
            if geo.single:
                s1 = geo.spec_g1
                gxa = nv.conv([Group(gz, wg, WView(0, s1.wv.sI, s1.wv.sO), s1.M, 1)], gz.shape[0], s1.Cin, s1.T_in, s1.V_in)
            else:
                gxa = nv.conv([Group(gz, _wg_view(geo, wg), WView(0, sg.wv.sI, sg.wv.sO, sg.wv.sT, sg.Cin), sg.M, 1)],
                              gz.shape[0], sg.Cin * sg.taps, sg.T_in, sg.V_in)
        gx = None
        if need_gx:
            # gx = sum_k gxa_k A_k^T; the residual branch's input gradient and the LeakyReLU derivative of the block
            # input (= the previous block's activation output: no separate g * act'(out) pass for block i-1) ride
            # in the same launch's epilogue (kg_agg_reduce res / mask)
            gxa_s, akT = _rows(gxa, sel), ak_i.transpose(1, 2)
            if geo.res == "conv":
                masked = i > 0
                mask_t = tape[i - 1][3] if masked else None
                if res_scatter:
                    # a down-sampling block's residual reads x at the kept frames / vertices only: rs is the small dense
                    # product at the block's OUTPUT resolution (as one transposed conv over all of gx's columns
                    # 50-90 % of its MFMAs multiplied zeros), added where it lands
                    gx = nv.agg_reduce(gxa_s, akT, 1, res=rs, res_tstride=sr.t_stride, res_inv=sr.inv_vmap, mask=mask_t, slope=SLOPE)
                else:
                    gx = nv.agg_reduce(gxa_s, akT, 1)
                    gx = nv.conv([Group(gm_s, wr, WView(0, sr.wv.sI, sr.wv.sO), sr.M, 1, TAP_TIME, sr.t_stride, True,
                                        sr.inv_vmap)], gm_s.shape[0], sr.Cin, sr.T_in, sr.V_in, add=gx,
                                 mask=mask_t, slope=SLOPE)
            elif geo.res == "identity":
                # (identity residual: gm itself lands on the kept frames / vertices)
                masked = i > 0
                gx = nv.agg_reduce(gxa_s, akT, 1, res=gm_s, res_tstride=geo.stride, res_inv=geo.inv_keep,
                                   mask=tape[i - 1][3] if masked else None, slope=SLOPE)
            else:
                gx = nv.agg_reduce(gxa_s, akT, 1)
        if want_params:
            po = meta.poff[i]
            xp, zp, gmp, gzp, gxap = _rows(x, prow), _rows(z, prow), _rows(gm, prow), _rows(gz, prow), _rows(gxa, prow)
            # the forward pass may not have kept the aggregated planes
            xap = nv.agg_expand(xp, ak_i, 1) if xa is None else _rows(xa, prow)
            sk = ops._sink_of if use_sink else (lambda t: None)
            pgr[po + 0] = _gcn_wgrad(geo, use_sink, xap, gzp, wg)
            pgr[po + 1] = _param_wgrad(sk(wt), zp, gmp, st, wt)
            if geo.res == "conv":
                pgr[po + 3] = _param_wgrad(sk(wr), xp, gmp, sr, wr)
            s_bt = sk(bt)
            s_br = sk(br) if br is not None else None
            if s_bt is not None and (br is None or s_br is not None):
                ops._rowsum_into([s_bt] + ([s_br] if s_br is not None else []), gmp)
            else:
                gb = nv.rowsum(gmp)[0]
                pgr[po + 2] = gb
                if br is not None:
                    pgr[po + 4] = gb
            nv.agg_outer(xp, gxap, 1 if geo.single else geo.K, 1, out=dviews[i][:1] if geo.single else dviews[i], defer=outer_jobs)
            if geo.cc and lb is not None:
                lb_job = (gzp, geo, wg, pgr, po)      # after the adjacency outer products (it ADDS into block 0's)
            elif geo.cc:
                gzl = gzp.sum(2, keepdim=True)        # gradient of the per-sample label bias (N, Cout, 1, W)
        if keep:
            tape2[i] = (_rows(gm, krow), _rows(gz, krow), _rows(gxa, krow))
        g = gx
    if outer_jobs:
        nv.agg_outer_finish(outer_jobs)
    if lb_job is not None:
        gzp, geo, wg, pgr_, po = lb_job
        emb = lb["emb"]
        labels = lb["labels"] if prow is None else lb["labels"][prow[0]:prow[1]]
        s_emb = ops._sink_of(emb) if use_sink else None
        s_wg = ops._sink_of(wg) if use_sink else None
        demb = s_emb.view(emb.shape) if s_emb is not None else torch.zeros_like(emb)
        if s_wg is not None:
            dwg = s_wg
        else:
            dwg = pgr_[po + 0]                           # _gcn_wgrad wrote the data columns into a zero-filled tensor
            if dwg is None:
                dwg = pgr_[po + 0] = torch.zeros_like(wg)
        nv.label_bias_bwd(gzp, labels, emb, wg, geo.K, geo.cout, geo.cin, geo.cc, aks[0].contiguous(),
                          demb=demb, dw=dwg.reshape(-1), dak=dviews[0], accumulate=True, dak_accumulate=True)
        gzl = None if s_emb is not None else demb
    return g, gzl, dak, pgr, (tape2 if keep else None)


def dbl_pass(meta: TrunkMeta, outs, tape2, h, aks, params, want_params: bool = True):
    """DBL: the adjoint of BWD.  h = cotangent of gx0; returns (cotangent of the top gradient, dAk list, param
    grads list)."""
    dak = (torch.zeros if meta.any_single else torch.empty)(meta.ak_numel, dtype=torch.float32, device=h.device) if want_params else None
    dviews = meta.ak_views(dak) if dak is not None else None
    pgr = [None] * meta.nparams
    outer_jobs = []
    for i, geo in enumerate(meta.geoms):
        gm, gz, gxa = tape2[i]
        wg, wt, bt = params[meta.poff[i]:meta.poff[i] + 3]
        wr = params[meta.poff[i] + 3] if geo.res == "conv" else None
        st, sr, sg = geo.spec_t, geo.spec_r, geo.spec_g
        z, xa = _agg_gcn(geo, h, aks[i], wg, None, want_params)
        u = _tail(geo, z, h, wt, None, wr, None, linear=True, mask=outs[i])
        if want_params:
            po = meta.poff[i]
            pgr[po + 0] = _gcn_wgrad(geo, True, xa, gz, wg)
            pgr[po + 1] = _param_wgrad(ops._sink_of(wt), z, gm, st, wt)
            if geo.res == "conv":
                pgr[po + 3] = _param_wgrad(ops._sink_of(wr), h, gm, sr, wr)
            nv.agg_outer(h, gxa, 1 if geo.single else geo.K, 1, out=dviews[i][:1] if geo.single else dviews[i], defer=outer_jobs)
        h = u
    if outer_jobs:
        nv.agg_outer_finish(outer_jobs)
    return h, dak, pgr


def _gcn_wgrad(geo: BlockGeom, use_sink: bool, xa, gz, wg):
    """weight gradient of the block's gcn conv (into the sink, or returned)"""
    if geo.single:          # only the first partition's rows of the weight take part (the others' gradient is zero)
        n1 = geo.cout * geo.cin
        v = ops._sink_of(wg) if use_sink else None
        if v is not None:
            ops._wgrad_into(v[:n1], xa, gz, geo.spec_g1)
            return None
        full = torch.zeros(wg.numel(), dtype=wg.dtype, device=wg.device)
        s1 = geo.spec_g1
        nv.wgrad(gz, xa, s1.Cin, 1, s1.tap_mode, 1, None, n1, WView(s1.wv.sT, s1.wv.sO, s1.wv.sI), out=full[:n1])
        return full.view(wg.shape)
    return _param_wgrad(_sink_view(wg, geo.cc) if use_sink else None, xa, gz, geo.spec_g, wg, geo.cc)


def _sink_view(wg, cc):
    v = ops._sink_of(wg)
    if v is None or not cc:
        return v
    return v[cc:]


def _param_wgrad(view, x, g, spec, w, cc: int = 0):
    """Weight gradient of one layer: deferred into the flat-bucket sink (returns None) or computed now."""
    if view is not None:
        ops._wgrad_into(view, x, g, spec)
        return None
    wv = WView(spec.wv.sT, spec.wv.sO, spec.wv.sI)
    if not cc:
        flat = nv.wgrad(g, x, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, spec.vmap, w.numel(), wv)
        return flat.view(w.shape)
    full = torch.zeros(w.numel(), dtype=w.dtype, device=w.device)
    nv.wgrad(g, x, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, spec.vmap, w.numel() - cc, wv, out=full[cc:])
    return full.view(w.shape)


# ---- autograd nodes ---------------------------------------------------------------------------------------------------

_CHECK_PROMISE = os.environ.get("KG_TRUNK_CHECK", "0") == "1"      # tests: compare the promised gradient (synchronises)

def _join_parts(parts):
    """One tensor over the concatenated batch.  Adjacent views of one buffer are joined without a copy."""
    if len(parts) == 1:
        return parts[0]
    a, b = parts
    if (a.is_contiguous() and b.is_contiguous() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel() and a.shape[1:] == b.shape[1:]):
        return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat((a, b), 0)


def _split_params(meta: TrunkMeta, params):
    """(block parameters, fcn weight | None, fcn bias | None, label_emb.weight | None)"""
    n = meta.nparams
    blk, extra = list(params[:n]), list(params[n:])
    fw = fb = emb = None
    if meta.head:
        fw, fb = extra[0], extra[1]
        extra = extra[2:]
    if meta.lb:
        emb = extra[0]
    return blk, fw, fb, emb


def _n_extra(meta: TrunkMeta) -> int:
    return (2 if meta.head else 0) + (1 if meta.lb else 0)


def _extra_grads(meta: TrunkMeta, dfw=None, dfb=None, demb=None):
    out = []
    if meta.head:
        out += [dfw, dfb]
    if meta.lb:
        out += [demb]
    return out


def _head_param_grads(fw, fb, x, gv, use_sink: bool = True):
    """Linear(latent, 1)'s gradients from the pooled operand x and d loss / d validity gv: into the sinks (returns
    (None, None)) or as new tensors."""
    sw = ops._sink_of(fw) if use_sink else None
    sb = ops._sink_of(fb) if (use_sink and fb is not None) else None
    dw = sw if sw is not None else torch.zeros(fw.numel(), dtype=torch.float32, device=fw.device)
    db = None
    if fb is not None:
        db = sb if sb is not None else torch.zeros(1, dtype=torch.float32, device=fw.device)
    nv.head_wgrad(x, gv, dw, db, accumulate=True)
    return (None if sw is not None else dw.view(fw.shape)), (None if (sb is not None or fb is None) else db.view(fb.shape))


class DiscTrunkFn(Function):
    """h_parts = trunk(x_parts) - or, for a meta with ``head``, their validities.  Arguments: meta, x_a, x_b | None,
    zl | None (meta.lb: the int64 class labels of all samples instead - the label bias is computed inside), the packed
    kept-column adjacencies (MaskedAdjacencyFn), the block parameters (Wg, Wt, bt[, Wr, br]) x 6, then fcn.weight,
    fcn.bias (meta.head) and label_emb.weight (meta.lb)."""

    @staticmethod
    def forward(ctx, meta: TrunkMeta, x_a, x_b, zl, ak_all, *params):
        ctx.set_materialize_grads(False)
        ctx.ga = None
        if isinstance(meta, tuple):
            # (meta, ga): the caller PROMISES that the gradient arriving for part a will be `ga` - d loss / d validity
            # (n_a,) for a trunk with head, else d loss / d h_a broadcast over frames and vertices.  The WGAN critic loss
            # is linear in D(real), D(fake), so wgan_gp.Trainer knows it before the backward pass starts.  The first
            # differentiable backward call for part b (the gradient penalty's d D(inter) / d inter) then runs ONE pass
            # over all samples: part a's parameter gradients are taken along (better-filled launches, one launch
            # sequence less) and the later call for part a only hands over the stored adjacency gradients.
            meta, ctx.ga = meta
        ctx.merged = None
        params = list(params)
        blk, fw, fb, emb = _split_params(meta, params)
        aks = meta.ak_views(ak_all.detach())
        parts = [x_a] if x_b is None else [x_a, x_b]
        x = _join_parts([p.detach() for p in parts])
        # the aggregated planes are only kept (written at all, on the fused path) when a weight gradient may follow
        want_xa = any(ctx.needs_input_grad[5:])
        with torch.no_grad():
            if meta.lb:
                g0 = meta.geoms[0]
                ctx.labels = zl
                zl_v = nv.label_bias_fwd(zl, emb.detach(), blk[0].detach(), g0.K, g0.cout, g0.cin, g0.cc, aks[0].contiguous())
            else:
                zl_v = None if zl is None else zl.detach()
            h, tape = fwd_pass(meta, x, zl_v, aks, [p.detach() for p in blk], want_xa)
            v = nv.head_fwd(h, fw.detach(), None if fb is None else fb.detach()).view(-1, 1) if meta.head else None
        ctx.meta = meta
        ctx.n_a = x_a.shape[0]
        ctx.n_b = 0 if x_b is None else x_b.shape[0]
        ctx.has_zl = zl is not None and not meta.lb
        ctx.tape = tape
        ctx.save_for_backward(ak_all, *params)
        res = v if meta.head else h
        if x_b is None:
            return (res,)
        return res[:ctx.n_a], res[ctx.n_a:]

    @staticmethod
    def backward(ctx, *gs):
        meta, n_a, n_b = ctx.meta, ctx.n_a, ctx.n_b
        saved = ctx.saved_tensors
        ak_all, params = saved[0], list(saved[1:])
        blk, fw, fb, emb = _split_params(meta, params)
        aks = meta.ak_views(ak_all.detach())
        nret = 5 + len(params)
        if all(g is None for g in gs):
            return (None,) * nret
        stash = None
        if ctx.merged is not None and gs[0] is not None and "result" in ctx.merged:
            # part a's pass already ran with the promised gradient (see forward)
            if _CHECK_PROMISE and not (gs[0].is_cuda and torch.cuda.is_current_stream_capturing()):
                want = ctx.ga.view(-1, 1) if meta.head else ctx.ga.expand_as(gs[0])
                assert torch.allclose(gs[0], want, rtol=1e-6, atol=0), \
                    "DiscTrunkFn: the gradient of part a differs from the promised one"
            stash = ctx.merged.pop("result")
            gs = (None, gs[1] if len(gs) > 1 else None)
            if gs[1] is None:
                gzl_a, dak_a = stash
                gzl_full = None
                if gzl_a is not None and ctx.has_zl and ctx.needs_input_grad[3]:
                    gzl_full = gzl_a.new_zeros((n_a + n_b,) + tuple(gzl_a.shape[1:]))
                    gzl_full[:n_a] = gzl_a
                demb = gzl_a if meta.lb else None
                return (None, None, None, gzl_full, dak_a) + (None,) * meta.nparams + tuple(_extra_grads(meta, None, None, demb))
        # sample range whose output gradient arrived
        if n_b == 0 or (gs[0] is not None and gs[1] is not None):
            lo, hi = 0, n_a + n_b
            g = gs[0] if n_b == 0 else torch.cat((gs[0], gs[1]), 0)
        elif gs[0] is not None:
            lo, hi, g = 0, n_a, gs[0]
        else:
            lo, hi, g = n_a, n_a + n_b, gs[1]
        need = ctx.needs_input_grad
        need_x = [need[1] and lo < n_a, n_b > 0 and need[2] and hi > n_a]
        need_gx0 = any(need_x)
        want_params = (not ops._SKIP_PARAM_GRADS) and any(need[3:])      # zl, adjacencies, block parameters
        tape = [tuple(_sub(t, lo, hi) for t in b_) for b_ in ctx.tape]
        lb = dict(labels=ctx.labels[lo:hi], emb=emb.detach()) if meta.lb else None
        dfw = dfb = None
        if torch.is_grad_enabled():
            # create_graph=True (the gradient penalty): the data path of BWD becomes a differentiable node of its
            # own.  Parameter gradients asked for in the same call (the penalty of kinetic-gan.py:104-111 does not
            # consume them; wgan_gp.gradient_penalty switches them off) are returned as plain first-order values.
            gx0 = gzl = dak = pgr = None
            merge = (ctx.ga is not None and ctx.merged is None and n_b > 0 and lo == n_a and need_gx0 and not need[1]
                     and any(need[3:]) and all(ops._sink_of(p) is not None for p in params))
            if need_gx0:
                outs = [b_[3] for b_ in tape]
                if merge:
                    ctx.merged = dict(ga=ctx.ga, tape=ctx.tape, n_a=n_a, n_b=n_b,
                                      labels=ctx.labels if meta.lb else None)
                    gx0 = DiscTrunkBwdFn.apply((meta, ctx.merged), g, ak_all, *params, *outs)
                else:
                    gx0 = DiscTrunkBwdFn.apply(meta, g, ak_all, *params, *outs)
            if want_params:
                with torch.no_grad():
                    gd = g.detach()
                    top = nv.head_bwd(gd, fw.detach(), tape[-1][3]) if meta.head else gd
                    _, gzl, dak, pgr, _ = bwd_pass(meta, tape, top, aks, [p.detach() for p in blk], False, True,
                                                   keep=False, use_sink=False, top_masked=meta.head, lb=lb)
                    if meta.head:
                        dfw, dfb = _head_param_grads(fw.detach(), fb, tape[-1][3], gd, use_sink=False)
        else:
            with torch.no_grad():
                top = nv.head_bwd(g, fw, tape[-1][3]) if meta.head else g
                gx0, gzl, dak, pgr, _ = bwd_pass(meta, tape, top, aks, blk, need_gx0, want_params, keep=False,
                                                 top_masked=meta.head, lb=lb)
                if meta.head and want_params:
                    dfw, dfb = _head_param_grads(fw, fb, tape[-1][3], g)
        # scatter to the inputs
        gxa = gxb = None
        if gx0 is not None:
            if n_b == 0:
                gxa = gx0
            elif lo == 0 and hi == n_a + n_b:
                gxa, gxb = (gx0[:n_a] if need_x[0] else None), (gx0[n_a:] if need_x[1] else None)
            elif lo == 0:
                gxa = gx0
            else:
                gxb = gx0
        gzl_full = None
        demb = None
        if meta.lb:
            demb = gzl
        elif gzl is not None and ctx.has_zl and need[3]:
            if lo == 0 and hi == n_a + n_b:
                gzl_full = gzl
            else:
                gzl_full = gzl.new_zeros((n_a + n_b,) + tuple(gzl.shape[1:]))
                gzl_full[lo:hi] = gzl
        if stash is not None:           # (rare) part b's first-order gradient arrived together with part a's
            gzl_a, dak_a = stash
            dak = dak_a if dak is None else dak + dak_a
            if meta.lb:
                if gzl_a is not None:
                    demb = gzl_a if demb is None else demb + gzl_a
            elif gzl_a is not None and ctx.has_zl and need[3]:
                if gzl_full is None:
                    gzl_full = gzl_a.new_zeros((n_a + n_b,) + tuple(gzl_a.shape[1:]))
                gzl_full[:n_a] += gzl_a
        out = [None, gxa, gxb, gzl_full, dak]
        out += (pgr if pgr is not None else [None] * meta.nparams)
        out += _extra_grads(meta, dfw, dfb, demb)
        return tuple(out)


class DiscTrunkBwdFn(Function):
    """gx0 = BWD(g) as a differentiable function of g, the adjacencies and the weights (first derivative of the
    trunk w.r.t. its input).  Arguments: meta, g (a trunk with head: d loss / d validity (n, 1)), the packed adjacencies,
    the parameters (blocks[, fcn.weight, fcn.bias][, label_emb.weight]), the six block outputs (LeakyReLU masks: no
    gradient)."""

    @staticmethod
    def forward(ctx, meta: TrunkMeta, g, ak_all, *rest):
        ctx.set_materialize_grads(False)
        merged = None
        if isinstance(meta, tuple):
            meta, merged = meta
        nb = meta.nb
        npar = meta.nparams + _n_extra(meta)
        params, outs = list(rest[:npar]), list(rest[npar:])
        blk, fw, fb, emb = _split_params(meta, params)
        with torch.no_grad():
            blk_d = [p.detach() for p in blk]
            aks = meta.ak_views(ak_all.detach())
            if merged is not None:
                # ONE pass over part a (promised gradient; its parameter gradients go to the bucket sink, its
                # adjacency / label-bias gradients are parked for DiscTrunkFn.backward) and part b (g; gx0 and the
                # double backward's tape are part b's)
                n_a, n_b = merged["n_a"], merged["n_b"]
                top = merged["tape"][-1][3]
                if meta.head:
                    gv = torch.cat((merged["ga"].reshape(-1).to(g.dtype), g.detach().reshape(-1)))
                    g3 = nv.head_bwd(gv, fw.detach(), top)
                else:
                    g3 = nv.new_plane(*top.shape, top.device)
                    g3[:n_a].copy_(merged["ga"].expand(n_a, *top.shape[1:]))
                    g3[n_a:].copy_(g)
                lb = dict(labels=merged["labels"], emb=emb.detach()) if meta.lb else None
                gx0, gzl, dak, _, tape2 = bwd_pass(meta, merged["tape"], g3, aks, blk_d, need_gx0=True, want_params=True,
                                                   keep=True, prow=(0, n_a), krow=(n_a, n_a + n_b),
                                                   xrow=(n_a, n_a + n_b), top_masked=meta.head, lb=lb)
                if meta.head:
                    _head_param_grads(fw.detach(), fb, top[:n_a], gv[:n_a])
                merged["result"] = (gzl, dak)
            else:
                tape = [(None, None, None, o) for o in outs]
                top = nv.head_bwd(g.detach(), fw.detach(), outs[-1]) if meta.head else g
                gx0, _, _, _, tape2 = bwd_pass(meta, tape, top, aks, blk_d, need_gx0=True, want_params=False, keep=True,
                                               top_masked=meta.head)
        ctx.meta = meta
        ctx.tape2 = tape2
        ctx.gv = g.detach() if meta.head else None
        ctx.save_for_backward(ak_all, *params, *outs)
        return gx0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, h):
        meta = ctx.meta
        nb = meta.nb
        npar = meta.nparams + _n_extra(meta)
        nret = 3 + npar + nb
        if h is None:
            return (None,) * nret
        saved = ctx.saved_tensors
        ak_all, params, outs = saved[0], list(saved[1:1 + npar]), list(saved[1 + npar:])
        blk, fw, fb, emb = _split_params(meta, params)
        want = not ops._SKIP_PARAM_GRADS
        gg, dak, pgr = dbl_pass(meta, outs, ctx.tape2, h, meta.ak_views(ak_all.detach()), blk, want_params=want)
        if not meta.head:
            return (None, gg, dak) + tuple(pgr) + tuple(_extra_grads(meta)) + (None,) * nb
        # the top gradient was gv[n] w[c] / (T V): its cotangent gg reaches the Linear's weight (the penalty
        # differentiates THROUGH the backward pass, kinetic-gan.py:104-111); gv is a constant (ones)
        dfw = None
        if want and ctx.needs_input_grad[3 + meta.nparams]:
            dfw, _ = _head_param_grads(fw, None, gg, ctx.gv)
        return (None, None, dak) + tuple(pgr) + tuple(_extra_grads(meta, dfw, None, None)) + (None,) * nb
