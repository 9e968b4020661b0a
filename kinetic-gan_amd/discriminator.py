"""Discriminator and its down-sampling st_gcn block on the HIP path.

Same constructor / forward signatures, attribute names and state_dict keys as the reference's
models/discriminator.py (Discriminator :14-74, st_gcn :78-142).  What differs is HOW a block is
evaluated (results agree to fp32 round-off, tests/test_parity_gpu.py):

  reference (discriminator.py:125-136)        here
  res = residual(x)                           xa  = kg_agg_expand(x, A_eff[:, :, keep])       aggregate FIRST, on C_in
  y   = conv1x1(x)  -> 3*C_out channels       z   = kg_conv(xa; W_gcn as 3 channel-block taps) (<  C_out) planes and only
  z   = einsum(y, A)                          out = kg_conv(z; W_tcn 3 temporal taps, stride s | for the vertices the block
  u   = tcn(z) + res                                        + x[keep]; W_res + biases, LeakyReLU)  keeps; tcn/residual/act only
  u   = u[..., keep]; nearest T -> T/s; lrelu                                                   at kept (t, v)

sum_k (W_k x) A_k == sum_k W_k (x A_k), and the temporal conv is per vertex, so dropping the
vertices / frames the block throws away BEFORE computing them does not change any kept value.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._native import TAP_CHANBLOCK, TAP_TIME, WView
from .graph import build_graph
from .tgcn import ConvTemporalGraphical


def _adjacency_list(graph):
    return [torch.tensor(a, dtype=torch.float32, requires_grad=False) for a in graph.As]


class _GraphModule(nn.Module):
    """Keeps the reference's plain-list ``self.A`` (not a buffer, generator.py:47) but moves it with
    the module so ``.cuda()`` / ``.to()`` behave."""

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.A = [fn(t) for t in self.A]
        return self

    # Several forward passes between two optimiser steps see the same parameters (the WGAN-GP critic step runs D on
    # the real+fake batch and on the interpolates): inside `shared_adjacency()` the masked adjacencies
    # A[lvl] * edge_importance[i] (and the blocks' kept-column copies of them) are built once and shared by those
    # passes - one autograd sub-graph, one backward through it - instead of once per pass.
    _shared = None

    @contextlib.contextmanager
    def shared_adjacency(self):
        prev, self._shared = self._shared, {}
        try:
            yield self
        finally:
            self._shared = prev
            for blk in self.st_gcn_networks:
                blk._ak = None

    def _masked_adjacency(self, i, gcn, importance):
        if self._shared is None:
            return self.A[gcn.lvl] * importance
        a = self._shared.get(i)
        if a is None:
            a = self._shared[i] = self.A[gcn.lvl] * importance
        return a


class LabelBiasTable(torch.autograd.Function):
    """table[l, (c, w)] = sum_k sum_j W[k, c, j] E[l, j] S[k, w]: the bias the label channels of discriminator block 0
    (discriminator.py:57-60: the class embedding broadcast over (t, v) and concatenated in front of x) add to the gcn
    output of a sample of class l, with S[k, w] = sum_v A_eff[k, v, w].  The products are 60 deep with 60-352 rows: the
    small ones are broadcast products with contiguous reductions, the two with a 352-deep contraction vendor GEMMs
    (the 60-deep GEMMs over the 3n batch this replaces ran 20-40 us each).  First order only: the gradient penalty's
    double backward does not reach the bias (it shifts LeakyReLU inputs, whose second derivative is zero)."""

    @staticmethod
    def forward(ctx, Wc, E, S):
        K, C, J = Wc.shape
        WS = (Wc.unsqueeze(2) * S.view(K, 1, -1, 1)).sum(0)                      # (C, W, J)
        WS2 = WS.view(-1, J)
        ctx.save_for_backward(Wc, E, S, WS2)
        return (E.unsqueeze(1) * WS2.unsqueeze(0)).sum(-1)                        # (L, C W)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dT):
        Wc, E, S, WS2 = ctx.saved_tensors
        K, C, J = Wc.shape
        dE = dT @ WS2 if ctx.needs_input_grad[1] else None                       # (L, J)
        dWc = dS = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dWS = (dT.t() @ E).view(C, -1, J)                                     # (C, W, J)
            if ctx.needs_input_grad[0]:
                dWc = (dWS.unsqueeze(0) * S.view(K, 1, -1, 1)).sum(2)             # (K, C, J)
            if ctx.needs_input_grad[2]:
                dS = (Wc.unsqueeze(2) * dWS.unsqueeze(0)).sum((1, 3))             # (K, W)
        return dWc, dE, dS


def _joined_labels(ls):
    """The parts' label vectors as one vector: when they are consecutive slices of ONE contiguous tensor (the trainer hands
    over slices of its 3n label vector) that tensor's slice, else a concatenation."""
    a = ls[0]
    ok = a.dim() == 1 and a.is_contiguous()
    end = a.storage_offset() + a.numel()
    for b in ls[1:]:
        ok = ok and b.dim() == 1 and b.is_contiguous() and b.dtype == a.dtype and b.device == a.device and \
            b.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and b.storage_offset() == end
        if not ok:
            return torch.cat(ls, 0)
        end += b.numel()
    return torch.as_strided(a, (end - a.storage_offset(),), (1,), a.storage_offset())


class Discriminator(_GraphModule):
    def __init__(self, in_channels, n_classes, t_size, latent, edge_importance_weighting=True,
                 dataset='ntu', **kwargs):
        super().__init__()
        self.graph = build_graph(dataset)
        self.A = _adjacency_list(self.graph)

        spatial_kernel_size = [A.size(0) for A in self.A]
        temporal_kernel_size = [3 for _ in self.A]
        kernel_size = (temporal_kernel_size, spatial_kernel_size)
        self.t_size = t_size
        g = self.graph
        self.st_gcn_networks = nn.ModuleList((
            st_gcn(in_channels + n_classes, 32, kernel_size, 1, graph=g, lvl=0, dw_s=True, dw_t=t_size, residual=False, **kwargs),
            st_gcn(32, 64, kernel_size, 1, graph=g, lvl=1, dw_s=False, dw_t=t_size, **kwargs),
            st_gcn(64, 128, kernel_size, 1, graph=g, lvl=1, dw_s=True, dw_t=int(t_size / 2), **kwargs),
            st_gcn(128, 256, kernel_size, 1, graph=g, lvl=2, dw_s=False, dw_t=int(t_size / 4), **kwargs),
            st_gcn(256, 512, kernel_size, 1, graph=g, lvl=2, dw_s=True, dw_t=int(t_size / 8), **kwargs),
            st_gcn(512, latent, kernel_size, 1, graph=g, lvl=3, dw_s=False, dw_t=int(t_size / 16), **kwargs),
        ))
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList([
                nn.Parameter(torch.ones(self.A[i.lvl].size())) for i in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)
        self.label_emb = nn.Embedding(n_classes, n_classes)
        self.fcn = nn.Linear(latent, 1)
        # True: the six blocks run as one hand-scheduled autograd node (disc_trunk.py); False: block by block
        # through ops.py (same results; the block-wise path is also what st_gcn.forward offers on its own)
        self.use_trunk = os.environ.get("KG_TRUNK", "1") != "0"
        self._trunk_cache = {}

    # ---- trunk path ------------------------------------------------------------------------------------------------
    def _trunk_meta(self, T, V, device):
        key = (T, V, str(device))
        meta = self._trunk_cache.get(key)
        if meta is None:
            from .disc_trunk import BlockGeom, TrunkMeta
            geoms, t, v = [], T, V
            ok = True
            for i, blk in enumerate(self.st_gcn_networks):
                cc = self.label_emb.embedding_dim if (i == 0 and blk.res_kind == "none") else 0
                if not (blk.dw_t <= t and t % blk.dw_t == 0):
                    ok = False          # non-integer frame ratio: F.interpolate fallback of the block-wise path
                    break
                g = BlockGeom(blk, t, v, device, const_channels=cc)
                geoms.append(g)
                t, v = g.t_out, g.W
            # the node also holds the pool + Linear(latent, 1) head and block 0's label bias (disc_trunk.DiscTrunkFn)
            fused = os.environ.get("KG_TRUNK_FUSED_ENDS", "1") != "0"
            meta = TrunkMeta(geoms, [self.A[blk.lvl] for blk in self.st_gcn_networks],
                             head=fused and self.fcn.out_features == 1, label_bias=fused) if ok else False
            self._trunk_cache[key] = meta
        return meta

    def forward_parts(self, parts, promised_grad=None):
        """``parts``: one or two (x, labels) batches evaluated as ONE launch sequence with the same parameters (the
        critic step of kinetic-gan.py:143-150 runs D on real+fake and on the interpolates).  Returns the validity
        of every part.  Gradients of the parts stay independent (a backward pass only touches the samples whose
        gradient arrived).
        ``promised_grad`` (n_a,): the caller's promise that the loss gradient w.r.t. the FIRST part's validities will
        be exactly this vector (``-1/n`` / ``+1/n`` for the critic loss of kinetic-gan.py:150) - lets the trunk fold
        that part's backward pass into the gradient penalty's (disc_trunk.DiscTrunkFn)."""
        from .disc_trunk import DiscTrunkFn, MaskedAdjacencyFn
        xs = [p[0] for p in parts]
        N, C, T, V = xs[0].shape
        meta = self._trunk_meta(T, V, xs[0].device)
        if meta is False or len(parts) > 2:
            return [self._forward_blockwise(x, lab) for x, lab in parts]
        labels = parts[0][1] if len(parts) == 1 else _joined_labels([p[1] for p in parts])
        if isinstance(self.edge_importance, nn.ParameterList):
            ak_all = MaskedAdjacencyFn.apply(meta, *self.edge_importance)
        else:       # edge_importance_weighting=False: the plain adjacencies
            ak_all = meta.A_sel
        aks = meta.ak_views(ak_all)
        g0 = meta.geoms[0]
        zl = None
        if meta.lb:
            zl = labels.contiguous()        # the trunk computes the label bias itself (kg_label_bias_fwd)
        elif g0.cc:
            # label channels of block 0 (discriminator.py:57-60) folded into a per-sample bias, see st_gcn: the bias
            # depends on the sample only through its class, so it is computed for the n_classes embedding rows
            # (a (classes, C_out, W) table) and looked up per sample.  Broadcast products + sums: the vendor GEMM
            # heuristics pick 20-40 us configurations for these 60-deep, 96-row products.
            Wc = self.st_gcn_networks[0].gcn.conv.weight.view(g0.K, g0.cout, g0.cin)[:, :, :g0.cc]
            table = LabelBiasTable.apply(Wc, self.label_emb.weight, aks[0].sum(1))           # (classes, C_out * W)
            zl = torch.nn.functional.embedding(labels, table).view(labels.shape[0], g0.cout, 1, -1)
        else:
            c = self.label_emb(labels)
            xs = [torch.cat((c_.view(x.shape[0], -1, 1, 1).expand(-1, -1, T, V), x), 1)
                  for x, c_ in zip(xs, torch.split(c, [x.shape[0] for x in xs]))]
        params = []
        for blk in self.st_gcn_networks:
            params += [blk.gcn.conv.weight, blk.tcn.weight, blk.tcn.bias]
            if blk.res_kind == "conv":
                params += [blk.residual.weight, blk.residual.bias]
        w, b = self.fcn.weight, self.fcn.bias
        targ = meta
        if meta.head:
            params += [w, b]
            if promised_grad is not None and len(xs) == 2:
                targ = (meta, promised_grad.detach())       # d loss / d validity of the first part
        elif promised_grad is not None and len(xs) == 2 and w.shape[0] == 1:
            # d loss / d h_a[n, c, t, v] through mean-pool + Linear: promised[n] * w[c] / (T' V')
            last = meta.geoms[-1]
            ga = promised_grad.detach().view(-1, 1) * (w.detach().view(1, -1) / float(last.t_out * last.W))
            targ = (meta, ga.view(ga.shape[0], ga.shape[1], 1, 1))
        if meta.lb:
            params += [self.label_emb.weight]
        hs = DiscTrunkFn.apply(targ, xs[0], xs[1] if len(xs) > 1 else None, zl, ak_all, *params)
        if meta.head:
            return list(hs)
        # global average pool + Linear(latent, 1) (discriminator.py:68-72) as a matrix-VECTOR product: the GEMM
        # path picks a 16x256 tile for the single output column (18-20 us per call at bs=64 against ~5 for gemv)
        if w.shape[0] == 1:
            return [torch.addmv(b.expand(h.shape[0]), h.mean(dim=(2, 3)), w.view(-1)).unsqueeze(1) for h in hs]
        return [self.fcn(h.mean(dim=(2, 3))) for h in hs]

    def forward(self, x, labels):
        if self.use_trunk:
            return self.forward_parts([(x, labels)])[0]
        return self._forward_blockwise(x, labels)

    def _forward_blockwise(self, x, labels):
        N, C, T, V = x.size()
        c = self.label_emb(labels)
        # The reference broadcasts c to (N, n_cls, T, V) and concatenates it in front of x
        # (discriminator.py:57-60).  Those n_cls channels are constant over (t, v), so the first block takes
        # them as `const_channels` and folds them into a per-sample bias instead of materialising them.
        for i, (gcn, importance) in enumerate(zip(self.st_gcn_networks, self.edge_importance)):
            A = self._masked_adjacency(i, gcn, importance)
            if i == 0 and gcn.res_kind == "none":
                x, _ = gcn(x, A, const_channels=c)
            else:
                if i == 0:
                    x = torch.cat((c.view(N, -1, 1, 1).expand(-1, -1, T, V), x), 1)
                x, _ = gcn(x, A)
        x = x.mean(dim=(2, 3))          # global average pool (discriminator.py:68-69)
        return self.fcn(x)


class st_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, graph=None, lvl=3, dropout=0,
                 residual=True, dw_s=False, dw_t=64):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0][lvl] % 2 == 1
        if stride != 1 or kernel_size[0][lvl] != 3:
            raise NotImplementedError("HIP st_gcn: temporal kernel 3, conv stride 1 (all the reference uses)")
        padding = ((kernel_size[0][lvl] - 1) // 2, 0)
        self.graph, self.lvl, self.dw_s, self.dw_t = graph, lvl, dw_s, dw_t
        self.in_channels, self.out_channels = in_channels, out_channels
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1][lvl])
        self.tcn = nn.Conv2d(out_channels, out_channels, (kernel_size[0][lvl], 1), (stride, 1), padding)
        if not residual:
            self.res_kind = "none"
            self.residual = lambda x: 0
        elif (in_channels == out_channels) and (stride == 1):
            self.res_kind = "identity"
            self.residual = lambda x: x
        else:
            self.res_kind = "conv"
            self.residual = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1))
        self.l_relu = nn.LeakyReLU(0.2, inplace=True)
        self._cache = {}
        self._ak = None          # (A, kept-column copy, grad mode) of the last forward, see shared_adjacency

    # ---- geometry (cached per input shape / device) ----------------------------------------------------------
    def _plan(self, T, V, device):
        key = (T, V, str(device))
        p = self._cache.get(key)
        if p is not None:
            return p
        cin, cout, K = self.in_channels, self.out_channels, self.gcn.kernel_size
        if self.dw_s:
            keep_np = np.asarray(self.graph.map[self.lvl + 1][:, 1], dtype=np.int64)
            inv = np.full(V, -1, dtype=np.int32)
            inv[keep_np] = np.arange(len(keep_np), dtype=np.int32)
            keep_l = torch.as_tensor(keep_np, device=device)
            keep_i = torch.as_tensor(keep_np.astype(np.int32), device=device)
            inv_i = torch.as_tensor(inv, device=device)
            W = len(keep_np)
        else:
            keep_l = keep_i = inv_i = None
            W = V
        s = T // self.dw_t if (self.dw_t <= T and T % self.dw_t == 0) else 1
        t_out = T // s
        spec_g = ops.ConvSpec(M=cout, Cin=cin, taps=K, tap_mode=TAP_CHANBLOCK, t_stride=1, T_in=T, V_in=W,
                              T_out=T, V_out=W, wv=WView(sT=cout * cin, sO=cin, sI=1),
                              w_shape=(K * cout, cin, 1, 1))
        spec_t = ops.ConvSpec(M=cout, Cin=cout, taps=3, tap_mode=TAP_TIME, t_stride=s, T_in=T, V_in=W,
                              T_out=t_out, V_out=W, wv=WView(sT=1, sO=cout * 3, sI=3), w_shape=(cout, cout, 3, 1))
        spec_r = None
        if self.res_kind == "conv":
            spec_r = ops.ConvSpec(M=cout, Cin=cin, taps=1, tap_mode=TAP_TIME, t_stride=s, T_in=T, V_in=V,
                                  T_out=t_out, V_out=W, wv=WView(sT=0, sO=cin, sI=1), w_shape=(cout, cin, 1, 1),
                                  vmap=keep_i, inv_vmap=inv_i)
        p = dict(keep=keep_l, spec_g=spec_g, spec_t=spec_t, spec_r=spec_r, stride=s, t_out=t_out)
        self._cache[key] = p
        return p

    def _gcn_with_const_channels(self, x, Ak, const_channels, p):
        """gcn of cat(const broadcast over (t,v), x) without building the concatenation:
        z = sum_k W_k[:, Cc:] (x A_k)  +  sum_k (W_k[:, :Cc] e_n) * colsum(A_k)       (e_n = const_channels[n])."""
        N, C, T, V = x.shape
        K, cout = self.gcn.kernel_size, self.out_channels
        cc = const_channels.shape[1]
        assert cc + C == self.in_channels, (cc, C, self.in_channels)
        Wg = self.gcn.conv.weight.view(K, cout, self.in_channels)
        Wd = Wg[:, :, cc:].contiguous()
        key = ("split", T, Ak.shape[2], C)
        spec = self._cache.get(key)
        if spec is None:
            W = Ak.shape[2]
            spec = ops.ConvSpec(M=cout, Cin=C, taps=K, tap_mode=TAP_CHANBLOCK, t_stride=1, T_in=T, V_in=W,
                                T_out=T, V_out=W, wv=WView(sT=cout * C, sO=C, sI=1), w_shape=(K, cout, C))
            self._cache[key] = spec
        xa = ops.AggExpand.apply(x, Ak, 1)
        z = ops.Conv.apply(xa, Wd, None, spec)
        proj = torch.einsum("kcj,nj->nkc", Wg[:, :, :cc], const_channels)       # (N, K, Cout)
        zl = torch.einsum("nkc,kw->ncw", proj, Ak.sum(1))                         # (N, Cout, W)
        return z + zl.unsqueeze(2)

    def forward(self, x, A, const_channels=None):
        """``const_channels`` (N, Cc): channels that are constant over (t, v) and logically sit in FRONT of x
        (the discriminator's label embedding); only supported on blocks without a residual branch."""
        N, C, T, V = x.shape
        p = self._plan(T, V, x.device)
        # (the same A object again - see _GraphModule.shared_adjacency - reuses the kept-column copy)
        if self._ak is not None and self._ak[0] is A and self._ak[2] == torch.is_grad_enabled():
            Ak = self._ak[1]
        else:
            Ak = (A[:, :, p["keep"]] if self.dw_s else A).contiguous()
            self._ak = (A, Ak, torch.is_grad_enabled())
        if const_channels is not None:
            assert self.res_kind == "none"
            z = self._gcn_with_const_channels(x, Ak, const_channels, p)
        else:
            xa = ops.AggExpand.apply(x, Ak, 1)
            z = ops.Conv.apply(xa, self.gcn.conv.weight, None, p["spec_g"])
        if self.res_kind == "conv":
            xr, wr, br = x, self.residual.weight, self.residual.bias
        elif self.res_kind == "identity":
            xr, wr, br = (x[:, :, :, p["keep"]] if self.dw_s else x), None, None
        else:
            xr = wr = br = None
        out = ops.DiscTail.apply(z, xr, self.tcn.weight, self.tcn.bias, wr, br,
                                 p["spec_t"], p["spec_r"], self.res_kind)
        if out.shape[2] != self.dw_t:     # non-integer ratio: nearest resize commutes with the pointwise LeakyReLU
            out = F.interpolate(out, size=(self.dw_t, out.size(-1)))
        return out, A

    def downsample_s(self, tensor):
        keep = self.graph.map[self.lvl + 1][:, 1]
        return tensor[:, :, :, torch.as_tensor(keep, device=tensor.device)]
