"""Generator, mapping network and the up-sampling st_gcn block on the HIP path.

Same constructor / forward signatures, attribute names and state_dict keys as the reference's
models/generator.py (NoiseInjection :12-19, Mapping_Net :22-37, Generator :40-108, st_gcn
:110-200).  Per block:

  reference (generator.py:168-182)               here
  x = upsample_s(x)   python cat loops           x_up = kg_agg_expand(x, U, rep)     U = fixed (V_c x V_f) matrix of
  x = F.interpolate(x, (up_t, V))                                                     upsample_s, rep = up_t / T (nearest)
  res = BN(conv1x1(x))                           r    = kg_conv(x_up; W_res) + b
  y = conv1x1(x) ; z = einsum(y, A)              z    = kg_agg_reduce(kg_conv(x_up; W_gcn), A_eff)
  u = [BN](tcn(z)) + res                         u    = kg_conv(z; W_tcn 3 taps) + b
  x = act(u + w_noise * randn)                   out  = kg_affine_act: BN(u) + BN(r) + noise, LeakyReLU / tanh in one pass
                                                        (batch statistics from kg_rowsum)
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._native import ACT_LRELU, ACT_TANH, TAP_TIME, WView
from .discriminator import _GraphModule, _adjacency_list
from .graph import build_graph
from .tgcn import ConvTemporalGraphical


def truncate_z(latent, mean_size, truncation, t=None):
    """Truncation trick on Z (generate.py:14-21): pull every latent towards the mean of ``mean_size`` fresh normal
    draws.  ``t`` lets callers pin those draws; like the reference the result replaces ``latent`` row by row, here
    without the Python loop."""
    if t is None:
        t = torch.as_tensor(np.random.normal(0, 1, (mean_size, *latent.shape[1:])), dtype=latent.dtype,
                            device=latent.device)
    m = t.mean(0, keepdim=True)
    return m + truncation * (latent - m)


class NoiseInjection(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1, channel, 1, 1))

    def forward(self, image, noise):
        return image + self.weight * noise


class Mapping_Net(nn.Module):
    def __init__(self, latent=1024, mlp=4):
        super().__init__()
        layers = []
        for _ in range(mlp):
            linear = nn.Linear(latent, latent)
            linear.weight.data.normal_()
            linear.bias.data.zero_()
            layers.append(linear)
            layers.append(nn.LeakyReLU(0.2))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        for layer in self.mlp:          # same modules, same order; the Linears add their gradients into the bucket
            x = ops.SinkLinear.apply(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(x)
        return x


class Generator(_GraphModule):
    def __init__(self, in_channels, out_channels, n_classes, t_size, mlp_dim=4,
                 edge_importance_weighting=True, dataset='ntu', **kwargs):
        super().__init__()
        self.graph = build_graph(dataset)
        self.A = _adjacency_list(self.graph)

        spatial_kernel_size = [A.size(0) for A in self.A]
        temporal_kernel_size = [3 for _ in self.A]
        kernel_size = (temporal_kernel_size, spatial_kernel_size)
        self.t_size = t_size
        g = self.graph
        self.mlp = Mapping_Net(in_channels + n_classes, mlp_dim)
        self.st_gcn_networks = nn.ModuleList((
            st_gcn(in_channels + n_classes, 512, kernel_size, 1, graph=g, lvl=3, bn=False, residual=False, up_s=False, up_t=1, **kwargs),
            st_gcn(512, 256, kernel_size, 1, graph=g, lvl=3, up_s=False, up_t=int(t_size / 16), **kwargs),
            st_gcn(256, 128, kernel_size, 1, graph=g, lvl=2, bn=False, up_s=True, up_t=int(t_size / 16), **kwargs),
            st_gcn(128, 64, kernel_size, 1, graph=g, lvl=2, up_s=False, up_t=int(t_size / 8), **kwargs),
            st_gcn(64, 32, kernel_size, 1, graph=g, lvl=1, bn=False, up_s=True, up_t=int(t_size / 4), **kwargs),
            st_gcn(32, out_channels, kernel_size, 1, graph=g, lvl=1, up_s=False, up_t=int(t_size / 2), **kwargs),
            st_gcn(out_channels, out_channels, kernel_size, 1, graph=g, lvl=0, bn=False, up_s=True, up_t=t_size, tan=True, **kwargs),
        ))
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList([
                nn.Parameter(torch.ones(self.A[i.lvl].size())) for i in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)
        self.label_emb = nn.Embedding(n_classes, n_classes)
        self._adj_pack = None
        # True: the seven blocks run as one hand-scheduled autograd node, contract-first on each block's input grid
        # (gen_trunk.py) whenever its preconditions hold (training mode, flat-bucket gradient sinks); False: block by
        # block through ops.py (same results; also what st_gcn.forward offers on its own and the inference path uses)
        self.use_trunk = os.environ.get("KG_GEN_TRUNK", "1") != "0"
        self.map_kernels = os.environ.get("KG_MAP_KERNELS", "1") != "0"      # 0: embedding + mapping network on stock ops
        # exact data-parallel BatchNorm (optional): training-mode statistics over the GLOBAL batch - one
        # all-reduce of 2 C floats per BatchNorm layer and direction (ops.SyncBatchNorm2dFn).  The blocks then run one
        # by one (no trunk, no paired synthesis).  Default: per-rank statistics, like DDP without SyncBatchNorm.
        self.exact_bn = os.environ.get("KG_EXACT_BN", "0") == "1"
        self.bn_group = None             # process group of the exact mode (None: the default group)
        self._trunk = None

    def forward(self, x, labels, trunc=None, noise=None):
        """``noise``: optional list of 7 (N,1,T,V) tensors replacing the in-forward torch.randn
        (generator.py:179) - parity tests inject it."""
        return self.synthesis(self.mapping(x, labels, trunc), noise)

    def mapping(self, x, labels, trunc=None):
        """Label embedding + mapping network (+ W-space truncation): generator.py:80-87.  Deterministic in (x, labels)
        and the parameters - no noise, no BatchNorm - so one result serves every synthesis from the same latents
        until the parameters change (the WGAN-GP iteration runs G twice on the same z, kinetic-gan.py:143,167)."""
        layers = list(self.mlp.mlp)
        on_dev = (x.is_cuda and self.label_emb.weight.is_cuda) or ops.emulated()      # (CPU tensors: the stock ops below)
        fused = self.map_kernels and on_dev and x.dim() == 2 and x.dtype == torch.float32 and labels.dtype == torch.int64 and \
            all(isinstance(m, nn.Linear if i % 2 == 0 else nn.LeakyReLU) for i, m in enumerate(layers)) and \
            len({m.negative_slope for m in layers[1::2]}) == 1
        if fused:
            # whole batch at once (the reference loops per sample, generator.py:83-85); embedding + cat + the mlp's
            # Linear / LeakyReLU pairs as one autograd node over kg_linear_* (ops.MappingFn)
            wb = [p for m in layers[0::2] for p in (m.weight, m.bias)]
            w = ops.MappingFn.apply(x.contiguous(), labels.contiguous(), self.label_emb.weight, float(layers[1].negative_slope), *wb)
        else:
            c = self.label_emb(labels)
            x = torch.cat((c, x), -1)
            w = self.mlp(x)
        return self.truncate(w, 1000, trunc) if trunc is not None else w

    def synthesis_pair(self, w, noise_a=None, noise_b=None):
        """Two syntheses from the same mapped latents in ONE pass over a 2n batch: (G(w; noise_a) without autograd
        graph, G(w; noise_b) with it).  The WGAN-GP iteration draws a critic sample and, with unchanged parameters, a
        generator-step sample from the same z (kinetic-gan.py:143,167); per-sample arithmetic is batch independent
        and BatchNorm statistics / running-statistics updates are taken per half in that order, so both halves equal
        two separate forward passes while every launch is issued once."""
        n = w.shape[0]
        if (noise_a is None) != (noise_b is None):
            raise ValueError("synthesis_pair: give both noise lists or neither")
        noise = None if noise_a is None else [torch.cat((a, b), 0) for a, b in zip(noise_a, noise_b)]
        wd = w.detach()
        out, out_b = self.synthesis(torch.cat((wd, wd), 0), noise, w_b=w)
        return out[:n], out_b

    def synthesis(self, w, noise=None, w_b=None):
        """The seven st_gcn blocks on the mapped latents (generator.py:89-95).  ``w_b``: ``w`` holds two batches
        without history, ``w_b`` is the second one with it (``synthesis_pair``); returns (both results, the second
        one with history) then."""
        trunk = self._trunk_state(w)
        x = w.view((*w.shape, 1, 1))
        x_b = w_b.view((*w_b.shape, 1, 1)) if w_b is not None else None
        if noise is None:
            # the seven per-block noise planes of generator.py:179 from ONE randn launch (i.i.d. either way)
            n = x.shape[0]
            shapes = [(n, 1, gcn.up_t, self.graph.num_node[gcn.lvl]) for gcn in self.st_gcn_networks]
            sizes = [a * b * c * d for a, b, c, d in shapes]
            buf = torch.randn(sum(sizes), device=x.device, dtype=x.dtype)
            noise, off = [], 0
            for shp, sz in zip(shapes, sizes):
                noise.append(buf[off:off + sz].view(shp))
                off += sz
        if trunk is not None:
            from .gen_trunk import GenTrunkFn
            meta, params, bns = trunk
            cfg = (meta, bns, 2 if w_b is not None else 1)
            return GenTrunkFn.apply(cfg, w, w_b, *noise, *self.edge_importance, *params)
        if isinstance(self.edge_importance, nn.ParameterList) and x.is_cuda or getattr(self, "_pack_always", False):
            # A[lvl] * importance of all seven blocks in one launch (backward: one launch + one add into the bucket)
            from .disc_trunk import AdjacencyPack, MaskedAdjacencyFn
            key = str(x.device)
            if self._adj_pack is None or self._adj_pack[0] != key:
                self._adj_pack = (key, AdjacencyPack([self.A[g.lvl] for g in self.st_gcn_networks]))
            pack = self._adj_pack[1]
            adjs = MaskedAdjacencyFn.apply(pack, *self.edge_importance)
            packed = True
        else:
            packed = False
            adjs = [self.A[gcn.lvl] * importance for gcn, importance in zip(self.st_gcn_networks, self.edge_importance)]
        sync = self.bn_group if self._exact_bn_active() else False
        for i, gcn in enumerate(self.st_gcn_networks):
            gcn.sync_bn = sync               # False, or the process group (None = default) of the exact BatchNorm mode
            gcn.gcn.lazy_outer = packed      # the pack's backward computes all blocks' adjacency gradients at once
            if x_b is not None:
                (x, x_b), _ = gcn(x, adjs[i], noise[i], x_b)
            else:
                x, _ = gcn(x, adjs[i], noise[i])
        return x if w_b is None else (x, x_b)

    def _trunk_state(self, w):
        """(meta, params, bns) when this synthesis can take the hand-scheduled trunk (gen_trunk.py), else None:
        training mode (batch statistics; the inference path folds BatchNorm into the convs instead), learnable
        edge_importance, integer frame ratios, and - when a gradient will be asked for - a flat-bucket sink for every
        parameter (wgan_gp.FlatParams): the trunk hands ALL parameter gradients to the kernels' accumulate-into-bucket
        launches."""
        if not (self.use_trunk and self.training and isinstance(self.edge_importance, nn.ParameterList)):
            return None
        if self._exact_bn_active():
            return None
        if not (w.is_cuda or getattr(self, "_pack_always", False)):
            return None
        from . import gen_trunk as gt
        key = str(w.device)
        if self._trunk is None or self._trunk[0] != key:
            meta = gt.GenTrunkMeta(self, w.device)
            params, bns = gt.collect_params(self)
            self._trunk = (key, meta, params, bns, meta.ok and gt.trunk_supported(self, bns))
        _, meta, params, bns, ok = self._trunk
        if not ok:
            return None
        if torch.is_grad_enabled() and any(p.requires_grad for p in params) and not gt.all_sinks_registered(self):
            return None
        return meta, params, bns

    def _exact_bn_active(self) -> bool:
        if not (self.exact_bn and self.training):
            return False
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            if not getattr(self, "_exact_bn_warned", False):       # (round-4 ADVICE: the mode was silently off)
                import warnings
                warnings.warn("Generator.exact_bn / KG_EXACT_BN=1 needs an initialised torch.distributed process group; "
                              "BatchNorm statistics stay per process")
                self._exact_bn_warned = True
            return False
        return dist.get_world_size(self.bn_group) > 1

    def truncate(self, w, mean, truncation, t=None):
        """Truncation trick on W (generator.py:97-108); ``t`` lets callers pin the mean_size latent draws."""
        if t is None:
            t = torch.as_tensor(np.random.normal(0, 1, (mean, *w.shape[1:])), dtype=w.dtype, device=w.device)
        m = self.mlp(t).mean(0, keepdim=True)
        return m + truncation * (w - m)


class st_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, graph=None, lvl=3, dropout=0,
                 bn=True, residual=True, up_s=False, up_t=64, tan=False):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0][lvl] % 2 == 1
        if stride != 1 or kernel_size[0][lvl] != 3:
            raise NotImplementedError("HIP st_gcn: temporal kernel 3, conv stride 1 (all the reference uses)")
        padding = ((kernel_size[0][lvl] - 1) // 2, 0)
        self.graph, self.lvl, self.up_s, self.up_t, self.tan = graph, lvl, up_s, up_t, tan
        self.in_channels, self.out_channels = in_channels, out_channels
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1][lvl])
        if graph is not None:
            a_lvl = np.asarray(graph.As[lvl])
            self.gcn.single_partition = bool(a_lvl.shape[1] == 1 and a_lvl.shape[0] == 3 and not a_lvl[1:].any())
        tcn = [nn.Conv2d(out_channels, out_channels, (kernel_size[0][lvl], 1), (stride, 1), padding)]
        if bn:
            tcn.append(nn.BatchNorm2d(out_channels))
        self.tcn = nn.Sequential(*tcn)
        if not residual:
            self.res_kind = "none"
            self.residual = lambda x: 0
        elif (in_channels == out_channels) and (stride == 1):
            self.res_kind = "identity"
            self.residual = lambda x: x
        else:
            self.res_kind = "conv"
            self.residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels))
        self.noise = NoiseInjection(out_channels)
        self.l_relu = nn.LeakyReLU(0.2, inplace=True)
        self.tanh = nn.Tanh()
        self._cache = {}

    def _plan(self, T, V, device):
        key = (T, V, str(device))
        p = self._cache.get(key)
        if p is not None:
            return p
        cin, cout = self.in_channels, self.out_channels
        if self.up_s:
            U = torch.as_tensor(self.graph.upsample_matrix(self.lvl), dtype=torch.float32, device=device)[None]
        else:
            U = torch.eye(V, dtype=torch.float32, device=device)[None]
        Vf = U.shape[2]
        rep = self.up_t // T if (self.up_t >= T and self.up_t % T == 0) else None
        Tu = self.up_t
        spec_t = ops.ConvSpec(M=cout, Cin=cout, taps=3, tap_mode=TAP_TIME, t_stride=1, T_in=Tu, V_in=Vf,
                              T_out=Tu, V_out=Vf, wv=WView(sT=1, sO=cout * 3, sI=3), w_shape=(cout, cout, 3, 1))
        spec_r = ops.ConvSpec(M=cout, Cin=cin, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=Tu, V_in=Vf,
                              T_out=Tu, V_out=Vf, wv=WView(sT=0, sO=cin, sI=1), w_shape=(cout, cin, 1, 1))
        p = dict(U=U.contiguous(), rep=rep, spec_t=spec_t, spec_r=spec_r)
        self._cache[key] = p
        return p

    def _folded(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        """Inference only (eval mode, no autograd): BatchNorm with running statistics is a per-channel affine map and
        folds into the conv in front of it, W' = W * s[m], b' = b * s + (beta - mean * s), s = gamma / sqrt(var + eps)
        (generator.py:142,160 followed by generate.py:67 ``eval()``).
        Recomputed on EVERY call (a handful of per-channel launches): on the HIP path parameters and running statistics
        are rewritten through raw pointers - kg_adam_step on the flat buffer, kg_bn_fwd(_many), hipGraph replays that
        run no Python at all - so no tensor version counter or host-side generation count can tell when a cached
        fold has gone stale (round-2 ADVICE: a cache keyed on ``_version`` served the previous generator's weights
        after train -> sample -> train -> sample)."""
        with torch.no_grad():
            s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            w = (conv.weight * s.view(-1, 1, 1, 1)).contiguous()
            b = conv.bias * s + (bn.bias - bn.running_mean * s)
        return w, b

    @staticmethod
    def _bn_state(bn: nn.BatchNorm2d, training: bool):
        use_batch = training or bn.running_mean is None
        return (bn.running_mean, bn.running_var, bn.num_batches_tracked, use_batch, bn.momentum, bn.eps)

    def forward(self, x, A, noise=None, x_b=None):
        """``x_b`` given: ``x`` holds TWO independent batches back to back without autograd history (BatchNorm
        statistics stay separate, ops.GenTail) and ``x_b`` is the second one with history; every launch of the forward
        pass covers both, the backward pass only the second (ops.pair_apply).  Returns ((out, out_b), A) then."""
        N, C, T, V = x.shape
        p = self._plan(T, V, x.device)
        pair = x_b is not None

        def run(F, xf, xb, *rest):
            if pair:
                return ops.pair_apply(F, xf, xb, *rest)
            return F.apply(xf, *rest), None

        if p["rep"] is None:      # non-integer time ratio: fall back to torch's nearest resize first
            x = torch.nn.functional.interpolate(x, size=(self.up_t, V))
            if pair:
                x_b = torch.nn.functional.interpolate(x_b, size=(self.up_t, V))
            rep = 1
        else:
            rep = p["rep"]
        if self.up_s or rep > 1:
            x, x_b = run(ops.AggExpand, x, x_b, p["U"], rep)
        if pair:
            (y, y_b), _ = self.gcn(x, A, x_b)
        else:
            (y, _), y_b = self.gcn(x, A), None
        conv_t = self.tcn[0]
        # inference (eval mode, no autograd, running statistics present): BatchNorm folded into the conv weights
        fold = (not self.training) and (not torch.is_grad_enabled()) and not pair
        bn_t = gt = bt = None
        if len(self.tcn) > 1 and fold and self.tcn[1].running_mean is not None:
            wf, bf = self._folded(conv_t, self.tcn[1])
            u, u_b = ops.Conv.apply(y, wf, bf, p["spec_t"]), None
        else:
            u, u_b = run(ops.Conv, y, y_b, conv_t.weight, conv_t.bias, p["spec_t"])
            if len(self.tcn) > 1:
                b = self.tcn[1]
                gt, bt, bn_t = b.weight, b.bias, self._bn_state(b, self.training)
        r = r_b = gr = br = bn_r = None
        if self.res_kind == "conv":
            cr, b = self.residual[0], self.residual[1]
            if fold and b.running_mean is not None:
                wf, bf = self._folded(cr, b)
                r = ops.Conv.apply(x, wf, bf, p["spec_r"])
            else:
                r, r_b = run(ops.Conv, x, x_b, cr.weight, cr.bias, p["spec_r"])
                gr, br, bn_r = b.weight, b.bias, self._bn_state(b, self.training)
        elif self.res_kind == "identity":
            r, r_b = x, x_b
        if noise is None:
            noise = torch.randn(N, 1, u.shape[2], u.shape[3], device=x.device)
        act = ACT_TANH if self.tan else ACT_LRELU
        sync = getattr(self, "sync_bn", False)
        if sync is not False and self.training and not pair and (bn_t is not None or bn_r is not None):
            # exact data-parallel mode: global-batch statistics (ops.SyncBatchNorm2dFn), then the reference's own
            # arithmetic order: tcn(+BN) + residual(+BN) + noise, activation (generator.py:176-182)
            if bn_t is not None:
                u = ops.SyncBatchNorm2dFn.apply(u, gt, bt, bn_t[0], bn_t[1], bn_t[2], bn_t[4], bn_t[5], sync)
            rr = r
            if bn_r is not None:
                rr = ops.SyncBatchNorm2dFn.apply(r, gr, br, bn_r[0], bn_r[1], bn_r[2], bn_r[4], bn_r[5], sync)
            v = u + self.noise.weight * noise if rr is None else u + rr + self.noise.weight * noise
            out = torch.tanh(v) if self.tan else torch.nn.functional.leaky_relu(v, 0.2)
        elif pair:
            out = ops.GenTail.apply(u, r, noise, self.noise.weight, gt, bt, gr, br, bn_t, bn_r, act, 2, u_b, r_b)
        else:
            out = ops.GenTail.apply(u, r, noise, self.noise.weight, gt, bt, gr, br, bn_t, bn_r, act)
        return out, A

    def upsample_s(self, tensor):
        """Spatial up-sampling as one matrix product with U (same result as generator.py:185-200)."""
        U = torch.as_tensor(self.graph.upsample_matrix(self.lvl), dtype=tensor.dtype, device=tensor.device)
        return tensor @ U
