"""ORACLE (test infrastructure, not product code).

Device-agnostic stock-PyTorch fp32 restatement of the reference's st_gcn hot path, written
in the reference's own dense formulation and operator order so that it is the working
definition of "the reference's result" on a box where /root/reference does not exist:

* ``ConvTemporalGraphical``  <- models/init_gan/tgcn.py:36-68
* ``GenBlock``   (= generator.st_gcn)      <- models/generator.py:112-200
* ``DiscBlock``  (= discriminator.st_gcn)  <- models/discriminator.py:80-142
* ``Generator`` / ``MappingNet`` / ``NoiseInjection`` <- models/generator.py:12-108
* ``Discriminator``                         <- models/discriminator.py:14-74
* ``gradient_penalty`` / ``d_step_losses`` / ``g_step_loss`` <- kinetic-gan.py:94-114,137-174

Differences from the reference, all result-neutral: no hard-coded ``.cuda()`` /
``device='cuda:0'`` (generator.py:47,179; discriminator.py:19); the per-block noise can be
injected (``noise=`` list) instead of drawn inside forward; the per-sample mapping-net loop
(generator.py:83-85) stacks once at the end instead of growing a tensor with O(N^2) ``cat``.

PINNING: checked against outputs of the reference itself (imported in the dev container by
tests/golden/make_fixtures.py) - see tests/test_oracle_golden.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.graph_tables import load_graph


class ConvTemporalGraphical(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1,
                 t_padding=0, t_dilation=1, bias=False):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, (t_kernel_size, 1),
                              stride=(t_stride, 1), padding=(t_padding, 0),
                              dilation=(t_dilation, 1), bias=bias)

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size          # tgcn.py:59
        y = self.conv(x)                              # tgcn.py:61
        n, kc, t, v = y.shape
        y = y.view(n, self.kernel_size, kc // self.kernel_size, t, v)   # k-major channels, tgcn.py:64
        out = torch.einsum("nkctv,kvw->nctw", y, A)   # tgcn.py:66
        return out.contiguous(), A


class NoiseInjection(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1, channel, 1, 1))     # generator.py:16

    def forward(self, image, noise):
        return image + self.weight * noise


class MappingNet(nn.Module):
    def __init__(self, latent=1024, mlp=4):
        super().__init__()
        seq = []
        for _ in range(mlp):
            lin = nn.Linear(latent, latent)
            lin.weight.data.normal_()                 # generator.py:29-30
            lin.bias.data.zero_()
            seq += [lin, nn.LeakyReLU(0.2)]
        self.mlp = nn.Sequential(*seq)

    def forward(self, x):
        return self.mlp(x)


def _nearest_time(x, t_out):
    """F.interpolate(x, size=(t_out, V)) with the default nearest mode (generator.py:172)."""
    return F.interpolate(x, size=(t_out, x.size(-1)))


class GenBlock(nn.Module):
    """generator.st_gcn (models/generator.py:110-200)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, graph=None, lvl=3,
                 dropout=0, bn=True, residual=True, up_s=False, up_t=64, tan=False):
        super().__init__()
        assert len(kernel_size) == 2 and kernel_size[0][lvl] % 2 == 1
        kt = kernel_size[0][lvl]
        self.graph, self.lvl, self.up_s, self.up_t, self.tan = graph, lvl, up_s, up_t, tan
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1][lvl])
        tcn = [nn.Conv2d(out_channels, out_channels, (kt, 1), (stride, 1), ((kt - 1) // 2, 0))]
        if bn:
            tcn.append(nn.BatchNorm2d(out_channels))
        self.tcn = nn.Sequential(*tcn)
        if not residual:
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self.residual = lambda x: x
        else:
            self.residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels))
        self.noise = NoiseInjection(out_channels)

    def upsample_s(self, x):
        # generator.py:185-200: new vertices are means of listed coarse neighbours (extra /2 at lvl 2),
        # inserted one after another at their fine index.
        cols = [x[..., c:c + 1] for c in range(x.size(-1))]
        for hood in self.graph.mapping[self.lvl]:
            nbrs = torch.stack([x[..., int(c)] for c in hood[1:]], -1)
            new = nbrs.mean(-1, keepdim=True) / (2 if self.lvl == 2 else 1)
            cols.insert(int(hood[0]), new)
        return torch.cat(cols, -1)

    def forward(self, x, A, noise=None):
        if self.up_s:
            x = self.upsample_s(x)
        x = _nearest_time(x, self.up_t)
        res = self.residual(x)
        x, A = self.gcn(x, A)
        x = self.tcn(x) + res
        if noise is None:
            noise = torch.randn(x.size(0), 1, x.size(2), x.size(3), device=x.device)
        x = self.noise(x, noise)
        return (torch.tanh(x) if self.tan else F.leaky_relu(x, 0.2)), A


class DiscBlock(nn.Module):
    """discriminator.st_gcn (models/discriminator.py:78-142)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, graph=None, lvl=3,
                 dropout=0, residual=True, dw_s=False, dw_t=64):
        super().__init__()
        assert len(kernel_size) == 2 and kernel_size[0][lvl] % 2 == 1
        kt = kernel_size[0][lvl]
        self.graph, self.lvl, self.dw_s, self.dw_t = graph, lvl, dw_s, dw_t
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1][lvl])
        self.tcn = nn.Conv2d(out_channels, out_channels, (kt, 1), (stride, 1), ((kt - 1) // 2, 0))
        if not residual:
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self.residual = lambda x: x
        else:
            self.residual = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1))

    def forward(self, x, A):
        res = self.residual(x)
        x, A = self.gcn(x, A)
        x = self.tcn(x) + res
        if self.dw_s:
            keep = torch.as_tensor(self.graph.map[self.lvl + 1][:, 1], device=x.device)
            x = x[:, :, :, keep]                      # discriminator.py:139-142
        x = _nearest_time(x, self.dw_t)               # discriminator.py:134
        return F.leaky_relu(x, 0.2), A


def _adjacency(graph, device=None):
    return [torch.tensor(a, dtype=torch.float32, device=device) for a in graph.As]


class Generator(nn.Module):
    def __init__(self, in_channels, out_channels, n_classes, t_size, mlp_dim=4,
                 edge_importance_weighting=True, dataset="ntu", **kwargs):
        super().__init__()
        self.graph = load_graph(dataset)      # the reference's own tables (tests/golden/graph_tables.json)
        self.A = _adjacency(self.graph)
        ks = ([3 for _ in self.A], [a.size(0) for a in self.A])
        self.t_size = t_size
        g = self.graph
        lat = in_channels + n_classes
        self.mlp = MappingNet(lat, mlp_dim)
        self.st_gcn_networks = nn.ModuleList((          # generator.py:56-64
            GenBlock(lat, 512, ks, 1, graph=g, lvl=3, bn=False, residual=False, up_s=False, up_t=1, **kwargs),
            GenBlock(512, 256, ks, 1, graph=g, lvl=3, up_s=False, up_t=int(t_size / 16), **kwargs),
            GenBlock(256, 128, ks, 1, graph=g, lvl=2, bn=False, up_s=True, up_t=int(t_size / 16), **kwargs),
            GenBlock(128, 64, ks, 1, graph=g, lvl=2, up_s=False, up_t=int(t_size / 8), **kwargs),
            GenBlock(64, 32, ks, 1, graph=g, lvl=1, bn=False, up_s=True, up_t=int(t_size / 4), **kwargs),
            GenBlock(32, out_channels, ks, 1, graph=g, lvl=1, up_s=False, up_t=int(t_size / 2), **kwargs),
            GenBlock(out_channels, out_channels, ks, 1, graph=g, lvl=0, bn=False, up_s=True,
                     up_t=t_size, tan=True, **kwargs)))
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList(
                [nn.Parameter(torch.ones(self.A[b.lvl].size())) for b in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)
        self.label_emb = nn.Embedding(n_classes, n_classes)

    def forward(self, x, labels, trunc=None, noise=None):
        c = self.label_emb(labels)
        x = torch.cat((c, x), -1)                       # generator.py:80-81
        w = torch.stack([self.mlp(row) for row in x], 0)  # per-sample loop as generator.py:83-85
        if trunc is not None:
            w = self.truncate(w, 1000, trunc)
        x = w.view(*w.shape, 1, 1)
        for i, (blk, imp) in enumerate(zip(self.st_gcn_networks, self.edge_importance)):
            A = self.A[blk.lvl].to(x.device) * imp
            x, _ = blk(x, A, None if noise is None else noise[i])
        return x

    def truncate(self, w, mean, truncation, t=None):   # generator.py:97-108
        if t is None:
            t = torch.as_tensor(np.random.normal(0, 1, (mean, *w.shape[1:])), dtype=w.dtype, device=w.device)
        # per-sample like the reference's loop (a batched GEMM rounds differently and the difference is visible
        # at 1e-5 after seven blocks)
        m = torch.stack([self.mlp(i) for i in t]).mean(0, keepdim=True)
        return m + truncation * (w - m)


class Discriminator(nn.Module):
    def __init__(self, in_channels, n_classes, t_size, latent, edge_importance_weighting=True,
                 dataset="ntu", **kwargs):
        super().__init__()
        self.graph = load_graph(dataset)      # the reference's own tables (tests/golden/graph_tables.json)
        self.A = _adjacency(self.graph)
        ks = ([3 for _ in self.A], [a.size(0) for a in self.A])
        self.t_size = t_size
        g = self.graph
        self.st_gcn_networks = nn.ModuleList((          # discriminator.py:28-35
            DiscBlock(in_channels + n_classes, 32, ks, 1, graph=g, lvl=0, dw_s=True, dw_t=t_size, residual=False, **kwargs),
            DiscBlock(32, 64, ks, 1, graph=g, lvl=1, dw_s=False, dw_t=t_size, **kwargs),
            DiscBlock(64, 128, ks, 1, graph=g, lvl=1, dw_s=True, dw_t=int(t_size / 2), **kwargs),
            DiscBlock(128, 256, ks, 1, graph=g, lvl=2, dw_s=False, dw_t=int(t_size / 4), **kwargs),
            DiscBlock(256, 512, ks, 1, graph=g, lvl=2, dw_s=True, dw_t=int(t_size / 8), **kwargs),
            DiscBlock(512, latent, ks, 1, graph=g, lvl=3, dw_s=False, dw_t=int(t_size / 16), **kwargs)))
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList(
                [nn.Parameter(torch.ones(self.A[b.lvl].size())) for b in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)
        self.label_emb = nn.Embedding(n_classes, n_classes)
        self.fcn = nn.Linear(latent, 1)

    def forward(self, x, labels):
        n, _, t, v = x.shape
        c = self.label_emb(labels)
        c = c.view(n, -1, 1, 1).repeat(1, 1, t, v)      # discriminator.py:57-58
        x = torch.cat((c, x), 1)
        for blk, imp in zip(self.st_gcn_networks, self.edge_importance):
            x, _ = blk(x, self.A[blk.lvl].to(x.device) * imp)
        x = F.avg_pool2d(x, x.shape[2:]).view(n, -1)    # discriminator.py:68-69
        return self.fcn(x)


# ---- WGAN-GP step (row T) -----------------------------------------------------------------

def gradient_penalty(D, real, fake, labels, alpha):
    """kinetic-gan.py:94-114 with alpha injected."""
    inter = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    d_inter = D(inter, labels)
    ones = torch.ones(real.shape[0], 1, device=real.device)
    grads = torch.autograd.grad(outputs=d_inter, inputs=inter, grad_outputs=ones,
                                create_graph=True, retain_graph=True, only_inputs=True)[0]
    grads = grads.reshape(grads.size(0), -1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def d_step_losses(G, D, real, labels, z, alpha, noise=None, lambda_gp=10):
    """kinetic-gan.py:137-152.  Returns dict of the step's scalars/vectors; d_loss carries the graph."""
    fake = G(z, labels, noise=noise)
    real_v = D(real, labels)
    fake_v = D(fake, labels)
    gp = gradient_penalty(D, real.detach(), fake.detach(), labels, alpha)
    d_loss = -real_v.mean() + fake_v.mean() + lambda_gp * gp
    return {"fake": fake, "real_validity": real_v, "fake_validity": fake_v,
            "gradient_penalty": gp, "d_loss": d_loss}


def g_step_loss(G, D, labels, z, noise=None):
    """kinetic-gan.py:167-171."""
    fake = G(z, labels, noise=noise)
    fake_v = D(fake, labels)
    return {"fake": fake, "fake_validity": fake_v, "g_loss": -fake_v.mean()}
