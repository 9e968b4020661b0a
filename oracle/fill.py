"""ORACLE support (test infrastructure): deterministic parameter / input filler.

Weights are ~29 MB and are never committed; the reference (when fixtures are made), the oracle
and the HIP modules are all filled by this function, keyed by state-dict name, so that one seed
defines the whole model.  Ranges are chosen so that every st_gcn feature is exercised:
edge_importance in [0.5, 1.5], noise weights and BN affine terms non-trivial, BN running stats
non-default.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rs(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31 - 1))


def fill_tensor(name: str, t: torch.Tensor, seed: int = 0) -> torch.Tensor:
    rs = _rs(name, seed)
    shape = tuple(t.shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros_like(t)
    if "edge_importance" in name:
        a = rs.uniform(0.5, 1.5, shape)
    elif name.endswith("noise.weight"):
        a = rs.uniform(-0.3, 0.3, shape)
    elif name.endswith("running_mean"):
        a = rs.uniform(-0.2, 0.2, shape)
    elif name.endswith("running_var"):
        a = rs.uniform(0.5, 1.5, shape)
    elif "label_emb" in name:
        a = rs.normal(0, 1, shape)
    elif t.dim() == 1:
        # bias / BN affine: BN weight (".1.weight" in the generator) should sit near 1
        if name.endswith(".1.weight"):
            a = rs.uniform(0.6, 1.4, shape)
        else:
            a = rs.uniform(-0.2, 0.2, shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        if name.startswith("mlp."):
            bound = np.sqrt(3.0 / fan_in) * 1.2
        else:
            bound = np.sqrt(3.0 / fan_in) * 1.4
        a = rs.uniform(-bound, bound, shape)
    return torch.as_tensor(a, dtype=t.dtype).reshape(shape)


@torch.no_grad()
def fill_module(module: torch.nn.Module, seed: int = 0) -> None:
    sd = module.state_dict()
    for k, v in sd.items():
        v.copy_(fill_tensor(k, v, seed))


def rand_inputs(n, channels, t, v, n_classes, latent, seed=0, device="cpu"):
    """Synthetic batch as SURVEY.md 8(d): real~U(-1,1), labels~randint, z~N(0,1), alpha~U(0,1)."""
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(n, channels, t, v, generator=g) * 2 - 1
    labels = torch.randint(0, n_classes, (n,), generator=g)
    z = torch.randn(n, latent, generator=g)
    alpha = torch.rand(n, 1, 1, 1, generator=g)
    return real.to(device), labels.to(device), z.to(device), alpha.to(device)


def gen_noise_shapes(n, t_size, num_node):
    """(N,1,T,V) of the noise each generator block draws (generator.py:179), NTU or H36M."""
    v3, v2, v1, v0 = num_node[3], num_node[2], num_node[1], num_node[0]
    return [(n, 1, 1, v3), (n, 1, t_size // 16, v3), (n, 1, t_size // 16, v2), (n, 1, t_size // 8, v2),
            (n, 1, t_size // 4, v1), (n, 1, t_size // 2, v1), (n, 1, t_size, v0)]


def rand_noise(n, t_size, num_node, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(10_000 + seed)
    return [torch.randn(*s, generator=g).to(device) for s in gen_noise_shapes(n, t_size, num_node)]


def block_input(shape, seed):
    g = torch.Generator().manual_seed(777 + seed)
    return torch.randn(*shape, generator=g)


def gen_block_in_shapes(n, lat, channels, t, num_node):
    """Input (N,C,T,V) of each of the seven generator blocks."""
    v0, v1, v2, v3 = num_node
    return [(n, lat, 1, v3), (n, 512, 1, v3), (n, 256, t // 16, v3), (n, 128, t // 16, v2),
            (n, 64, t // 8, v2), (n, 32, t // 4, v1), (n, channels, t // 2, v1)]


def disc_block_in_shapes(n, cin0, latent, t, num_node):
    """Input (N,C,T,V) of each of the six discriminator blocks."""
    v0, v1, v2, v3 = num_node
    return [(n, cin0, t, v0), (n, 32, t, v1), (n, 64, t, v1), (n, 128, t // 2, v2),
            (n, 256, t // 4, v2), (n, 512, t // 8, v3)]
