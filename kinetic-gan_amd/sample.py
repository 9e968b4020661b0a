"""Inference path of the reference's sampling script on the HIP generator (row N2 of SURVEY.md 8f).

``sample_actions`` is the loop of generate.py:85-105 without its file I/O: per round it draws ``qtd`` latents for
every class still short of ``gen_qtd`` samples, applies Z-space truncation (generate.py:14-21) or passes the W-space
truncation factor to ``Generator.forward`` (generator.py:86,97-108), and collects skeleton sequences, labels and
latents.  The generator runs in eval mode under ``torch.no_grad()``: BatchNorm uses its running statistics, folded
into the tcn / residual conv weights (generator.st_gcn) so that no statistics launch remains.
"""
from __future__ import annotations

from collections import Counter
from typing import Optional, Sequence

import numpy as np
import torch

from .generator import truncate_z


@torch.no_grad()
def sample_actions(G, n_classes: int, latent_dim: int, gen_qtd: int, qtd: int = 25, label: int = -1,
                   trunc: Optional[float] = None, trunc_mode: str = "-", mean_size: int = 1000,
                   stochastic_z: Optional[torch.Tensor] = None, keep_on_device: bool = False):
    """Returns (imgs (M, C, T, V), labels (M,), z (M, latent)) with at least ``gen_qtd`` samples of every requested
    class (all classes for ``label == -1``), in the order generate.py produces them.  ``trunc_mode``: 'z', 'w' or
    '-' (generate.py:38-41); ``stochastic_z``: one fixed latent point for every sample (generate.py:80-83)."""
    was_training = G.training
    G.eval()
    dev = next(G.parameters()).device
    classes = list(range(n_classes)) if label == -1 else [label]
    imgs, labs, zs = [], [], []
    count = Counter()
    try:
        while classes:
            n = qtd * len(classes)
            if stochastic_z is not None:
                z = stochastic_z.to(dev).reshape(1, latent_dim).repeat(n, 1)
            else:
                z = torch.as_tensor(np.random.normal(0, 1, (n, latent_dim)), dtype=torch.float32, device=dev)
            if trunc_mode == "z":
                z = truncate_z(z, mean_size, trunc)
            labels_np = np.array([num for _ in range(qtd) for num in classes])
            labels = torch.as_tensor(labels_np, dtype=torch.long, device=dev)
            out = G(z, labels, trunc) if trunc_mode == "w" else G(z, labels)
            imgs.append(out if keep_on_device else out.cpu())
            zs.append(z if keep_on_device else z.cpu())
            labs.append(labels_np)
            count.update(labels_np.tolist())
            classes = [c for c in classes if count[c] < gen_qtd]
    finally:
        G.train(was_training)
    return torch.cat(imgs, 0), np.concatenate(labs, 0), torch.cat(zs, 0)
