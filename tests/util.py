"""Shared helpers for the test-suite (tests only)."""
import contextlib

import torch

import kinetic_gan_amd  # noqa: F401
from kinetic_gan_amd import _native
from kinetic_gan_amd.discriminator import Discriminator
from kinetic_gan_amd.generator import Generator
from oracle import modules_ref as M
from oracle import prim_ref
from oracle.fill import fill_module

CFG = {"ntu": dict(channels=3, n_classes=60, t_size=64, latent=512, mlp=4),
       "h36m": dict(channels=2, n_classes=10, t_size=32, latent=512, mlp=4),
       "ntu120": dict(channels=3, n_classes=120, t_size=64, latent=512, mlp=8),
       "stress": dict(channels=3, n_classes=60, t_size=256, latent=512, mlp=4)}      # C5b: full G / D at t_size = 256


def ds_name(cfg_name):
    return "h36m" if cfg_name == "h36m" else "ntu"


@contextlib.contextmanager
def emulated_native():
    """Run the host logic on CPU: the native entry points are replaced by oracle/prim_ref.py."""
    restore = prim_ref.install(_native)
    try:
        yield
    finally:
        restore()


def build_pair(cfg_name, device="cpu", seed_g=1, seed_d=2):
    """(G, D) on the HIP path and (Go, Do) oracle modules with identical parameters."""
    c = CFG[cfg_name]
    ds = ds_name(cfg_name)
    G = Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=ds)
    D = Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=ds)
    Go = M.Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=ds)
    Do = M.Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=ds)
    for m, s in ((G, seed_g), (Go, seed_g), (D, seed_d), (Do, seed_d)):
        fill_module(m, seed=s)
    return c, G.to(device), D.to(device), Go, Do


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2_rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def grad_close(a, b, tol, floor=1e-6):
    """Relative-L2 agreement.  The only escape is for gradients that are analytically zero (a conv bias in front of
    a train-mode BatchNorm): there both sides are pure round-off, accepted when the difference stays below
    `floor` (absolute, 1e-6) AND below 1e-3 of the reference's own magnitude scale, whichever is larger - a tensor
    with real content can no longer hide behind an absolute floor."""
    if l2_rel(a, b) < tol:
        return True
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() < max(floor, 1e-3 * b.abs().max().item())


def grad_sample(t, k=64):
    """the strided 64-element sample of a gradient that tests/golden/make_fixtures.py stores (Dgs_* / Ggs_*)"""
    f = t.detach().reshape(-1)
    return f[:: max(1, f.numel() // k)][:k]
