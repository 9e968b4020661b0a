"""Data path: kinetic_gan_amd.feeder against the REFERENCE's Feeder outputs (tests/golden/
feeder_ref.npz, written by make_feeder_fixture.py from the imported reference on the synthetic data files next to
it) and against the oracle restatement; the device batch pipeline against DataLoader-style collation."""
import os

import numpy as np
import pytest
import torch

import kinetic_gan_amd  # noqa: F401
from kinetic_gan_amd.feeder import DeviceBatches, Feeder
from oracle import feeder_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def paths(base):
    return os.path.join(G, f"feeder_{base}_data.npy"), os.path.join(G, f"feeder_{base}_label.pkl")


@pytest.mark.parametrize("tag,base,ds,classes,norm", [("ntu", "ntu", "ntu", None, True), ("ntu_sub", "ntu", "ntu", [3, 1], True),
                                                      ("h36m", "h36m", "h36m", None, True), ("h36m_raw", "h36m", "h36m", None, False)])
def test_feeder_matches_reference_outputs(tag, base, ds, classes, norm):
    gold = np.load(os.path.join(G, "feeder_ref.npz"))
    d, l = paths(base)
    f = Feeder(d, l, classes=classes, norm=norm, dataset=ds)
    assert len(f) == int(gold[tag + "_len"])
    assert float(f.max) == float(gold[tag + "_max"]) and float(f.min) == float(gold[tag + "_min"])
    xs = np.stack([np.asarray(f[i][0]) for i in range(len(f))])
    ys = np.asarray([f[i][1] for i in range(len(f))])
    assert np.array_equal(xs, gold[tag + "_x"])            # bit-exact: same arithmetic in the same order
    assert np.array_equal(ys, gold[tag + "_y"])
    # the oracle restatement agrees with the reference as well
    data, label, mx, mn = feeder_ref.load(d, l, classes, ds)
    for i in range(len(f)):
        x, y = feeder_ref.sample(data, label, mx, mn, i, ds, norm)
        assert np.array_equal(x, gold[tag + "_x"][i]) and y == gold[tag + "_y"][i]


@pytest.mark.parametrize("base,ds,t_size,bs", [("ntu", "ntu", 16, 5), ("h36m", "h36m", 16, 4), ("ntu", "ntu", 64, 3)])
def test_device_batches_equal_collated_reference_samples(base, ds, t_size, bs):
    """every epoch: the batches are the crop-to-t_size collation (kinetic-gan.py:129-131) of the reference's samples
    for that epoch's permutation, normalised on the 'device'; ragged last batch with drop_last=False; rank sharding"""
    d, l = paths(base)
    f = Feeder(d, l, dataset=ds)
    data, label, mx, mn = feeder_ref.load(d, l, None, ds)
    for drop_last in (True, False):
        it = DeviceBatches(f, bs, t_size, "cpu", shuffle=True, drop_last=drop_last, seed=3)
        for epoch in range(2):
            idx = np.arange(len(f))
            np.random.RandomState(3 + epoch).shuffle(idx)
            nb = len(idx) // bs if drop_last else -(-len(idx) // bs)
            got = list(it)
            assert len(got) == nb == len(it)
            for b, (x, y) in enumerate(got):
                rx, ry = feeder_ref.batch(data, label, mx, mn, idx[b * bs:(b + 1) * bs], t_size, ds)
                assert x.dtype == torch.float32 and y.dtype == torch.int64
                np.testing.assert_allclose(x.numpy(), rx, rtol=0, atol=2e-6)     # x*scale+shift vs 2((x-min)/(max-min))-1
                assert np.array_equal(y.numpy(), ry)
    r0 = [y.tolist() for _, y in DeviceBatches(f, bs, t_size, "cpu", seed=5, rank=0, world=2)]
    r1 = [y.tolist() for _, y in DeviceBatches(f, bs, t_size, "cpu", seed=5, rank=1, world=2)]
    allb = [y.tolist() for _, y in DeviceBatches(f, bs, t_size, "cpu", seed=5)]
    per = len(allb) // 2
    assert r0 == allb[0:2 * per:2] and r1 == allb[1:2 * per:2]
    # ranks never disagree about the number of steps of an epoch (each step ends in collectives): n % world != 0
    for world in (2, 3, 4):
        its = [DeviceBatches(f, bs, t_size, "cpu", seed=5, rank=r, world=world) for r in range(world)]
        got = [[y.tolist() for _, y in it] for it in its]
        assert len({len(g) for g in got}) == 1 and all(len(it) == len(got[0]) for it in its), (world, [len(g) for g in got])
        assert len(got[0]) == len(allb) // world
        for r in range(world):
            assert got[r] == allb[r:len(got[0]) * world:world]


@pytest.mark.gpu
def test_device_batches_on_gpu():
    d, l = paths("ntu")
    f = Feeder(d, l, dataset="ntu")
    dev = torch.device("cuda:0")
    cpu = list(DeviceBatches(f, 4, 16, "cpu", seed=1))
    gpu = list(DeviceBatches(f, 4, 16, dev, seed=1))
    assert len(cpu) == len(gpu) == 3
    for (a, b), (c, e) in zip(cpu, gpu):
        assert c.device == dev and torch.equal(b, e.cpu())
        assert (a - c.cpu()).abs().max().item() < 2e-6
