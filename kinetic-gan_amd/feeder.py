"""Data path of the training script (row N4 of SURVEY.md 8f): the reference's ``Feeder`` and a device-side batch
pipeline on top of it.

``Feeder`` keeps the reference's surface and arithmetic (feeder/feeder.py:21-80): ``.npy`` skeleton data memory-mapped
as (N, C, T, V, M) (NTU; first person only) or (N, C, T, V) (Human3.6M), labels from a pickle ``(sample_name, label)``,
optional class subset, global min / max, samples scaled to [-1, 1] - so ``torch.utils.data.DataLoader(Feeder(...))``
works exactly as in kinetic-gan.py:68-74.

``DeviceBatches`` is the MI355X-side replacement for DataLoader workers + ``imgs.type(Tensor)`` (kinetic-gan.py:
129-131): a batch is gathered from the memory map straight into a PINNED host buffer - only the first person and the
first ``t_size`` frames the loop keeps (kinetic-gan.py:129 crops after the copy) - shipped with an asynchronous copy
on a side stream while the previous batch trains (double buffered), and normalised ON the device, where the scale
and shift are two scalars.  Shuffling, ``drop_last`` and per-rank sharding for data parallelism are index arithmetic.
"""
from __future__ import annotations

import pickle
from typing import Iterator, Optional, Tuple

import numpy as np
import torch


class Feeder(torch.utils.data.Dataset):
    """Feeder for skeleton-based action synthesis (feeder/feeder.py:21-80).

    data_path: '.npy' of shape (N, C, T, V, M) for NTU and (N, C, T, V) for h36m;  label_path: pickle of
    (sample_name, label)."""

    def __init__(self, data_path, label_path, classes=None, norm=True, dataset='ntu', mmap=True):
        self.data_path = data_path
        self.label_path = label_path
        self.classes = classes
        self.norm = norm
        self.dataset = dataset
        self.load_data(mmap)

    def load_data(self, mmap):
        with open(self.label_path, 'rb') as f:
            self.sample_name, self.label = pickle.load(f)
        self.label = np.array(self.label, dtype=int)
        self.data = np.load(self.data_path, mmap_mode='r') if mmap else np.load(self.data_path)
        self.max, self.min = self.data.max(), self.data.min()
        if self.classes is not None:
            sel = np.where(np.isin(self.label, self.classes))
            tmp = self.label[sel]
            self.data = self.data[sel]
            self.label = np.nonzero(tmp[:, None] == self.classes)[1]
        if self.dataset == 'ntu':
            self.N, self.C, self.T, self.V, self.M = self.data.shape
        else:
            self.N, self.C, self.T, self.V = self.data.shape

    def __len__(self):
        return len(self.label)

    def __getitem__(self, index):
        data_numpy = np.array(self.data[index, :, :, :, 0]) if self.dataset == 'ntu' else np.array(self.data[index])
        if self.norm:
            data_numpy = 2 * ((data_numpy - self.min) / (self.max - self.min)) - 1
        return data_numpy, self.label[index]


class DeviceBatches:
    """Iterate (real (B, C, t_size, V) fp32 on `device`, labels (B,) int64 on `device`) over a Feeder.

    One epoch = the reference's DataLoader(batch_size, shuffle=True, drop_last=True) order for the permutation drawn
    from ``seed + epoch`` (numpy), restricted to every `world`-th batch starting at `rank` under data parallelism.
    Every rank yields the SAME number of batches per epoch - ``n_batches // world``, the trailing ``n_batches % world``
    batches of the epoch's permutation are dropped (what DistributedSampler(drop_last=True) does): each training
    iteration ends in collectives, so ranks with unequal step counts would pair batches of different epochs in one
    all-reduce and the shorter ranks would leave the last collectives hanging."""

    def __init__(self, feeder: Feeder, batch_size: int, t_size: int, device, shuffle: bool = True, drop_last: bool = True,
                 seed: int = 0, rank: int = 0, world: int = 1):
        self.f, self.bs, self.t = feeder, batch_size, min(t_size, feeder.T)
        self.device = torch.device(device)
        self.shuffle, self.drop_last, self.seed = shuffle, drop_last, seed
        self.rank, self.world = rank, world
        self.epoch = 0
        cuda = self.device.type == "cuda"
        shape = (batch_size, feeder.C, self.t, feeder.V)
        # two pinned staging buffers: the gather of batch i+1 overlaps the copy / use of batch i
        self._host = [torch.empty(shape, dtype=torch.float32, pin_memory=cuda) for _ in range(2)]
        self._lab = [torch.empty(batch_size, dtype=torch.int64, pin_memory=cuda) for _ in range(2)]
        self._stream = torch.cuda.Stream(device=self.device) if cuda else None
        span = float(feeder.max) - float(feeder.min)
        # 2 (x - min) / (max - min) - 1  ==  x * scale + shift
        self.scale = 2.0 / span if feeder.norm else 1.0
        self.shift = -2.0 * float(feeder.min) / span - 1.0 if feeder.norm else 0.0

    def _n_batches(self) -> int:
        """batches of one epoch PER RANK (identical on every rank)"""
        n = len(self.f) // self.bs if self.drop_last else -(-len(self.f) // self.bs)
        return n // self.world

    def __len__(self):
        return self._n_batches()

    def _order(self):
        idx = np.arange(len(self.f))
        if self.shuffle:
            np.random.RandomState(self.seed + self.epoch).shuffle(idx)
        per = self._n_batches()
        return [idx[b * self.bs:(b + 1) * self.bs] for b in range(self.rank, per * self.world, self.world)]

    def _gather(self, ids: np.ndarray, slot: int) -> int:
        """raw samples -> pinned buffer `slot` (sorted reads: the memory map is walked forwards)"""
        host, lab = self._host[slot], self._lab[slot]
        order = np.argsort(ids, kind="stable")
        dst = host.numpy()
        for pos in order:
            i = ids[pos]
            src = self.f.data[i, :, :self.t, :, 0] if self.f.dataset == 'ntu' else self.f.data[i, :, :self.t]
            dst[pos] = src
        lab.numpy()[:len(ids)] = self.f.label[ids]
        return len(ids)

    def _ship(self, slot: int, n: int) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.cuda.Event]]:
        if self._stream is None:
            x = self._host[slot][:n].clone()
            return x.mul_(self.scale).add_(self.shift), self._lab[slot][:n].clone(), None
        with torch.cuda.stream(self._stream):
            x = self._host[slot][:n].to(self.device, non_blocking=True)
            y = self._lab[slot][:n].to(self.device, non_blocking=True)
            x.mul_(self.scale).add_(self.shift)          # normalisation on the device
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return x, y, ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        batches = self._order()
        self.epoch += 1
        pending = None
        for b, ids in enumerate(batches):
            slot = b & 1
            if pending is not None and pending[3] is not None and b >= 2:
                pass        # slot reuse is safe: the copy out of it was waited for when its batch was yielded
            n = self._gather(ids, slot)
            nxt = self._ship(slot, n)
            if pending is not None:
                yield self._hand_over(pending)
            pending = (nxt[0], nxt[1], slot, nxt[2])
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, item):
        x, y, _, ev = item
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            x.record_stream(torch.cuda.current_stream(self.device))
            y.record_stream(torch.cuda.current_stream(self.device))
            ev.synchronize()     # the pinned slot may be refilled by the host after this point
        return x, y
