"""ORACLE (test infrastructure): numpy restatement of the reference's data path for one sample / one batch.

* ``sample``  <- feeder/feeder.py:74-80 (__getitem__) with the statistics of :59 (global max / min) and the class
  subset of :61-64
* ``batch``   <- kinetic-gan.py:68-74,129-131: DataLoader collation of Feeder samples, crop to t_size, fp32 / int64
Pinned by tests/golden/feeder_ref.npz, produced from the imported reference Feeder by
tests/golden/make_feeder_fixture.py.  Only tests/ may import this file.
"""
import pickle

import numpy as np


def load(data_path, label_path, classes=None, dataset="ntu"):
    with open(label_path, "rb") as f:
        _, label = pickle.load(f)
    label = np.array(label, dtype=int)
    data = np.load(data_path)
    mx, mn = data.max(), data.min()                      # feeder.py:59 (before the class subset)
    if classes is not None:                              # feeder.py:61-64
        sel = np.where(np.isin(label, classes))
        tmp = label[sel]
        data = data[sel]
        label = np.nonzero(tmp[:, None] == classes)[1]
    return data, label, mx, mn


def sample(data, label, mx, mn, index, dataset="ntu", norm=True):
    x = np.array(data[index, :, :, :, 0]) if dataset == "ntu" else np.array(data[index])     # feeder.py:76
    if norm:
        x = 2 * ((x - mn) / (mx - mn)) - 1                                                    # feeder.py:77
    return x, label[index]


def batch(data, label, mx, mn, ids, t_size, dataset="ntu", norm=True):
    xs = np.stack([sample(data, label, mx, mn, i, dataset, norm)[0] for i in ids])
    return xs[:, :, :t_size, :].astype(np.float32), np.asarray([label[i] for i in ids], dtype=np.int64)   # kinetic-gan.py:129-131
