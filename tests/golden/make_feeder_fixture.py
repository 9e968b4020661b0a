#!/usr/bin/env python3
"""Golden vectors for the data path from the REFERENCE's own Feeder (feeder/feeder.py), dev container only.

A tiny synthetic dataset (seeded; written next to this script as data files, they ARE the fixture inputs) is read
by the imported reference class; its __getitem__ outputs, statistics and lengths go to feeder_ref.npz.  The
reference module imports torchvision / PIL / scipy at the top without using them in the class: absent here, they
are satisfied by empty module objects for the duration of the import.

    python tests/golden/make_feeder_fixture.py
"""
import os
import pickle
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
import numpy as np  # noqa: E402

rs = np.random.RandomState(7)
ntu = (rs.randn(12, 3, 20, 25, 2) * 1.7 + 0.3).astype(np.float32)
h36 = (rs.randn(9, 2, 16, 16) * 0.8 - 0.1).astype(np.float32)
np.save(os.path.join(HERE, "feeder_ntu_data.npy"), ntu)
np.save(os.path.join(HERE, "feeder_h36m_data.npy"), h36)
lab_ntu = [int(v) for v in rs.randint(0, 5, 12)]
lab_h36 = [int(v) for v in rs.randint(0, 4, 9)]
pickle.dump(([f"s{i}" for i in range(12)], lab_ntu), open(os.path.join(HERE, "feeder_ntu_label.pkl"), "wb"))
pickle.dump(([f"s{i}" for i in range(9)], lab_h36), open(os.path.join(HERE, "feeder_h36m_label.pkl"), "wb"))

for name in ("torchvision", "PIL", "scipy", "scipy.ndimage"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
for name, attrs in (("torchvision", ("datasets", "transforms")), ("PIL", ("Image",)), ("scipy.ndimage", ("gaussian_filter1d",))):
    for a in attrs:
        if not hasattr(sys.modules[name], a):
            setattr(sys.modules[name], a, None)
sys.path.insert(0, "/root/reference")
from feeder.feeder import Feeder  # noqa: E402

out = {}
for tag, ds, classes in (("ntu", "ntu", None), ("ntu_sub", "ntu", [3, 1]), ("h36m", "h36m", None), ("h36m_raw", "h36m", None)):
    base = "ntu" if ds == "ntu" else "h36m"
    f = Feeder(os.path.join(HERE, f"feeder_{base}_data.npy"), os.path.join(HERE, f"feeder_{base}_label.pkl"),
               classes=classes, norm=(tag != "h36m_raw"), dataset=ds)
    out[tag + "_len"] = np.int64(len(f))
    out[tag + "_max"], out[tag + "_min"] = np.float64(f.max), np.float64(f.min)
    out[tag + "_x"] = np.stack([np.asarray(f[i][0]) for i in range(len(f))])
    out[tag + "_y"] = np.asarray([f[i][1] for i in range(len(f))], dtype=np.int64)
np.savez(os.path.join(HERE, "feeder_ref.npz"), **out)
print({k: getattr(v, "shape", v) for k, v in out.items()})
