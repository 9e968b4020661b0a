#!/usr/bin/env python3
"""Key / shape / dtype listing of the REFERENCE's Generator and Discriminator state_dicts (checkpoint drop-in
contract: generate.py:66, kinetic-gan.py:189-192) for the BASELINE configs.  Dev container only (imports
/root/reference under the CPU shims of make_fixtures.py); writes tests/golden/state_dict_listing.json.

    python tests/golden/make_state_dict_listing.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
from models.generator import Generator  # noqa: E402
from models.discriminator import Discriminator  # noqa: E402

CFG = {
    "ntu": dict(channels=3, n_classes=60, t_size=64, latent=512, mlp=4, dataset="ntu"),
    "ntu120": dict(channels=3, n_classes=120, t_size=64, latent=512, mlp=8, dataset="ntu"),
    "h36m": dict(channels=2, n_classes=10, t_size=32, latent=512, mlp=4, dataset="h36m"),
    "stress": dict(channels=3, n_classes=60, t_size=256, latent=512, mlp=4, dataset="ntu"),
}


def listing(m):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]


out = {}
for name, c in CFG.items():
    G = Generator(c["latent"], c["channels"], c["n_classes"], c["t_size"], c["mlp"], dataset=c["dataset"])
    D = Discriminator(c["channels"], c["n_classes"], c["t_size"], c["latent"], dataset=c["dataset"])
    out[name] = {"G": listing(G), "D": listing(D)}
json.dump(out, open(os.path.join(HERE, "state_dict_listing.json"), "w"), indent=0)
print({k: (len(v["G"]), len(v["D"])) for k, v in out.items()})
