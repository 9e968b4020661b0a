#!/usr/bin/env python3
"""Times kg_conv at the discriminator's 13 shapes (N = 64 and 192 samples, hipGraph replay of 20 launches) with the
library named by KG_LIB (default: the in-tree one) and writes one line per case; tools/exp_conv.sh runs it once per
variant and tabulates.  KG_EXP_N="64,192", KG_EXP_CASES=substring filter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kinetic_gan_amd
from kinetic_gan_amd import _native as nv
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("ab_conv_cases", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_conv.py"))
src = open(spec.origin).read().split("ab = os.environ.get")[0]        # the case builders only
ns = {"__name__": "ab_conv_cases", "__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
cases, timeit = ns["cases"], ns["timeit"]
filt = os.environ.get("KG_EXP_CASES")
tag = os.environ.get("KG_EXP_TAG", os.path.basename(os.environ.get("KG_LIB", "base")))
for N in [int(v) for v in os.environ.get("KG_EXP_N", "64,192").split(",")]:
    tot = 0.0
    for name, (fn, flops) in cases(N).items():
        if filt and not any(f in name for f in filt.split(",")):
            continue
        out = fn()
        chk = out.double().abs().mean().item()
        t = min(timeit(fn), timeit(fn))
        tot += t
        print(f"RES {tag} N={N} | {name} | {t:.2f} us | {flops/t/1e6:.1f} TF | chk {chk:.6e}", flush=True)
    print(f"RES {tag} N={N} | total | {tot:.2f} us | - | -", flush=True)
