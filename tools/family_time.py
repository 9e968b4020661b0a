#!/usr/bin/env python3
"""rocprofv3 kernel stats CSV of `bench.py --steps S --warmup W --no-graph` -> per-family kernel time per iteration
(profiles/r<NN>_final_eager_kernel_stats.json, read by bench.py's work.family_rates).

    python tools/family_time.py <kernel_stats.csv> <iterations in the profile> <commit> > profiles/r<NN>_final_eager_kernel_stats.json
"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2])
fam = {"kg_conv": ("kg_conv_kernel", "kg_conv_many_kernel", "kg_conv_tiny_kernel", "kg_conv_splitk_epilogue", "kg_conv_bsw_kernel", "kg_conv_bs_kernel", "kg_conv_bs_pack_kernel"), "kg_wgrad": ("kg_wgrad",),
       "kg_aggconv": ("kg_aggconv",), "kg_agg": ("kg_agg_",), "kg_genblock": ("kg_genblock",)}
us, calls = {k: 0.0 for k in fam}, {k: 0 for k in fam}
total = 0.0
for r in rows:
    name, t, n = r["Name"], float(r["TotalDurationNs"]) / 1e3, int(r["Calls"])
    total += t
    for k, pats in fam.items():
        if any(p in name for p in pats) and not (k == "kg_agg" and "kg_aggconv" in name):
            us[k] += t
            calls[k] += n
            break
print(json.dumps({"commit": sys.argv[3], "source": sys.argv[1], "iterations": iters,
                  "kernel_us_per_step": {k: round(v / iters, 1) for k, v in us.items()},
                  "launches_per_step": {k: round(v / iters, 1) for k, v in calls.items()},
                  "all_kernels_us_per_step": round(total / iters, 1)}, indent=1))
