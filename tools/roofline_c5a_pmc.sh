#!/bin/bash
# rocprofv3 evidence for the two C5a legs of bench.py (roofline_c5a: kg_conv_kernel<128,4>, roofline_agg: kg_agg_mfma_kernel):
# kernel stats and HBM byte counters of `bench.py --roofline-only`, counters in separate --pmc passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/roofline_c5a
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o rf -- python3 $R/bench.py --roofline-only > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o rf -- python3 $R/bench.py --roofline-only > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o rf -- python3 $R/bench.py --roofline-only > $O/write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json
O = "gpurun_out/roofline_c5a"
def per_launch(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for f in glob.glob(path) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / max(1, len(vals)), len(vals)
import os
rec = {"commit": os.environ.get("KG_COMMIT", "unknown")}
for key, kern, algo in (("roofline_c5a", "kg_conv_kernel<128, 4", (64 * 512 * 256 * 25 * 3 + 512 * 512 * 3) * 4),
                        ("roofline_agg", "kg_agg_mfma_kernel<3, 1", 4 * 4 * 512 * 256 * 25 * 64)):
    fetch, nf = per_launch(O + "/fetch/*counter_collection.csv", "FETCH_SIZE", kern)
    write, nw = per_launch(O + "/write/*counter_collection.csv", "WRITE_SIZE", kern)
    stats = [r for f in glob.glob(O + "/stats/*kernel_stats.csv") for r in csv.DictReader(open(f)) if kern in r["Name"]]
    rec[key] = {"kernel": kern, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
                "launches_sampled": [nf, nw],
                # MI355X_MICROARCH.md: counters in KB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream
                "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
                "algorithmic_bytes": algo,
                "kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in stats]}
json.dump(rec, open(O + "/roofline_c5a_pmc.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
find $O -type f ! -name "*stats.csv" ! -name "*.json" ! -name "*.log" -delete
