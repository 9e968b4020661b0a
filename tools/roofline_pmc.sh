#!/bin/bash
# Roofline evidence for the dominant kernel (GPU box): rocprofv3 kernel stats and HBM byte counters of
# `bench.py --roofline-only` (the disc-block-1 tail launch: the direct kg_conv_kernel<32,4>, the opt-in bf16-split tile kernel next to it).
# Counters in separate --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots), never together with tracing.
# usage: roofline_pmc.sh [batch]   (64: the bench's `roofline` leg -> roofline_pmc.json; 192: `roofline_critic` ->
# roofline_pmc_bs192.json)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-64}
export KG_RF_BATCH=$B
O=$R/gpurun_out/roofline_bs$B
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o rf -- python3 $R/bench.py --roofline-only --no-c5a --batch $B > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o rf -- python3 $R/bench.py --roofline-only --no-c5a --batch $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o rf -- python3 $R/bench.py --roofline-only --no-c5a --batch $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o rf -- python3 $R/bench.py --roofline-only --no-c5a --batch $B > $O/mfma.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, os
B = int(os.environ.get("KG_RF_BATCH", "64"))
O = "gpurun_out/roofline_bs%d" % B
KD = "kg_conv_kernel<32, 4"          # the direct fp32 kernel: the plan takes its 32-row tile for this (shallow) contraction
# the leg's kernel is the direct fp32 kernel at both sizes (round 6: the bf16-split form is opt-in, KG_CONV_BS=2 brings the
# round-5 plan rule back - then the 192-sample leg is its tile kernel); the other form is timed next to it on the same operands
BS = os.environ.get("KG_CONV_BS", "") in ("1", "2") and B >= 128
KN = "kg_conv_bsw_kernel" if BS else KD
KO = KD if BS else "kg_conv_bsw_kernel"
def per_launch(path, counter, kernel=None):
    kernel = kernel or KN
    vals = [float(r["Counter_Value"]) for f in glob.glob(path) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / max(1, len(vals)), len(vals)
fetch, nf = per_launch(O + "/fetch/*counter_collection.csv", "FETCH_SIZE")
write, nw = per_launch(O + "/write/*counter_collection.csv", "WRITE_SIZE")
busy, _ = per_launch(O + "/mfma/*counter_collection.csv", "SQ_VALU_MFMA_BUSY_CYCLES")
gui, _ = per_launch(O + "/mfma/*counter_collection.csv", "GRBM_GUI_ACTIVE")
stats = [r for f in glob.glob(O + "/stats/*kernel_stats.csv") for r in csv.DictReader(open(f)) if KN in r["Name"] or KD in r["Name"] or "kg_conv_bs_pack" in r["Name"]]
fetch_o, _ = per_launch(O + "/fetch/*counter_collection.csv", "FETCH_SIZE", KO)
write_o, _ = per_launch(O + "/write/*counter_collection.csv", "WRITE_SIZE", KO)
wg = [r for f in glob.glob(O + "/stats/*kernel_stats.csv") for r in csv.DictReader(open(f)) if "kg_wgrad" in r["Name"]]
rec = {
    "commit": os.environ.get("KG_COMMIT", "unknown"),
    "kernel": "%s disc block 1 tail, %d samples (bench.py --roofline-only --no-c5a --batch %d)" % ("kg_conv_bsw_kernel<2,1,4> (bf16-split tile kernel)" if BS else "kg_conv_kernel<32,4,true,1,2>", B, B),
    ("direct_fp32_kernel_hbm_bytes_per_launch" if BS else "bf16_split_tile_kernel_hbm_bytes_per_launch"): int((2 * fetch_o + write_o) * 1024),
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "launches_sampled": [nf, nw],
    # MI355X_MICROARCH.md: counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream
    "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
    "hbm_bytes_per_launch_uncorrected": int((fetch + write) * 1024),
    "algorithmic_min_bytes": (64 * B * 704 + 32 * B * 704 + 64 * B * 704) * 4 + (64 * 64 * 3 + 64 * 32) * 4,
    "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "GRBM_GUI_ACTIVE_per_launch": gui,
    "kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in stats],
    "wgrad_leg_kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in wg],
}
json.dump(rec, open(O + ("/roofline_pmc.json" if B == 64 else "/roofline_pmc_bs%d.json" % B), "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
find $O -type f ! -name "*stats.csv" ! -name "*.json" ! -name "*.log" -delete
