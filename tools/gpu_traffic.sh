#!/bin/bash
# HBM traffic per kernel family of one eager G+D iteration: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extras > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extras > $O/write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, re
O = "gpurun_out/traffic"
def fam(n):
    m = re.search(r"(kg_[a-z0-9_]+?)(_kernel|<|\(|$)", n)
    return m.group(1) if m else ("hipblaslt" if "Cijk" in n or "hipblaslt" in n.lower() else "aten/other")
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for key, idx in (("fetch", 0), ("write", 1)):
    for f in glob.glob(f"{O}/{key}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            t = tot[fam(r["Kernel_Name"])]
            t[idx] += float(r["Counter_Value"])
            if idx == 0: t[2] += 1
# iterations that really ran (warm-up + timed + the work-accounting iteration of bench.work_leg): kg_mix3 runs once per
# iteration (round-3 VERDICT: a hard-wired 3 divided what 4 iterations had moved)
iters = float(max(1, tot["kg_mix3"][2]))
print(f"iterations in the trace: {iters:.0f}")
print("family                      launches/it   fetch MB/it (x2 corrected)   write MB/it   total MB/it")
g = 0.0
for k, (fe, wr, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    fmb, wmb = 2 * fe / 1024 / iters, wr / 1024 / iters
    g += fmb + wmb
    print(f"{k:28s} {n / iters:8.1f} {fmb:16.1f} {wmb:22.1f} {fmb + wmb:14.1f}")
print(f"all kernels: {g / 1024:.2f} GB per iteration (8 TB/s x 3.6 ms = 29 GB)")
PY
find $O -type f ! -name "*.log" -delete
