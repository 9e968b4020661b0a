#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gstep
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gstep -o g -- python3 $R/tools/one_gstep.py > $R/gpurun_out/gstep/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/gstep/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
n = 7
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/n:6.1f}/it {float(r['TotalDurationNs'])/n/1e3:8.0f}us/it {float(r['AverageNs'])/1e3:7.1f}")
print(tot / n / 1e6, "ms/iter kernel time;", sum(int(r['Calls']) for r in rows) / n, "launches/iter")
PY
find gpurun_out/gstep -name "*trace*" -delete
