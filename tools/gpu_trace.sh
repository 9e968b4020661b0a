#!/bin/bash
# dispatch traces (aten ops + issuing repo line) of the critic and generator steps, plus a per-launch kernel trace of one eager step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
WHICH=d timeout 300 python tools/dispatch_trace.py > gpurun_out/trace_d.log 2>&1
WHICH=g timeout 300 python tools/dispatch_trace.py > gpurun_out/trace_g.log 2>&1
tail -3 gpurun_out/trace_d.log gpurun_out/trace_g.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o kt -- python3 $R/bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/ktrace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last step: find the last kg_adam launches
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "kg_adam" in n]
lo = adam[-3] + 1 if len(adam) >= 3 else 0
with open("gpurun_out/ktrace_last_step.txt", "w") as o:
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:]:
        n = r["Kernel_Name"]
        n = n.split("(")[0][-70:] if "kg_" in n else n[:90]
        o.write("%9.1f %7.1f  %s  grid=%s wg=%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
print("launches in last step:", len(rows) - lo)
PY
find gpurun_out/ktrace -type f -name "*.csv" -size +3M -delete
