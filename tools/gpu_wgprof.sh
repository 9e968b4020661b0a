#!/bin/bash
# GPU box: per-launch times of kg_wgrad_many inside the eager iteration for the in-tree library and each variant
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base ${VARIANTS:-}; do
  if [ $v = base ]; then unset KG_LIB; else export KG_LIB=$R/build_ab/libkgan_$v.so; fi
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wgprof_$v -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/wgprof_$v.log 2>&1
  cd $R
  echo "== $v"
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/wgprof_$v/**/t_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "wgrad" in r["Kernel_Name"]]
rows=rows[-6:]
for r in rows: print(r["Kernel_Name"][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size"))
PY
  rm -rf gpurun_out/wgprof_$v
done
