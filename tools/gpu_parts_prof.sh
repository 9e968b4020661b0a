#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
for T in 1 0; do
  export KG_TRUNK=$T PART=g_step
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pp -o t$T -- python3 $R/tools/one_part.py > $R/gpurun_out/pp_t$T.log 2>&1
  cd $R
  echo "== KG_TRUNK=$T g_step"; python tools/graph_timeline.py gpurun_out/pp/t${T}_kernel_trace.csv
done
rm -rf gpurun_out/pp
