#!/bin/bash
# kernel stats + timeline of the hipGraph-replayed iteration (what bench.py times)
set -u
TAG=${1:-graph}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_graph -o $TAG -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/prof_graph.log 2>&1
echo "prof rc=$?"
cd $R
python tools/last_step.py gpurun_out/prof_graph/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_graph_last_step.txt 2> gpurun_out/last_step.err; tail -3 gpurun_out/${TAG}_graph_last_step.txt
KG_TL_WINDOW=${KG_TL_WINDOW:-1200,1600} python tools/graph_timeline.py gpurun_out/prof_graph/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_graph_timeline.txt 2>> gpurun_out/last_step.err; cat gpurun_out/${TAG}_graph_timeline.txt
find gpurun_out/prof_graph -type f ! -name "*stats*" -delete
tail -1 gpurun_out/prof_graph.log
for i in 1 2; do timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras | tail -1 | cut -c1-200; done
