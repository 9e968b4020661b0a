#!/bin/bash
# round 5 measurement records (GPU box): per-family HBM traffic of one iteration, the C5a PMC passes, and the data-parallel
# launch structures on ONE GPU (segmented graphs + eager Adam; kg_allreduce_flat captured into the graph) - round-4 VERDICT
# items 6 and 7.  Outputs under gpurun_out/, copied to profiles/r05_* by the caller.
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
bash tools/gpu_traffic.sh > gpurun_out/r05_traffic_per_family.log 2>&1; tail -25 gpurun_out/r05_traffic_per_family.log
KG_COMMIT=${KG_COMMIT:-r05} bash tools/roofline_c5a_pmc.sh > gpurun_out/roofline_c5a_pmc.log 2>&1; tail -5 gpurun_out/roofline_c5a_pmc.log
{
echo "# single GPU, whole-iteration hipGraph (default)"
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>&1 | grep "^{.metric" | cut -c1-420
echo "# single GPU, data-parallel launch structure: two graphs + eager all-reduce slot + eager Adam (--segmented)"
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras --segmented 2>&1 | grep "^{.metric" | cut -c1-420
echo "# single GPU, kg_comm communicator, all-reduce + Adam captured into the step's graph (--comm kg --dp-graph --segmented)"
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras --comm kg --dp-graph --segmented 2>&1 | grep "^{.metric" | cut -c1-420
echo "# single GPU, kg_comm communicator, eager all-reduce (--comm kg --segmented)"
timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras --comm kg --segmented 2>&1 | grep "^{.metric" | cut -c1-420
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dp_structures_one_gpu.log
bash tools/gpu_dp_check.sh 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r05_dp2_one_gpu_gloo.log
