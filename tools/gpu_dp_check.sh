#!/bin/bash
# the multi-rank control flow of bench.py on ONE GPU: 2 ranks, gloo instead of RCCL, both pinned to device 0
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
export KG_BENCH_BACKEND=gloo KG_BENCH_DEVICE=0
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -5 | cut -c1-600 | tee gpurun_out/dp2_one_gpu_gloo.log
echo "--- no overlap"
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-overlap 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -3 | cut -c1-400 | tee -a gpurun_out/dp2_one_gpu_gloo.log
unset KG_BENCH_BACKEND KG_BENCH_DEVICE
echo "--- single GPU, segmented+overlap structure"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --segmented 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/dp2_one_gpu_gloo.log
echo "--- launcher path: python bench.py --gpus 2 (expected to fail cleanly on a 1-GPU box or run 2 ranks)"
KG_BENCH_BACKEND=gloo KG_BENCH_DEVICE=0 timeout 240 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -2 | cut -c1-300
