#!/bin/bash
# other BASELINE configs on one GPU + the 2-rank control flow (gloo on one GPU), default and exact-BN mode
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
: > gpurun_out/r05_final_other_configs.log
for c in ${R4K_CONFIGS:-"stress 64" "ntu120 32" "h36m 64"}; do set -- $c
  echo "== --config $1 --batch $2" >> gpurun_out/r05_final_other_configs.log
  timeout 600 python bench.py --config $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | cut -c1-700 >> gpurun_out/r05_final_other_configs.log
done
cat gpurun_out/r05_final_other_configs.log | cut -c1-260
export KG_BENCH_BACKEND=gloo KG_BENCH_DEVICE=0
: > gpurun_out/r05_dp2_one_gpu_gloo.log
for extra in "" "--exact-bn"; do
  echo "== 2 ranks on one GPU over gloo $extra" >> gpurun_out/r05_dp2_one_gpu_gloo.log
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras $extra 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -2 | cut -c1-900 >> gpurun_out/r05_dp2_one_gpu_gloo.log
done
cat gpurun_out/r05_dp2_one_gpu_gloo.log | cut -c1-400
