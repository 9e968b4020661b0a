set -u
mkdir -p gpurun_out
KTESTS="-k genblock" SKIP_PARITY=1 SKIP_PROF=1 VARIANTS="staged:KG_GEN_FUSED=0" bash tools/gpu_r06.sh tapelate
timeout 300 python tools/time_genblock.py > gpurun_out/tg_tapelate.log 2>&1; tail -12 gpurun_out/tg_tapelate.log
KG_LIB=$GRAFT_REPO_ROOT/build_ab/libkgan_gbstamp.so timeout 300 python tools/time_genblock.py > gpurun_out/tg_tapelate_stamp.log 2>&1; tail -30 gpurun_out/tg_tapelate_stamp.log
