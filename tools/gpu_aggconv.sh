#!/bin/bash
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "aggconv" 2>&1 | tail -5
rm -f gpurun_out/time_aggconv.log
for P in ${PLANS:-321 641}; do
  echo "== KG_AGGCONV_PLAN=$P" | tee -a gpurun_out/time_aggconv.log
  KG_AGGCONV_PLAN=$P timeout 600 python tools/time_aggconv.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/time_aggconv.log
done
