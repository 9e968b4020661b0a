#!/bin/bash
# usage: pmc_kernel.sh <script.py> <kernel-name substring> <tag>   - SQ counter passes for one kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; KERN=$2; TAG=$3
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    2) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT";;
    3) C="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python3 $R/$SCRIPT > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
KERN=$KERN TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os
kern, tag = os.environ['KERN'], os.environ['TAG']
out = open(f'gpurun_out/pmc_{tag}/summary.txt', 'w')
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/p*/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in acc:
        line = '%-12s %-34s %16.0f per launch (%d launches)' % (tag, k, acc[k] / n[k], n[k])
        print(line); out.write(line + '\n')
PY
find gpurun_out/pmc_$TAG -name "*.csv" -delete
