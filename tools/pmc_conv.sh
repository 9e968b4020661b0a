#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+|\bTCC_[A-Z_0-9]+|\bGRBM_[A-Z_]+|\bTCP_[A-Z_0-9]+|\bTA_[A-Z_0-9]+" | sort -u > $R/gpurun_out/pmc/counters.txt
wc -l $R/gpurun_out/pmc/counters.txt
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    2) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT";;
    3) C="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/p$i -o p -- python3 $R/tools/one_conv.py > $R/gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/p*/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'kg_conv' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(f)
    for k in acc: print('   %-34s %16.0f per launch (%d launches)' % (k, acc[k] / n[k], n[k]))
PY
