#!/bin/bash
# which kernels surround the D2D memcpys of one step (eager)?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/copytrace
rm -rf $O; mkdir -p $O
cd /tmp
export REPS=1
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/tools/one_gstep.py > $O/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/copytrace/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')
    return n[:70]
# the last step: from the last-but-one kg_adam_kernel to the last one
adam = [i for i, n in enumerate(names) if 'kg_adam_kernel' in n]
i0, i1 = adam[-2] + 1, adam[-1] + 1
step = list(range(i0, i1))
copies = [i for i in step if 'copyBuffer' in names[i]]
print(len(step), 'kernels in the step,', len(copies), 'copyBuffer launches')
for i in copies:
    print('grid', rows[i]['Grid_Size'], '| before:', short(names[i - 2])[:34], '|', short(names[i - 1])[:34], '| after:', short(names[i + 1])[:34], '|', short(names[i + 2])[:34])
PY
rm -rf $O
