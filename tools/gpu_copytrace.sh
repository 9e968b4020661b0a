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
# last third of the trace = the last step
idx = [i for i, n in enumerate(names) if 'copyBuffer' in n]
idx = idx[-60:]
ctx = collections.Counter()
for i in idx:
    prev = short(names[i - 1]) if i > 0 else ''
    nxt = short(names[i + 1]) if i + 1 < len(names) else ''
    ctx[(prev, nxt)] += 1
for (p, n), c in ctx.most_common(12):
    print(c, '|', p, '  ->  COPY  ->  ', n)
# context of the runs of consecutive copies
i = 0
runs = []
while i < len(names):
    if 'copyBuffer' in names[i]:
        j = i
        while j < len(names) and 'copyBuffer' in names[j]: j += 1
        if j - i >= 5: runs.append((i, j))
        i = j
    else:
        i += 1
for (i, j) in runs[-4:]:
    print('RUN of', j - i, 'copies; before:', [short(n)[:40] for n in names[max(0, i - 4):i]], ' after:', [short(n)[:40] for n in names[j:j + 4]])
    print('   sizes(grid*wg):', [(rows[k]['Grid_Size'], rows[k]['Workgroup_Size']) for k in range(i, min(j, i + 6))])
PY
rm -rf $O
