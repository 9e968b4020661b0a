#!/bin/bash
# kernel-busy time vs wall time of the hipGraph-replayed iteration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/graphgap
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/log.txt 2>&1
cd $R
tail -1 $O/log.txt | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/graphgap/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last 10 iterations are graph replays: take the tail of the trace covering 10 * n kernels where n = launches per iteration
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
# find the adam kernel occurrences (2 per iteration) to delimit iterations
adam = [i for i, r in enumerate(rows) if 'kg_adam_kernel' in r['Kernel_Name']]
last = adam[-21:]            # 10 iterations = 20 adam launches (+1 boundary)
i0, i1 = last[0] + 1, last[-1] + 1
n = i1 - i0
busy = sum(en[i] - st[i] for i in range(i0, i1))
wall = en[i1 - 1] - st[i0]
gaps = [st[i + 1] - en[i] for i in range(i0, i1 - 1)]
gaps_pos = [g for g in gaps if g > 0]
print(f"10 iterations: {n} kernels ({n/10:.0f}/it), wall {wall/1e4:.1f} us/it, kernel busy {busy/1e4:.1f} us/it, sum of gaps {sum(gaps_pos)/1e4:.1f} us/it, median gap {sorted(gaps)[len(gaps)//2]} ns, overlapping pairs {sum(1 for g in gaps if g < 0)}")
PY
rm -rf $O
