#!/usr/bin/env python3
"""Per-launch timeline + per-kernel summary of the LAST iteration in a rocprofv3 kernel trace of bench.py --no-graph."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "kg_adam" in n]
lo = adam[-3] + 1 if len(adam) >= 3 else 0
rows = rows[:adam[-1] + 1]          # bench.py's executed-FLOP leg runs two compute halves (no Adam) after the last iteration
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("kg_"):
        return n.split("(")[0]
    if n.startswith("Cijk"):
        return "hipblaslt:" + re.search(r"MT\d+x\d+x\d+", n).group(0)
    m = re.search(r"at::native::(\w+)<", n)
    k = m.group(1) if m else n[:40]
    for key in ("FillFunctor", "CUDAFunctor_add", "MulFunctor", "leaky_relu_backward", "leaky_relu", "direct_copy", "normal", "MeanOps",
                "sum_functor", "index_kernel", "DivFunctor", "neg", "pow", "CatArray", "scatter_gather", "embedding", "BinaryOpList",
                "norm", "masked_fill", "compare", "sub", "AddFunctor"):
        if key in n:
            k += ":" + key
            break
    return k
t0 = int(rows[lo]["Start_Timestamp"])
cnt = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for i, r in enumerate(rows[lo:]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = short(r["Kernel_Name"])
    print("%4d %9.1f dur%7.1f  %-60s grid=%sx%sx%s wg=%s" % (i, (s - t0) / 1e3, (e - s) / 1e3, nm, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
    k = re.sub(r"<.*", "", nm) if nm.startswith("kg_") else ("hipblaslt" if nm.startswith("hipblaslt") else "aten:" + nm.split(":")[-1])
    cnt[k][0] += 1
    cnt[k][1] += (e - s) / 1e3
    tot += (e - s) / 1e3
print("---- summary of the last iteration")
for k, (n, d) in sorted(cnt.items(), key=lambda kv: -kv[1][1]):
    print("   %-40s %4d %8.1f us" % (k, n, d))
print("launches in last step: %d, sum of kernel time %.0f us" % (len(rows) - lo, tot))
