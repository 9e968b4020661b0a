#!/usr/bin/env python3
"""Summarise the LAST graph replay in a rocprofv3 kernel trace: launches, kernel time, idle gaps, wall; biggest gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# the capture warm-up runs eagerly; the replays are the last nrep * n launches where n = launches per replay
# find n: the last launch name sequence repeats; use adam count
adam = [i for i, r in enumerate(rows) if "kg_adam" in r["Kernel_Name"]]
per = adam[-1] - adam[-2]
seg = rows[adam[-2] + 1: adam[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
wall = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3
gaps = []
for a, b in zip(seg, seg[1:]):
    gaps.append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:60], b["Kernel_Name"][:60], (int(a["Start_Timestamp"]) - t0) / 1e3))
print("launches %d  kernel time %.0f us  wall %.0f us  idle %.0f us" % (len(seg), dur, wall, wall - dur))
import collections
h = collections.Counter()
for g in gaps:
    h[min(int(g[0] // 2) * 2, 20)] += 1
print("gap histogram (us bucket: count):", sorted(h.items()))
for g in sorted(gaps, key=lambda t: -t[0])[:12]:
    print("  gap %7.1f us at t=%8.1f  after %-50s before %s" % (g[0], g[3], g[1], g[2]))

# optional: every launch inside a time window of the segment (KG_TL_WINDOW="t0,t1" in us), and the same summary for
# the critic half (between the two preceding Adam launches)
import os
if os.environ.get("KG_TL_WINDOW"):
    w0, w1 = (float(v) for v in os.environ["KG_TL_WINDOW"].split(","))
    prev_end = None
    for r in seg:
        st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if w0 <= st <= w1:
            print("  t=%8.1f gap %6.1f dur %6.1f  %s  grid=%s wg=%s" % (st, st - prev_end if prev_end is not None else 0.0, en - st,
                  r["Kernel_Name"][:70], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
        prev_end = en
if len(adam) >= 3:
    segd = rows[adam[-3] + 1: adam[-2] + 1]
    td = int(segd[0]["Start_Timestamp"])
    durd = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in segd) / 1e3
    walld = (int(segd[-1]["End_Timestamp"]) - td) / 1e3
    print("critic half: launches %d  kernel time %.0f us  wall %.0f us  idle %.0f us" % (len(segd), durd, walld, walld - durd))
    gd = sorted((((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:50], b["Kernel_Name"][:50],
                  (int(a["Start_Timestamp"]) - td) / 1e3) for a, b in zip(segd, segd[1:])), key=lambda t: -t[0])[:8]
    for g in gd:
        print("  gap %7.1f us at t=%8.1f  after %-50s before %s" % g)
