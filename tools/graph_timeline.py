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
