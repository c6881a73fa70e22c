#!/usr/bin/env python3
"""Where a group's chain of kernels waits: for every stream of a traced bench run, the gap between each kernel and the one before it
in the same stream, averaged by the name of the kernel that FOLLOWS the gap, inside a time window given as fractions of the trace.
    python tools/trace_chain.py <kernel_trace.csv[.gz]> <from_fraction> <to_fraction>"""
import csv
import gzip
import re
import sys
from collections import defaultdict

rows = []
op = gzip.open if sys.argv[1].endswith(".gz") else open
with op(sys.argv[1], "rt") as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.split(r"[(<]", name)[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * float(sys.argv[2]), t0 + (t1 - t0) * float(sys.argv[3])
rows = [r for r in rows if lo <= r[0] <= hi]
by = defaultdict(list)
for r in rows:
    by[r[3]].append(r)
gap = defaultdict(list)
dur = defaultdict(list)
for s, lst in by.items():
    if len(lst) < 50:
        continue
    for a, b in zip(lst, lst[1:]):
        gap[b[2]].append((b[0] - a[1]) / 1e3)
    for r in lst:
        dur[r[2]].append((r[1] - r[0]) / 1e3)
wall = (rows[-1][1] - rows[0][0]) / 1e3
n_streams = sum(1 for l in by.values() if len(l) >= 50)
tot_gap = sum(sum(v) for v in gap.values())
tot_dur = sum(sum(v) for v in dur.values())
print("window %.1f ms, %d streams; per stream: kernels %.1f ms, gaps %.1f ms" % (wall / 1e3, n_streams, tot_dur / n_streams / 1e3, tot_gap / n_streams / 1e3))
print("%-28s %6s %10s %10s %10s | %10s" % ("kernel after the gap", "n", "gap mean", "gap p50", "gap total", "dur mean"))
for k in sorted(gap, key=lambda k: -sum(gap[k])):
    v = sorted(gap[k])
    print("%-28s %6d %8.1f us %8.1f us %7.1f ms | %8.1f us" % (k, len(v), sum(v) / len(v), v[len(v) // 2], sum(v) / n_streams / 1e3, sum(dur[k]) / len(dur[k])))
