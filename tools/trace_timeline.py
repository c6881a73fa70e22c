#!/usr/bin/env python3
"""One stream's kernels over a couple of steps of a traced bench run: start offset, duration, gap to the previous one.
    python tools/trace_timeline.py <kernel_trace.csv> [stream_rank] [n_rows]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", "0")))
rows.sort()
by = defaultdict(list)
for r in rows:
    by[r[3]].append(r)
streams = sorted(by, key=lambda s: -len(by[s]))
sid = streams[int(sys.argv[2]) if len(sys.argv) > 2 else 3]
lst = by[sid]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
lst = lst[len(lst) * 2 // 3:][:n]
t0 = lst[0][0]
prev_end = None
for s, e, k, _ in lst:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +%7.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, k))
    prev_end = e
