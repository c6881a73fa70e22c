#!/usr/bin/env python3
"""Per-stream gaps between back-to-back dependent kernels in a rocprofv3 kernel trace (no host work in between):
how long a queued kernel waits after its predecessor in the same stream has finished.
    python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", "0"), r.get("Queue_Id", "0")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
by_stream = defaultdict(list)
for r in rows:
    by_stream[r[3]].append(r)
queues = defaultdict(set)
for r in rows:
    queues[r[4]].add(r[3])
print("streams %d, hardware queues %d (streams per queue: %s)" % (len(by_stream), len(queues), sorted(len(v) for v in queues.values())))
pairs = [("pyr_down_kernel", "pyr_down_kernel"), ("pyr_down_kernel", "image_align_lds_kernel"), ("fast_cells_wave_kernel", "select_cells_kernel"),
         ("select_cells_kernel", "select_pack_kernel"), ("search_prepare_kernel", "search_points_kernel"),
         ("select_matches_kernel", "pose_hypotheses_kernel"), ("pose_hypotheses_kernel", "pose_refine_kernel"),
         ("shi_tomasi_kernel", "orb_describe_kernel"), ("orb_describe_kernel", "filter_gather_kernel"), ("shi_tomasi_kernel", "filter_gather_kernel")]
gaps = defaultdict(list)
for s, lst in by_stream.items():
    for a, b in zip(lst, lst[1:]):
        if (a[2], b[2]) in pairs:
            gaps[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
for k, v in gaps.items():
    v.sort()
    print("  %-26s -> %-26s n=%5d  gap median %7.1f us  p90 %7.1f us  mean %7.1f us" % (k[0], k[1], len(v), v[len(v) // 2], v[int(len(v) * 0.9)], sum(v) / len(v)))
