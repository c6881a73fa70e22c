#!/usr/bin/env python3
"""Per-frame timeline of a lone camera's chain out of a rocprofv3 kernel trace (tools/lone_camera_trace.sh): frames are delimited by
frames_upload; prints, for the frame with the median span, every kernel with its start offset, duration and the gap in front of it, and the
per-kernel medians over all non-keyframe frames.
    python tools/lone_timeline.py gpurun_out/TAG/kernel_trace.csv"""
import csv, re, sys, statistics as st
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.split(r"[(<]", name)[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
first = "frames_upload_kernel" if any(r[2] == "frames_upload_kernel" for r in rows) else "pyr_down_kernel"
frames, cur = [], []
for r in rows:
    if r[2] == first and (not cur or cur[-1][2] != first):
        if cur: frames.append(cur)
        cur = []
    cur.append(r)
frames.append(cur)
frames = [f for f in frames[5:-1]]
plain = [f for f in frames if not any(k[2].startswith("shi_tomasi") for k in f)]
span = lambda f: (max(k[1] for k in f) - f[0][0]) / 1e3
plain.sort(key=span)
med = plain[len(plain) // 2]
print("frames %d (without keyframe kernels %d); span p50 %.1f us, min %.1f, p90 %.1f" % (len(frames), len(plain), span(med), span(plain[0]), span(plain[int(len(plain) * .9)])))
t0 = med[0][0]
prev_end = {}
print("%-36s %8s %8s %8s  %s" % ("kernel", "start", "dur", "gap", "stream"))
last_end = t0
for k in med:
    print("%-36s %8.1f %8.1f %8.1f  %s" % (k[2], (k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, (k[0] - last_end) / 1e3, k[3]))
    last_end = max(last_end, k[1])
dur = defaultdict(list)
for f in plain:
    c = defaultdict(float)
    for k in f: c[k[2]] += (k[1] - k[0]) / 1e3
    for n, v in c.items(): dur[n].append(v)
print("\nper-frame kernel time, median over frames:")
tot = 0
for n, v in sorted(dur.items(), key=lambda kv: -st.median(kv[1])):
    print("  %-36s %7.1f" % (n, st.median(v))); tot += st.median(v)
print("  %-36s %7.1f" % ("sum", tot))
