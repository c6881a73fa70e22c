#!/usr/bin/env python3
"""GPU occupancy of the timed part of a traced bench run, from rocprofv3's kernel_trace.csv:
busy = union of the dispatch intervals / wall, overlap = sum of the dispatch durations / union.
    python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction] [anchor kernel: window = its dispatches after the skip]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
anchor = sys.argv[3] if len(sys.argv) > 3 else None
t0, t1 = rows[0][0], max(r[1] for r in rows)
if anchor:   # window = from the dispatch of `anchor` at fraction `skip` of its dispatches to its last one (the timed steps)
    a = [r for r in rows if anchor in r[2]]
    lo, hi = a[int(len(a) * skip)][0], a[-1][1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi and "synth_render" not in r[2]]
else:
    lo = t0 + (t1 - t0) * skip                       # drop rendering + warm-up at the front
    rows = [r for r in rows if r[0] >= lo and "synth_render" not in r[2]]
wall = max(r[1] for r in rows) - rows[0][0]
union, cur_s, cur_e, total = 0, None, None, 0
for s, e, _ in rows:
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("dispatches %d  wall %.1f ms  busy(union) %.1f ms = %.1f %%  sum %.1f ms  overlap x%.2f" %
      (len(rows), wall / 1e6, union / 1e6, 100.0 * union / wall, total / 1e6, total / union))
by, cnt = {}, {}
for s, e, k in rows:
    k = k.replace("(anonymous namespace)::", "").split("(")[0]
    by[k] = by.get(k, 0) + (e - s)
    cnt[k] = cnt.get(k, 0) + 1
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:16]:
    print("  %-32s %8.1f ms  %5.1f %% of wall  %6d dispatches  avg %7.1f us" % (k, v / 1e6, 100.0 * v / wall, cnt[k], v / cnt[k] / 1e3))

# time-weighted histogram of the number of kernels in flight
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
hist, level, last = {}, 0, ev[0][0]
for t, d in ev:
    hist[level] = hist.get(level, 0) + (t - last)
    last = t
    level += d
tot = float(sum(hist.values()))
print("kernels in flight (share of the window): " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
