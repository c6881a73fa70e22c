#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the small CSV / JSON files kept under profiles/.
    python tools/summarize_profiles.py <gpurun_out> <dest_dir>"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def biggest(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(files, key=os.path.getsize) if files else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0]


# kernel-trace stats of the bench process (the largest stats file belongs to the python process, not to helpers)
ks = biggest("prof_kt/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, "kernel_stats.csv"))
ds = biggest("prof_kt/**/*domain_stats.csv")
if ds:
    shutil.copy(ds, os.path.join(dst, "domain_stats.csv"))
for name in ("bench_default.json", "prof_kt.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "bench_under_rocprof.json" if name == "prof_kt.json" else name))

# dispatches per group-step: calls of every kernel in the traced run / (bootstrap + warm-up + timed steps) x groups
try:
    cfgj = json.loads(open(os.path.join(src, "prof_kt.json")).read().strip().splitlines()[-1])
    gsteps = (1 + cfgj["warmup"] + cfgj["steps"]) * cfgj["config"]["groups_per_gpu"]
    rows = list(csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))))
    with open(os.path.join(dst, "dispatches_per_group_step.csv"), "w") as out:
        out.write("kernel,calls,calls_per_group_step,avg_us,note\n")
        tot = 0.0
        for r in rows:
            name = short(r["Name"])
            if name.startswith(("synth_render", "__amd_rocclr_fillBuffer")):
                continue   # set-up (rendering the inputs, zeroing the frame pool's headers), not part of a step
            per = int(r["Calls"]) / gsteps
            tot += per
            csv.writer(out).writerow([name, r["Calls"], "%.2f" % per, "%.1f" % (float(r["AverageNs"]) / 1e3),
                                      "%d group-steps in the traced run (%d groups x (1 bootstrap + %d warm-up + %d timed))" % (gsteps, cfgj["config"]["groups_per_gpu"], cfgj["warmup"], cfgj["steps"])])
        csv.writer(out).writerow(["TOTAL", "", "%.2f" % tot, "", "keyframe-only kernels (shi_tomasi, filter_*, track_upload, stage_pull, frames_own) run in the group-steps that have keyframes"])
except Exception as e:   # the summary is a convenience; a missing trace must not lose the other files
    print("dispatch count not written:", e)

# PMC passes: per kernel average FETCH_SIZE / WRITE_SIZE (KB as reported by rocprofv3)
acc = {}
for counter, sub in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    f = biggest(sub + "/**/*counter_collection.csv")
    if not f:
        continue
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            a = acc.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            a[counter][0] += float(row["Counter_Value"])
            a[counter][1] += 1
cfg = ""
pj = os.path.join(src, "prof_fetch.json")
if os.path.exists(pj) and os.path.getsize(pj):
    try:
        c = json.loads(open(pj).read().strip().splitlines()[-1])["config"]
        cfg = "%d sequences in %d groups -> %d frames per dispatch" % (c["sequences_per_gpu"], c["groups_per_gpu"], c["sequences_per_group"])
    except Exception:
        pass
with open(os.path.join(dst, "pmc_hbm_traffic.csv"), "w") as out:
    # fetch_x2: on gfx950 FETCH_SIZE tallies a 128-B fabric request as 64 B (MI355X_MICROARCH.md, HBM).  The guide calibrates that for
    # 16-B-per-lane streams; these kernels load 1-4 B per lane, and the one whose compulsory traffic is known exactly — fast_cells reads
    # pyramid levels 0-2 of every frame once, 403,200 B x frames — shows the same half (round 2: 52 MB by the counter, 103 MB
    # compulsory), so the x2 is applied to EVERY row: HBM bytes = (FETCH_SIZE_avg_KB * 2 + WRITE_SIZE_avg_KB) * 1024.
    out.write("kernel,dispatches,FETCH_SIZE_avg_KB,WRITE_SIZE_avg_KB,fetch_x2,hbm_MB_per_dispatch,note\n")
    note = "separate --pmc passes; %s; raw counter values, the x2 of gfx950's FETCH_SIZE applied in hbm_MB_per_dispatch only" % cfg
    for k in sorted(acc):
        a = acc[k]
        n = max(a["FETCH_SIZE"][1], a["WRITE_SIZE"][1])
        fa = a["FETCH_SIZE"][0] / a["FETCH_SIZE"][1] if a["FETCH_SIZE"][1] else float("nan")
        wa = a["WRITE_SIZE"][0] / a["WRITE_SIZE"][1] if a["WRITE_SIZE"][1] else float("nan")
        csv.writer(out).writerow([k, n, "%.2f" % fa, "%.2f" % wa, 1, "%.2f" % ((fa * 2 + wa) * 1024 / 1e6), note])

# SQ pass (LDS activity / bank conflicts, wave counts, VALU issue): per kernel averages per dispatch, every counter one column
f = biggest("prof_sq/**/*counter_collection.csv")
if f:
    sq, names = {}, []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k, c = short(row["Kernel_Name"]), row["Counter_Name"]
            if c not in names:
                names.append(c)
            a = sq.setdefault(k, {})
            v = a.setdefault(c, [0.0, 0])
            v[0] += float(row["Counter_Value"])
            v[1] += 1
    with open(os.path.join(dst, "pmc_sq_lds.csv"), "w") as out:
        out.write("kernel,dispatches," + ",".join(n + "_avg" for n in names) + ",lds_conflict_share,valu_insts_per_wave,note\n")
        for k in sorted(sq):
            a = sq[k]
            avg = {n: (a[n][0] / a[n][1] if n in a and a[n][1] else float("nan")) for n in names}
            disp = max(v[1] for v in a.values())
            conf = avg.get("SQ_LDS_BANK_CONFLICT", float("nan")) / avg["SQ_LDS_IDX_ACTIVE"] if avg.get("SQ_LDS_IDX_ACTIVE") else float("nan")
            ipw = avg.get("SQ_INSTS_VALU", float("nan")) / avg["SQ_WAVES"] if avg.get("SQ_WAVES") else float("nan")
            csv.writer(out).writerow([k, disp] + ["%.1f" % avg[n] for n in names] + ["%.4f" % conf, "%.1f" % ipw,
                                     "one --pmc pass of SQ counters; %s; lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE "
                                     "(extra LDS cycles per LDS-array cycle)" % cfg])
# which kernel sources, workload and texture these passes belong to (bench.py: roofline.traffic_stale)
try:
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    stamp = importlib.import_module("slam-sdvl_amd.stamp").kernel_source_stamp()
    wl, tex = None, None
    for name in ("prof_fetch.json", "prof_sq.json", "prof_kt.json"):
        pj = os.path.join(src, name)
        if os.path.exists(pj) and os.path.getsize(pj):
            try:
                c = json.loads(open(pj).read().strip().splitlines()[-1])["config"]
                wl, tex = c["workload"].split(":")[0], c.get("texture", "plane").split(":")[0]
                break
            except Exception:
                pass
    stamp.update({"workload": wl, "texture": tex})
    json.dump(stamp, open(os.path.join(dst, "source_stamp.json"), "w"), indent=1, sort_keys=True)
except Exception as e:
    print("source stamp not written:", e)
print("wrote", sorted(os.listdir(dst)))
