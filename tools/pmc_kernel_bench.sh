# bash tools/pmc_kernel_bench.sh [kernel-name-pattern]: SQ counters of tools/kernel_bench.py's launches (a pass of their own), one row per counter
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/kb_pmc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/kb_pmc -- python3 tools/kernel_bench.py 256 3 > gpurun_out/kb_pmc.txt 2>&1
find gpurun_out/kb_pmc -name "*kernel_trace.csv" -delete
python3 - "$@" <<'PY'
import csv, glob, re, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/kb_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        if pat in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[name]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, c in sorted(acc.items()):
    w = sum(c["SQ_WAVES"]) / max(1, len(c["SQ_WAVES"]))
    print(name, "launches", len(c["SQ_WAVES"]), "us %.1f" % (sum(c["us"]) / len(c["us"])), "waves %d" % w,
          " ".join("%s/wave %.1f" % (k[3:], sum(v) / len(v) / max(1, w)) for k, v in sorted(c.items()) if k.startswith("SQ_") and k != "SQ_WAVES"))
PY
