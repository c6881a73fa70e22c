set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py -x -q -m gpu > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_fast.log
python tools/kernel_bench.py 256 10 > $O/kb_v.log 2>&1; echo "rc=$? $(grep -E '^  fast_cells' $O/kb_v.log)"
rm -rf $O/kbpmc_q
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/kbpmc_q -- python3 tools/kernel_bench.py 64 2 > $O/kbpmc_q.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/kbpmc_q/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'fast_cells' not in k: continue
    agg[k[:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items(): print(k, {c: round(x/v['SQ_WAVES'],1) for c,x in v.items()})
PY
bash tools/r4_ab.sh "SDVL_FAST_PAIRS=1" 3 40
