set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
python -m pytest tests -x -q -m gpu -k "not other_forms" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_fast.log
python tools/kernel_bench.py 256 10 > $O/kb_v.log 2>&1; echo "rc=$? $(grep -E '^  fast_cells|^  search_points|^  search_prepare' $O/kb_v.log)"
rm -rf $O/kbpmc_q
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/kbpmc_q -- python3 tools/kernel_bench.py 64 2 > $O/kbpmc_q.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/kbpmc_q/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'search_p' not in k: continue
    agg[k[:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items(): print(k, round(v['SQ_INSTS_VALU']/v['SQ_WAVES'],1))
PY
