set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fast or detect or corner" > $O/t_fast.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_fast.log
bash tools/r4_ab.sh "SDVL_FAST_INT_SCORES=1" 3 40
rm -rf $O/pmc_un
rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_un -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-steps 0 --sustained-frames 0 > $O/pmc_un.json 2> $O/pmc_un.err; echo rc=$?
python3 - <<'PY'
import csv, glob, collections, re
f=glob.glob('gpurun_out/pmc_un/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void (anonymous namespace)::','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]['SQ_LDS_IDX_ACTIVE']):
    print('%-40s n=%5d unaligned=%10d addr_conf=%10d bank_conf=%12d idx_active=%12d insts_lds=%11d' % (k, n[k], v['SQ_LDS_UNALIGNED_STALL'], v['SQ_LDS_ADDR_CONFLICT'], v['SQ_LDS_BANK_CONFLICT'], v['SQ_LDS_IDX_ACTIVE'], v['SQ_INSTS_LDS']))
PY
