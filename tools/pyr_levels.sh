#!/bin/bash
# per-launch durations of pyr_down_kernel by grid size (one grid size per pyramid level) from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl && rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pl/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Kernel_Name'].startswith('(anonymous namespace)::pyr_down') or 'pyr_down' in r['Kernel_Name']:
        d[(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, 'n=%d median %.1f us  min %.1f' % (len(v), v[len(v)//2], v[0]))
PY
