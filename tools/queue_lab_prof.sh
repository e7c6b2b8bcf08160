#!/bin/bash
# per-launch durations of small_ops_kernel in eager sweeps (rocprofv3 kernel trace, csv)
set -e
R=$PWD
OUT=${1:-$R/gpurun_out/queue_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BAYESPY_AMD_GRAPH=${GRAPH:-0} QUEUE_LAB_SM=${SM:-1} rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o q -- python $R/tools/queue_lab.py > $OUT/run.log 2>&1 || true
tail -3 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob('$OUT/raw/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {}
NL = int('${NLAST:-400}')
for r in rows[-NL:]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = n.split('(')[0][:60]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    names.setdefault(n, []).append(d)
for n, d in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print('%-62s n=%4d sum=%9.1f us  avg=%7.2f  min=%7.2f max=%7.2f' % (n, len(d), sum(d), sum(d)/len(d), min(d), max(d)))
so = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'small_ops_kernel' in r['Kernel_Name']]
print('small_ops_kernel durations (last 30 launches, us):', ' '.join('%.1f' % v for v in so[-30:]))
# gaps between consecutive kernels in the last 150 launches
last = rows[-150:]
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(last, last[1:])]
print('median gap us', sorted(gaps)[len(gaps)//2], 'busy us', sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last), 'span us', (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3)
PY
