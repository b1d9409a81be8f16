#!/bin/bash
# per-launch kernel trace of the CNN forward pass (120 frames): which layers take the time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cnn_kt -- python3 $R/tools/run_cnn.py 120 2 > $R/gpurun_out/cnn_kt.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/cnn_kt/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_conv', 'k_slab', 'k_cnn', 'k_stem', 'k_maxpool', 'k_avgpool', 'k_linear'))]
per = len(rows) // 3
last = rows[-per:]
tot = 0
for i, r in enumerate(last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = r['Kernel_Name'].split('(')[0][-40:]
    print('%2d %-42s grid %6s lds %6s  %8.1f us' % (i, name, r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('LDS_Block_Size', '?'), d))
print('sum of kernel durations %.1f us' % tot)
PY
