#!/bin/bash
# CNN extension: does a larger batch per forward pass (AVD_CNN_CHUNK) lift the late stages?  forward time at 120 / 480 / 960 frames,
# then the per-launch trace at 480
R=$GRAFT_REPO_ROOT
for n in 120 240 480 960; do
  AVD_CNN_CHUNK=$n timeout -k 10 300 python3 tools/run_cnn.py $n 3 || exit 1
done
cd /tmp && export TMPDIR=/tmp
export AVD_CNN_CHUNK=480
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cnn_kt480 -- python3 $R/tools/run_cnn.py 480 2 > $R/gpurun_out/cnn_kt480.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/cnn_kt480/*/*kernel_trace.csv')[0]
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
