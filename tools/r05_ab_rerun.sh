#!/bin/bash
# round 5: what the flag criteria cost on the bench clip -- AVD_FB_RERUN=1 (default: criteria evaluated, flag word read) against 0 (neither), alternating on ONE box
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for r in 1 0; do
    AVD_FB_RERUN=$r timeout -k 10 200 python bench.py --cpu-frames 0 --no-extras --no-vit --no-pcie --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('AVD_FB_RERUN=$r  value %.0f  ms_per_step %.4f  resident %.6f  level320 launch %.4f ms  pyramid %s us' % (d['value'], d['ms_per_step'], d['config']['sec_per_video_resident'], r['avg_launch_ms'], [k['us'] for k in r['kernels'] if k['name'] == 'pyramid']))"
  done
done
