#!/bin/bash
# round 5: the R1 window in LDS (AVD_FB_WIN=1) against gathers from memory (0), alternating on ONE box: bench clip, default flags otherwise
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for r in 1 0; do
    AVD_FB_WIN=$r timeout -k 10 200 python bench.py --cpu-frames 0 --no-extras --no-vit --no-pcie --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('AVD_FB_WIN=$r  value %.0f  ms_per_step %.4f  resident %.6f  level320 launch %.4f ms (events, mean of the three)' % (d['value'], d['ms_per_step'], d['config']['sec_per_video_resident'], r['avg_launch_ms']))"
  done
done
