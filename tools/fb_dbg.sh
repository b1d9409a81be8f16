#!/bin/bash
# timing-only ablation of the fused Farneback level kernel; needs a library built with EXTRA=-DAVD_FB_DEBUG
# (AVD_FB_DBG bits: 1 no scan, 2 no solve, 4 no refill loads of the vertical waves, 8 no touch loads)
for d in ${@:-0 8 1 2 3 4 12 5 7 15}; do
  echo -n "AVD_FB_DBG=$d  "
  AVD_FB_DBG=$d timeout -k 10 200 python bench.py --inflight 1 --cpu-frames 0 --steps 10 --repeats 1 --no-pcie --no-vit 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('level0 x3 iterations: %.4f ms   farneback stage %.4f ms' % (d['stages_ms']['level0_all_iterations'], d['stages_ms']['farneback_and_flow_stats']))"
done
