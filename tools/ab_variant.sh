#!/bin/bash
# A/B of fused-kernel variants in ONE gpurun call (same box): AVD_FB_VARIANT values as arguments
for v in "$@"; do for m in 1 3; do
  echo -n "variant=$v inflight=$m  "
  AVD_FB_VARIANT=$v timeout -k 10 200 python bench.py --inflight $m --cpu-frames 0 --steps 20 --repeats 5 --no-pcie --no-vit 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.0f frames/s (%.0f..%.0f)  %.3f ms/step  resident latency %.3f ms  level0 %.4f ms  farneback %.3f ms' % (d['value'], d['repeats']['value_min'], d['repeats']['value_max'], d['ms_per_step'], d['config']['sec_per_video_resident']*1e3, d['stages_ms']['level0_all_iterations'], d['stages_ms']['farneback_and_flow_stats']))"
done; done
