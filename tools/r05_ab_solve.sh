#!/bin/bash
# round 5: the fast level kernels' solve with the scale folded out (14 instead of 28 double operations per column, 19 instead of 21 additions per four windows)
# and the latency shapes' input loads with a scalar row offset (no spills in the 80-px kernel), each against the kernels before, alternating on ONE box.
# ai-video-detector_amd/lib_ab/{A,B,C,D}.so (make -C ai-video-detector_amd/csrc -B OUT=../lib_ab/X.so EXTRA="..."): A = before (-DAVD_FBF_SOLVE_R4 -DAVD_FBF_NO_SCALAR_ROW), B = both, C = solve only, D = loads only.
cd "$(dirname "$0")/.."
L=ai-video-detector_amd/lib
for i in 1 2 3; do
  for v in A B C D; do
    cp ai-video-detector_amd/lib_ab/$v.so $L/libavd_hip.so
    timeout -k 10 200 python bench.py --cpu-frames 0 --no-extras --no-vit --no-pcie --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
k = {x['name']: x['us'] for x in r['kernels']}
print('$v  value %.0f  ms_per_step %.4f  resident %.6f  level320 launch %.4f ms  levels 40/80/160/320 us: %s %s %s %s' % (d['value'], d['ms_per_step'], d['config']['sec_per_video_resident'], r['avg_launch_ms'], k.get('level40'), k.get('level80'), k.get('level160'), k.get('level320')))" || exit 1
  done
done
cp ai-video-detector_amd/lib_ab/B.so $L/libavd_hip.so
