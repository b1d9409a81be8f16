#!/bin/bash
# round 5: other latency shapes for the small levels, one clip alone, alternating on one box (profiles/r05_ab_latency_shapes_rejected.txt).
# ai-video-detector_amd/lib_ab/*.so are builds of avd_fbfast.hip with one geometry replaced (sed on the FGeo<...> of launch_fb_fast, make -B OUT=../lib_ab/X.so):
# G160a = FGeo<160, 2, 2, 4, 1> (12 waves, gather lead 2), G160b = FGeo<160, 2, 1, 4, 1>, G80b = FGeo<80, 2, 1, 4, 1>, G40a = FGeo<40, 1, 2, 4, 2>.  None is faster.
cd "$(dirname "$0")/.."
L=ai-video-detector_amd/lib
for i in 1 2; do
  for v in BASE G160a G160b G80b G40a; do
    cp ai-video-detector_amd/lib_ab/$v.so $L/libavd_hip.so
    timeout -k 10 200 python bench.py --inflight 1 --cpu-frames 0 --no-extras --no-vit --no-pcie --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
k = {x['name']: x['us'] for x in r['kernels']}
print('$v  alone fps %.0f  resident %.6f  levels 40/80/160/320 us: %s %s %s %s' % (d['value'], d['config']['sec_per_video_resident'], k.get('level40'), k.get('level80'), k.get('level160'), k.get('level320')))" || exit 1
  done
done
cp ai-video-detector_amd/lib_ab/BASE.so $L/libavd_hip.so
