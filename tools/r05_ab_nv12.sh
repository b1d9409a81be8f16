#!/bin/bash
# round 5: the NV12 conversion as three LDS tables (T) against the arithmetic form (A = make -B OUT=../lib_ab/A.so EXTRA=-DAVD_NV12_ARITH; T = the default build copied to
# ../lib_ab/T.so), alternating on one box: profiles/r05_ab_nv12_tables.txt
cd "$(dirname "$0")/.."
L=ai-video-detector_amd/lib
for i in 1 2 3; do
  for v in A T; do
    cp ai-video-detector_amd/lib_ab/$v.so $L/libavd_hip.so
    timeout -k 10 300 python bench.py --cpu-frames 0 --no-vit --no-extras --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
e = d['extensions']['nv12_ingest']
print('$v  nv12 launch %.4f ms  frac %.4f   sec_per_video_nv12 %s  value %.0f' % (e['avg_launch_ms'], e['frac'], d['config'].get('sec_per_video_nv12'), d['value']))" || exit 1
  done
done
cp ai-video-detector_amd/lib_ab/T.so $L/libavd_hip.so
