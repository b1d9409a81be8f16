#!/bin/bash
# round 5: timing-only ablation of the fast level kernel on the round-5 build (results of the ablated builds are wrong): what do the bilinear gathers cost today?
cd "$(dirname "$0")/.."
cp ai-video-detector_amd/lib/libavd_hip.so /tmp/libavd_release.so
for flags in "" "-DAVD_FBF_NOGATHER" "-DAVD_FBF_NOGATHER -DAVD_FBF_NOINLOAD" "-DAVD_FBF_NOSOLVE"; do
  make -C ai-video-detector_amd/csrc -B EXTRA="$flags" > /dev/null 2>&1 || { echo build failed; exit 1; }
  AVD_FB_RERUN=0 python bench.py --inflight 1 --steps 10 --warmup 3 --cpu-frames 0 --repeats 3 --no-pcie --no-vit --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('flags [$flags]  ms/step %.4f  level320 launch %.4f ms  kernels' % (d['ms_per_step'], r['avg_launch_ms']), {k['name']: k['us'] for k in r['kernels'] if k['name'].startswith('level')})"
done
cp /tmp/libavd_release.so ai-video-detector_amd/lib/libavd_hip.so
