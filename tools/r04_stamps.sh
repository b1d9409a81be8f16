#!/bin/bash
# round 4: barrier-wait stamps of the fast level kernel per role (debug build, -DAVD_FBF_DEBUG), one level size per run
cd $GRAFT_REPO_ROOT
cp ai-video-detector_amd/lib/libavd_hip.so /tmp/lib_release.so
make -C ai-video-detector_amd/csrc -B EXTRA=-DAVD_FBF_DEBUG > /dev/null 2>&1 || { echo build failed; exit 1; }
for w in ${@:-160 80 40 320}; do
  echo "== stamps at $w px"
  AVD_FBF_STAMPS=$w timeout -k 10 200 python bench.py --inflight 1 --cpu-frames 0 --steps 3 --warmup 1 --repeats 1 --no-pcie --no-vit --no-extras 2>&1 >/dev/null | grep "fbfast stamps"
done
cp /tmp/lib_release.so ai-video-detector_amd/lib/libavd_hip.so
