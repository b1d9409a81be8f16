#!/bin/bash
# round 5: timing-only ablation of the patch-embed GEMM in both wave shapes (library rebuilt with -DAVD_GEMM_DEBUG into a scratch copy)
# AVD_GEMM_DBG bits: 1 no global->LDS loads, 2 no MFMA, 4 no C stores, 8 no LDS fragment reads
cd "$(dirname "$0")/.."
cp ai-video-detector_amd/lib/libavd_hip.so /tmp/libavd_release.so
make -C ai-video-detector_amd/csrc EXTRA=-DAVD_GEMM_DEBUG -B > /dev/null 2>&1 || exit 1
for w in 8 16; do
  for d in 0 16 17 24 1 2; do
    echo -n "waves $w  AVD_GEMM_DBG=$d  "
    AVD_GEMM_WAVES=$w AVD_GEMM_DBG=$d timeout -k 10 120 python tools/run_vit.py 960 20 2>&1 | tail -1
  done
done
cp /tmp/libavd_release.so ai-video-detector_amd/lib/libavd_hip.so
