#!/bin/bash
# timing-only ablation of k_polyexp_all (EXTRA=-DAVD_POLY_ABL=n: bit 1 one load per pixel, bit 2 float accumulators; results wrong)
for a in 0 1 2 3; do
  (cd ai-video-detector_amd/csrc && make EXTRA=-DAVD_POLY_ABL=$a -B > /dev/null 2>&1) || exit 1
  echo -n "AVD_POLY_ABL=$a  "; bash tools/kt_quick.sh 2>&1 | grep -E "k_polyexp_all" | awk '{print $(NF-1), "us"}'
done
(cd ai-video-detector_amd/csrc && make -B > /dev/null 2>&1)
