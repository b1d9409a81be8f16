#!/bin/bash
# round 5: patch-embed GEMM, 8 waves (8 x 4 MFMA tiles per wave) vs 16 waves (4 x 4) with / without the half-step skew; tests first
cd "$(dirname "$0")/.."
make -C ai-video-detector_amd/csrc > /dev/null 2>&1
for w in 8 9 16 18; do
  echo "== AVD_GEMM_WAVES=$w"
  AVD_GEMM_WAVES=$w timeout -k 10 300 python -m pytest tests/test_vit.py -m gpu -x -q 2>&1 | tail -2
  for r in 1 2 3; do AVD_GEMM_WAVES=$w timeout -k 10 120 python tools/run_vit.py 960 20 2>&1 | tail -1; done
done
