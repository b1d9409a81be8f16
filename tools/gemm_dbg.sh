#!/bin/bash
# timing-only ablation of the patch-embed GEMM (AVD_GEMM_DBG bits: 1 no global->LDS loads, 2 no MFMA, 4 no C stores, 8 no LDS fragment reads; library built with EXTRA=-DAVD_GEMM_DEBUG)
for d in ${@:-0 1 2 4 5 6 7}; do
  echo -n "AVD_GEMM_DBG=$d  "
  AVD_GEMM_DBG=$d timeout -k 10 300 python bench.py --cpu-frames 0 --repeats 1 --no-pcie --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mfma_patch_embed']; print('GEMM %.4f ms  (%.1f TFLOP/s equivalent)' % (m['avg_launch_ms'], m['achieved']))"
done
