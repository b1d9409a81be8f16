#!/bin/bash
# ViT patch-embed GEMM: tests, then TFLOP/s (bench.py's mfma_patch_embed key)
timeout -k 10 300 python -m pytest tests/test_vit.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for v in 0; do
timeout -k 10 300 python bench.py --cpu-frames 0 --repeats 1 --no-pcie --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mfma_patch_embed']; print('GEMM %.4f ms  %.1f TFLOP/s  %.1f %% of 2.5 PF (M=%d); with f32 tokens %.1f TFLOP/s' % (m['avg_launch_ms'], m['achieved'], 100*m['frac'], m['M'], m['achieved_with_f32_tokens']))" || exit 1
done
