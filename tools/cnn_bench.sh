#!/bin/bash
# CNN extension: tests, then the forward pass of the bench clip (bench.py's mfma_cnn_forward key)
timeout -k 10 600 python -m pytest tests/test_cnn.py -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 400 python bench.py --cpu-frames 0 --repeats 1 --no-pcie --steps 3 2>gpurun_out/cnn_bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mfma_cnn_forward']; print('CNN forward %.3f ms for %d frames  %.1f TFLOP/s  %.1f %% of 2.5 PF  %.0f frames/s' % (m['forward_ms'], m['frames_per_forward'], m['achieved'], 100*m['frac'], m['frames_per_s']))"
