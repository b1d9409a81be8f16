#!/bin/bash
# NV12 ingest: tests, then the kernel time and the PCIe-inclusive rate
timeout -k 10 300 python -m pytest tests/test_nv12.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for g in 0; do
timeout -k 10 300 python bench.py --cpu-frames 0 --repeats 1 --no-vit --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['roofline_nv12_ingest']; print('NV12 ingest: %.4f ms  %.0f GB/s  %.1f %% of HBM peak  PCIe-inclusive %.0f frames/s' % (m['avg_launch_ms'], m['achieved'], 100*m['frac'], m['pcie_inclusive_fps_one_clip_at_a_time']))" || exit 1
done
