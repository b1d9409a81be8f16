#!/bin/bash
for skip in 0 2 4 16 15; do
  out=$(AVD_PRE_VARIANT=1 AVD_ROWS_PER_BAND=14 AVD_DBG_SKIP=$skip timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stages_ms']['preprocess'])")
  echo "skip=$skip -> $out"
done
