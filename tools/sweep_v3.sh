#!/bin/bash
for v in 3; do for r in 4 5 6 7 8; do for skip in 0; do
  out=$(AVD_PRE_VARIANT=$v AVD_ROWS_PER_BAND=$r AVD_DBG_SKIP=$skip timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stages_ms']['preprocess'])")
  echo "variant=$v rows=$r skip=$skip -> $out"
done; done; done
