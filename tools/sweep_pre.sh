#!/bin/bash
# sweep preprocess tuning knobs; prints stage-0 time per setting
for pipe in 1; do for r in 8 10 12 14 16; do
  out=$(AVD_PRE_VARIANT=$pipe AVD_ROWS_PER_BAND=$r timeout -k 10 120 python bench.py --steps 6 --warmup 2 --cpu-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stages_ms']['preprocess'], d['roofline']['frac'])")
  echo "pipe=$pipe rows=$r -> $out"
done; done
