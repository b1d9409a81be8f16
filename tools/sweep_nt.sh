#!/bin/bash
# preprocess tuning sweep: workgroup size x rows per band (one clip at a time)
for nt in 256 512; do for r in $*; do
  out=$(AVD_PRE_NT=$nt AVD_ROWS_PER_BAND=$r timeout -k 10 120 python bench.py --inflight 1 --steps 8 --warmup 2 --cpu-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stages_ms']['preprocess'], d['roofline']['frac'])")
  echo "nt=$nt rows=$r -> $out"
done; done
