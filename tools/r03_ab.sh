#!/bin/bash
# A/B of AVD_FB_VARIANT values in one box: one clip at a time, resident; prints latency + stage times per variant
for v in "$@"; do
  AVD_FB_VARIANT=$v python bench.py --inflight 1 --steps 10 --warmup 3 --cpu-frames 0 --repeats 3 --no-pcie --no-vit --no-extras > gpurun_out/r03_ab_$v.json 2> gpurun_out/r03_ab_$v.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r03_ab_$v.json'))
print('variant $v', 'fps', round(d['value']), 'ms/step', d['ms_per_step'], 'stages', {k: v for k, v in d['stages_ms'].items() if k != 'note'})
PY
done
