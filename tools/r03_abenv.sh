#!/bin/bash
# A/B of one environment variable ($1) over values ($2...) in one box: one clip at a time, resident
N=$1; shift
for v in "$@"; do
  env $N=$v python bench.py --inflight 1 --steps 10 --warmup 3 --cpu-frames 0 --repeats 3 --no-pcie --no-vit --no-extras > gpurun_out/r03_abenv.json 2> gpurun_out/r03_abenv.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r03_abenv.json'))
print('$N=$v', 'fps', round(d['value']), 'ms/step', d['ms_per_step'], 'stages', {k: v for k, v in d['stages_ms'].items() if k != 'note'})
PY
done
