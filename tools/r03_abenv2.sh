#!/bin/bash
# A/B of one environment variable ($1) over values ($2...) in one box: one clip at a time AND the default clips in flight
N=$1; shift
for v in "$@"; do
  for infl in 1 3; do
  env $N=$v python bench.py --inflight $infl --steps 10 --warmup 3 --cpu-frames 0 --repeats 5 --no-pcie --no-vit --no-extras > gpurun_out/r03_abenv.json 2> gpurun_out/r03_abenv.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r03_abenv.json'))
print('$N=$v inflight $infl', 'fps', round(d['value']), 'ms/step', d['ms_per_step'], 'level0', d['stages_ms'].get('level0_all_iterations'), 'fb', d['stages_ms'].get('farneback_and_flow_stats'))
PY
  done
done
