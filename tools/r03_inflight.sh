#!/bin/bash
# clips in flight sweep (resident frames), fast mode
for m in "$@"; do
  python bench.py --inflight $m --steps 24 --warmup 4 --cpu-frames 0 --repeats 5 --no-pcie --no-vit --no-extras > gpurun_out/r03_if.json 2> gpurun_out/r03_if.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r03_if.json'))
print('inflight $m', 'fps', round(d['value']), 'ms/step', d['ms_per_step'])
PY
done
