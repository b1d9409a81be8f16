#!/bin/bash
# throughput as a function of the number of clips in flight per GPU (bench.py --inflight)
for k in ${@:-1 2 3 4 5 6}; do
timeout -k 10 300 python bench.py --inflight $k --cpu-frames 0 --repeats 7 --no-vit --no-pcie 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $k: %.0f frames/s (%.0f..%.0f)  %.3f ms/step' % (d['value'], d['repeats']['value_min'], d['repeats']['value_max'], d['ms_per_step']))" || exit 1
done
