#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do python bench.py --cpu-frames 0 --no-vit --no-extras --no-pcie --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('events: preprocess %s us frac %s   value %.0f' % ([k['us'] for k in r['kernels'] if k['name']=='preprocess'], r.get('preprocess_frac'), d['value']))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prechk -- python3 $R/bench.py --inflight 1 --steps 5 --warmup 2 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras > /dev/null 2>&1
cd $R; ls -t $(find gpurun_out/r05_prechk -name "*kernel_stats.csv") | head -1 | xargs -I{} python tools/kstats.py {} 7 18 | grep -i "preprocess\|k_wake"
