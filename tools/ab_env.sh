#!/bin/bash
# A/B of an environment knob on ONE box, alternating: tools/ab_env.sh <rounds> VAR valA valB
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $1); do
  for v in $3 $4; do
    out=$(env $2=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-frames 0 2>/dev/null | tail -1)
    echo "$2=$v $(echo $out | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline_farneback']; print(d['ms_per_step'], d['stages_ms']['farneback_and_flow_stats'], r['k_uv_320']['avg_launch_ms'], r['k_hscan_320']['avg_launch_ms'])")"
  done
done
