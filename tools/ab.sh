#!/bin/bash
# A/B of two builds on ONE box, alternating: tools/ab.sh <rounds> -- prints ms_per_step and farneback stage per run
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $1); do
  for v in old new; do
    cp tools/libavd_$v.so ai-video-detector_amd/lib/libavd_hip.so
    out=$(timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-frames 0 2>/dev/null | tail -1)
    echo "$v $(echo $out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms']['farneback_and_flow_stats'])")"
  done
done
