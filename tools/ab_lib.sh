#!/bin/bash
# A/B of two builds on ONE box, alternating: tools/ab_lib.sh <rounds> <libA.so> <libB.so> [bench flags]
cd $GRAFT_REPO_ROOT
r=$1; a=$2; b=$3; shift 3
for i in $(seq 1 $r); do
  for v in $a $b; do
    cp $v ai-video-detector_amd/lib/libavd_hip.so
    out=$(timeout -k 10 120 python bench.py --cpu-frames 0 $* 2>/dev/null | tail -1)
    echo "$v $(echo $out | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline_farneback']; print(d['value'], d['ms_per_step'], d['config']['sec_per_video'], r['k_uv_320']['avg_launch_ms'], r['k_hscan_320']['avg_launch_ms'])")"
  done
done
