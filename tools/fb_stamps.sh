#!/bin/bash
# where do the waves of the fused Farneback kernel spend their time?  (library built with EXTRA=-DAVD_FB_DEBUG)
for d in ${@:-0 12}; do
  echo "== AVD_FB_DBG=$d"
  AVD_FB_STAMPS=1 AVD_FB_DBG=$d timeout -k 10 200 python bench.py --inflight 1 --cpu-frames 0 --steps 3 --warmup 1 --repeats 1 --no-pcie --no-vit 2>&1 >/dev/null | grep "fb stamps"
done
