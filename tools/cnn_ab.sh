#!/bin/bash
# CNN forward time per tile policy (AVD_CNN_TILES: 0 = heuristic, 1 = always 256-pixel tiles, 2 = 128 x 128 wherever possible) and batch size
for t in ${@:-0 1 2}; do
  for n in 120 32; do
    echo -n "AVD_CNN_TILES=$t n=$n  "; AVD_CNN_TILES=$t timeout -k 10 200 python tools/run_cnn.py $n 5 | tail -1
  done
done
