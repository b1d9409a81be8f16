#!/bin/bash
# CNN forward time over the tile-policy knobs (AVD_CNN_FILL percent, AVD_CNN_SHORTK half stages)
for f in 50 100 150 300 1000; do for k in 0 2 8 18 1000; do
  echo -n "FILL=$f SHORTK=$k  "; AVD_CNN_FILL=$f AVD_CNN_SHORTK=$k timeout -k 10 200 python tools/run_cnn.py 120 5 | tail -1
done; done
