#!/bin/bash
# quick per-kernel stats of one clip at a time (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktq
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktq -- python3 $R/bench.py --inflight 1 --steps 5 --warmup 2 --cpu-frames 0 --repeats 1 --no-pcie --no-vit > $R/gpurun_out/ktq.log 2>&1 || exit 1
cd $R
find gpurun_out/ktq -name "*kernel_stats.csv" | head -1 | xargs -I{} python tools/kstats.py {} 7 24
