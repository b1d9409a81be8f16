#!/bin/bash
# round 3: kernel-trace stats of one clip at a time (tag = $1), per-kernel table on stdout
R=$GRAFT_REPO_ROOT; T=${1:-kt}
cd /tmp && export TMPDIR=/tmp
B="--steps 5 --warmup 2 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_$T -- python3 $R/bench.py --inflight 1 $B > $R/gpurun_out/r03_$T.log 2>&1 || exit 1
cd $R
find gpurun_out/r03_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} python tools/kstats.py {} 7 24
