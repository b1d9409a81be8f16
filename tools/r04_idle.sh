#!/bin/bash
# how much of the timed region has NO kernel running (three clips in flight)?  kernel trace of a longer run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_idle -- python3 $R/bench.py --steps 40 --warmup 3 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras > $R/gpurun_out/r04_idle.log 2>&1 || exit 1
cd $R
python3 tools/r04_idle.py gpurun_out/r04_idle
