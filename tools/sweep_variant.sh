#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in $*; do
  (cd /tmp && TMPDIR=/tmp AVD_UV_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/var_$v.log 2>&1) || exit 1
  echo variant=$v done
done
