#!/bin/bash
# kernel trace of the in-flight loop with one flagged stripe pair: where does the chip wait?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_iftrace
rm -rf $O && mkdir -p $O
for P in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/p$P -- python3 $R/tools/r05_inflight_trace.py $P 3 24 > $O/run_p$P.log 2>&1 || exit 1
  tail -1 $O/run_p$P.log
  python3 $R/tools/r04_idle.py $O/p$P > $O/idle_p$P.txt 2>&1
  cat $O/idle_p$P.txt
  F=$(find $O/p$P -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/timeline.py $F 400 > $O/timeline_p$P.txt
  rm -rf $O/p$P
done
