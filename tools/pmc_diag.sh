#!/bin/bash
# diagnostic PMC passes (separate runs, no tracing domains, small sets) over the bench workload
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/diag_$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 > $R/gpurun_out/diag_$i.log 2>&1
  echo "pass $i rc=$? : $set" | tee -a $R/gpurun_out/diag_progress.txt
done
