#!/bin/bash
# PMC passes for the preprocess kernel alone (tools/run_preprocess.py): separate runs, no tracing domains
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prepmc_$i -- python3 $R/tools/run_preprocess.py > /dev/null 2>&1
  echo "pass $i rc=$? : $set"
done
