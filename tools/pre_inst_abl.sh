#!/bin/bash
# instruction counts of the preprocess kernel per ablated phase (timing-only build, results WRONG for masks != 0):
# 1 no Laplacian phase, 2 no INTER_AREA phase, 4 no INTER_LINEAR phase, 8 no gray arithmetic, 16 no loads
R=$GRAFT_REPO_ROOT
make -C $R/ai-video-detector_amd/csrc -B EXTRA="-w -DAVD_TIMING_EXPERIMENTS" > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for skip in 0 1 2 4 8 7 15; do
  AVD_DBG_SKIP=$skip timeout -k 10 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pia_$skip -- python3 $R/tools/run_preprocess.py > /dev/null 2>&1
  echo "skip=$skip rc=$?"
done
make -C $R/ai-video-detector_amd/csrc -B EXTRA="-w" > /dev/null 2>&1
