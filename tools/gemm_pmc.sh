#!/bin/bash
# LDS bank conflicts and MFMA busy cycles of the patch-embed GEMM (separate --pmc pass, no tracing domains)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/gemm_pmc -- python3 $R/bench.py --cpu-frames 0 --repeats 1 --no-pcie --steps 2 --warmup 1 > $R/gpurun_out/gemm_pmc.log 2>&1
cd $R
python tools/pmc_summary.py k_gemm gpurun_out/gemm_pmc/*/*counter_collection.csv
