#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, no tracing domains) of the two matrix-core extensions:
# the CNN forward pass (per launch) and the patch-embed GEMM
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ext_pmc_*
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ext_pmc_fetch_cnn -- python3 $R/tools/run_cnn.py 120 1 > $R/gpurun_out/ext_pmc.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ext_pmc_write_cnn -- python3 $R/tools/run_cnn.py 120 1 >> $R/gpurun_out/ext_pmc.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ext_pmc_fetch_vit -- python3 $R/tools/run_vit.py 960 2 >> $R/gpurun_out/ext_pmc.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ext_pmc_write_vit -- python3 $R/tools/run_vit.py 960 2 >> $R/gpurun_out/ext_pmc.log 2>&1 || exit 1
cd $R
python tools/ext_pmc_summary.py 50 r04_pmc_extensions.json && cp gpurun_out/r04_pmc_extensions.json profiles/r04_pmc_extensions.json   # bench.py reads the CNN traffic from here
