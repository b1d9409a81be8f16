#!/bin/bash
# L1 (TCP) -> L2 read requests and L1 accesses of the level kernels (separate --pmc pass, no tracing domains)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tcp_pmc
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $R/gpurun_out/tcp_pmc -- python3 $R/bench.py --inflight 1 --steps 2 --warmup 1 --cpu-frames 0 --repeats 1 --no-pcie --no-vit > $R/gpurun_out/tcp_pmc.log 2>&1 || { tail -5 $R/gpurun_out/tcp_pmc.log; exit 1; }
cd $R
python tools/pmc_summary.py k_fb_level gpurun_out/tcp_pmc/*/*counter_collection.csv
python tools/pmc_summary.py k_polyexp gpurun_out/tcp_pmc/*/*counter_collection.csv
