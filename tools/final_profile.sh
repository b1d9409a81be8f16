#!/bin/bash
# final per-round evidence: kernel trace stats (clips run alone = the durations bench.py's roofline uses; and the
# default serving mode with 3 clips in flight), PMC passes (separate, no tracing domains), bench line
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_kt1 -- python3 $R/bench.py --inflight 1 --steps 5 --warmup 2 --cpu-frames 0 > $R/gpurun_out/final_kt1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_kt -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-frames 0 > $R/gpurun_out/final_kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final_pmc_fetch -- python3 $R/bench.py --inflight 1 --steps 2 --warmup 1 --cpu-frames 0 > $R/gpurun_out/final_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final_pmc_write -- python3 $R/bench.py --inflight 1 --steps 2 --warmup 1 --cpu-frames 0 > $R/gpurun_out/final_pmc_write.log 2>&1 || exit 1
cd $R && python bench.py --pcie > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
python bench.py --inflight 1 --cpu-frames 0 > gpurun_out/final_bench_inflight1.json 2>> gpurun_out/final_bench.err || exit 1
head -c 250 gpurun_out/final_bench.json
