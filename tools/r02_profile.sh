#!/bin/bash
# round-2 evidence in one gpurun call: kernel-trace stats (clips alone / default 3 in flight), PMC passes (separate, no
# tracing domains, program directly after --), bench lines.  Results under gpurun_out/r02_*; summaries are copied to
# profiles/ by hand.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--steps 5 --warmup 2 --cpu-frames 0 --repeats 1 --no-pcie --no-vit"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_kt1 -- python3 $R/bench.py --inflight 1 $B > $R/gpurun_out/r02_kt1.log 2>&1 || exit 1
echo kt1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_kt3 -- python3 $R/bench.py $B > $R/gpurun_out/r02_kt3.log 2>&1 || exit 1
echo kt3 done
P="--inflight 1 --steps 2 --warmup 1 --cpu-frames 0 --repeats 1 --no-pcie --no-vit"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_fetch -- python3 $R/bench.py $P > $R/gpurun_out/r02_pmc_fetch.log 2>&1 || exit 1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_write -- python3 $R/bench.py $P > $R/gpurun_out/r02_pmc_write.log 2>&1 || exit 1
echo write done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/r02_pmc_sq -- python3 $R/bench.py $P > $R/gpurun_out/r02_pmc_sq.log 2>&1 || echo "sq pass failed"
echo sq done
cd $R
python tools/pmc_to_json.py gpurun_out/r02_pmc.json "round 2" gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write gpurun_out/r02_pmc_sq > gpurun_out/r02_pmc_summary.txt
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || exit 1
python bench.py --inflight 1 --cpu-frames 0 > gpurun_out/r02_bench_inflight1.json 2>> gpurun_out/r02_bench.err || exit 1
head -c 400 gpurun_out/r02_bench.json; echo
find gpurun_out/r02_kt1 -name "*kernel_stats.csv" | head -1 | xargs -I{} python tools/kstats.py {} 7 14
