#!/bin/bash
# round-5 evidence in one gpurun call: kernel-trace stats (default mode: clips alone / 3 in flight; exact mode alone), PMC passes
# (separate, no tracing domains, program directly after --), bench lines (driver defaults; inflight 1; 2 ranks on one GPU over gloo).
# Results under gpurun_out/r05_*; summaries are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--steps 5 --warmup 2 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_kt1 -- python3 $R/bench.py --inflight 1 $B > $R/gpurun_out/r05_kt1.log 2>&1 || exit 1
echo kt1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_kt3 -- python3 $R/bench.py $B > $R/gpurun_out/r05_kt3.log 2>&1 || exit 1
echo kt3 done
export AVD_FB_MODE=exact
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_kt1_exact -- python3 $R/bench.py --inflight 1 $B > $R/gpurun_out/r05_kt1_exact.log 2>&1 || exit 1
unset AVD_FB_MODE
echo kt1 exact done
P="--inflight 1 --steps 2 --warmup 1 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r05_pmc_fetch -- python3 $R/bench.py $P > $R/gpurun_out/r05_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r05_pmc_write -- python3 $R/bench.py $P > $R/gpurun_out/r05_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r05_pmc_sq -- python3 $R/bench.py $P > $R/gpurun_out/r05_pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/r05_pmc_ta -- python3 $R/bench.py $P > $R/gpurun_out/r05_pmc_ta.log 2>&1 || echo "ta pass failed"
echo pmc done
cd $R
python tools/pmc_to_json.py gpurun_out/r05_pmc.json "round 5 (default mode: fast level kernels + re-run)" gpurun_out/r05_pmc_fetch gpurun_out/r05_pmc_write gpurun_out/r05_pmc_sq gpurun_out/r05_pmc_ta > gpurun_out/r05_pmc_summary.txt
mkdir -p profiles && cp gpurun_out/r05_pmc.json profiles/r05_pmc.json     # bench.py reads the traffic figures and the busy fractions from here
python bench.py --details gpurun_out/r05_bench_details.json > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || exit 1
python bench.py --inflight 1 --cpu-frames 0 --no-extras --no-vit --details gpurun_out/r05_bench_inflight1_details.json > gpurun_out/r05_bench_inflight1.json 2>> gpurun_out/r05_bench.err || exit 1
AVD_BENCH_DEVICE=0 python bench.py --gpus 2 --backend gloo --cpu-frames 0 --no-extras --no-vit --no-pcie --repeats 3 > gpurun_out/r05_rehearsal_2ranks_1gpu_gloo.json 2> gpurun_out/r05_rehearsal.err || echo "rehearsal failed"
echo "stdout lines of the 2-rank rehearsal: $(wc -l < gpurun_out/r05_rehearsal_2ranks_1gpu_gloo.json)"
head -c 400 gpurun_out/r05_bench.json; echo
for t in kt1 kt3 kt1_exact; do echo "== $t"; ls -t $(find gpurun_out/r05_$t -name "*kernel_stats.csv") | head -1 | xargs -I{} python tools/kstats.py {} 7 18    # the NEWEST: gpurun_out/ accumulates earlier runs; done
