#!/bin/bash
# round 3: memory-path counters of the level kernels (separate passes); tag = $1
R=$GRAFT_REPO_ROOT; T=${1:-m}
cd /tmp && export TMPDIR=/tmp
P="--inflight 1 --steps 2 --warmup 1 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras"
rocprofv3 -L > $R/gpurun_out/r03_counters_list.txt 2>&1
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUSY_max TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r03_${T}_m$i -- python3 $R/bench.py $P > $R/gpurun_out/r03_${T}_m$i.log 2>&1 || echo "pass $i failed: $set"
done
cd $R
python tools/pmc_to_json.py gpurun_out/r03_${T}_mem.json "round 3 memory path" gpurun_out/r03_${T}_m1 gpurun_out/r03_${T}_m2 gpurun_out/r03_${T}_m3 gpurun_out/r03_${T}_m4 | grep -E "k_fb_fast<320|k_fb_fast<160|k_preprocess_vec"
