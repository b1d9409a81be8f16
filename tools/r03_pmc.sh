#!/bin/bash
# round 3: PMC passes (separate, no tracing domains, program directly after --); tag = $1
R=$GRAFT_REPO_ROOT; T=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
P="--inflight 1 --steps 2 --warmup 1 --cpu-frames 0 --repeats 1 --no-pcie --no-vit --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03_${T}_fetch -- python3 $R/bench.py $P > $R/gpurun_out/r03_${T}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03_${T}_write -- python3 $R/bench.py $P > $R/gpurun_out/r03_${T}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/r03_${T}_sq -- python3 $R/bench.py $P > $R/gpurun_out/r03_${T}_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r03_${T}_sq2 -- python3 $R/bench.py $P > $R/gpurun_out/r03_${T}_sq2.log 2>&1 || echo "sq2 pass failed"
cd $R
python tools/pmc_to_json.py gpurun_out/r03_${T}.json "round 3" gpurun_out/r03_${T}_fetch gpurun_out/r03_${T}_write gpurun_out/r03_${T}_sq gpurun_out/r03_${T}_sq2 > gpurun_out/r03_${T}_summary.txt
cat gpurun_out/r03_${T}_summary.txt | head -60
