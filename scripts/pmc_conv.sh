#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 scripts/conv_bench.py --layers conv_4_2 --tiles 128128,1128128,2128128 --iters 3"
rm -rf gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmcA -- $B > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmcB -- $B > gpurun_out/pmcB.log 2>&1
tail -2 gpurun_out/pmcB.log
