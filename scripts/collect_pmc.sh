#!/bin/bash
# Run ON THE GPU BOX (gpurun): per-kernel PMC counters of the benchmark step, one rocprofv3 pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC has 4 slots, they cost 3 + 2).  Output: gpurun_out/pmc_*/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $B > gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d gpurun_out/pmc_sq -- $B > gpurun_out/pmc_sq.log 2>&1
ls gpurun_out/pmc_fetch/*/ | head -3
