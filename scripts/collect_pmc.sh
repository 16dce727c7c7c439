#!/bin/bash
# Run ON THE GPU BOX (gpurun): per-kernel PMC counters of the benchmark step, one rocprofv3 pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC has 4 slots, they cost 3 + 2).  Output: gpurun_out/pmc_*/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# SUFFIX / EXTRA select a second workload, e.g. SUFFIX=_bf16 EXTRA="--dtype bf16" (pmc_summary.py merges kernels the fp32 run does not have)
S=${SUFFIX:-}
B="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline ${EXTRA:-}"
rm -rf gpurun_out/pmc_fetch$S gpurun_out/pmc_write$S gpurun_out/pmc_sq$S
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch$S -- $B > gpurun_out/pmc_fetch$S.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write$S -- $B > gpurun_out/pmc_write$S.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d gpurun_out/pmc_sq$S -- $B > gpurun_out/pmc_sq$S.log 2>&1
ls gpurun_out/pmc_fetch$S/*/ | head -3
