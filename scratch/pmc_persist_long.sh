#!/bin/bash
# SQ counters over ONE 4000-iteration launch of the four-chain kernel (the chains reach their working number of shifts after
# ~500 iterations: close to the steady state bench.py times): bash scratch/pmc_persist_long.sh gpurun_out/<dir>
export TMPDIR=/tmp LR_PROF_ITERS=4000
out=$1; mkdir -p $out
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  tag=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$tag -- python3 scratch/prof_persist.py > $out/$tag.log 2>&1 < /dev/null || echo "$tag pass failed"
done
python3 scratch/pmc_summary.py $out < /dev/null
