#!/bin/bash
# PMC passes over ONE launch of the speculative kernel (scratch/prof_spec.py N C): bash scratch/pmc_spec.sh <outdir> N C
export TMPDIR=/tmp
out=$1; mkdir -p $out
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$tag -- python3 scratch/prof_spec.py $2 $3 > $out/$tag.log 2>&1 < /dev/null || echo "$tag pass failed"
done
python3 scratch/pmc_summary.py $out < /dev/null
