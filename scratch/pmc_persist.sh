#!/bin/bash
# The round's profile set (run on the MI355X box: bash scratch/pmc_persist.sh gpurun_out/<dir>):
#   1. rocprofv3 --kernel-trace --stats over the bench command (kernel durations to compare with bench.py's HIP events)
#   2. one --pmc pass per counter group over scratch/prof_persist.py (exactly one 1000-iteration engine launch)
export TMPDIR=/tmp
out=$1; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --no-pmc > $out/stats_bench.json 2> $out/stats.err < /dev/null || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$tag -- python3 scratch/prof_persist.py > $out/$tag.log 2>&1 < /dev/null || echo "$tag pass failed"
done
python3 scratch/pmc_summary.py $out < /dev/null
