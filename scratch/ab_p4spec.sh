#!/bin/bash
# A/B of the four-chain kernel with and without speculation on rejection (LR_P4_SPEC, latched at init: one process per
# variant), cfg4 and cfg4 model 3, from the initial state (WARM 300: the bench's regime) and after 3000 iterations
# (few accepted moves), helper scan share swept.   bash scratch/ab_p4spec.sh > gpurun_out/ab_p4spec.txt
for warm in 300 3000; do
  for rep in 1 2; do
    for v in 0 1; do
      echo "== LR_P4_SPEC=$v warm=$warm (rep $rep)"
      LR_EXP_WARM=$warm LR_P4_SPEC=$v python scratch/exp_r2.py "cfg4 100k x 1024" "cfg4 model3" 2>&1 | grep -v amdgpu.ids
    done
  done
done
for t in 0 1 2 3 5; do
  echo "== LR_P4_SPEC=1 LR_P4_HELP_TRIPS=$t"
  LR_P4_HELP_TRIPS=$t LR_P4_SPEC=1 python scratch/exp_r2.py "cfg4 100k x 1024" 2>&1 | grep -v amdgpu.ids
done
