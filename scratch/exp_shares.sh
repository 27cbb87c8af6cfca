#!/bin/bash
# cfg4 under the four-chain kernel's fourteen-scanner form with unequal scanner shares (LR_P4_SHARES: wave pairs (2,3) ..
# (14,15), per 14 trips)
for s in "0,0,0,0,0,0,0" "0,0,2,0,0,-2,0" "0,0,2,0,2,-4,0" "0,0,4,0,0,-2,-2" "0,-2,4,-2,2,-2,0" "0,0,0,0,0,0,0" "-2,0,2,0,2,-2,0"; do
  echo "== shares $s"
  LR_P4_HELP=0 LR_P4_SHARES=$s LR_EXP_WARM=3000 python scratch/exp_r2.py "cfg4 100k x 1024" 2>&1 | grep -v amdgpu.ids
done
