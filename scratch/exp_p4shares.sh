#!/bin/bash
# four-chain kernel: scanner shares by wave age, pairs (2,3) (4,5) ... (14,15); values per 14 trips
for sh in "0,0,0,0,0,0,0" "5,2,2,0,0,-5,-4" "4,2,1,0,0,-4,-3" "7,3,2,0,-1,-6,-5" "3,1,1,0,0,-3,-2" "10,5,2,0,-3,-7,-7"; do
  echo "== LR_P4_SHARES=$sh"
  LR_P4_SHARES=$sh python scratch/exp_r2.py "cfg4 100k x 1024" "general 100k x 1024" 2>&1 | grep -v "amdgpu.ids\|p4general"
done
