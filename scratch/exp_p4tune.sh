#!/bin/bash
# four-chain kernel on the group-packed scan: unroll x scanner-wave shares (cfg4 unit and general)
for U in 1 2; do
  LR_EXTRA_FLAGS="-DLR_P4_UNROLL_U=$U" python -m literate_amd.build > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for SH in "0,0,0,0,0,0,0" "3,3,1,0,-1,-3,-3" "6,6,2,0,-2,-6,-6" "8,8,3,0,-3,-8,-8" "10,8,4,0,-4,-8,-10"; do
    echo "== unroll $U shares $SH"
    LR_P4_SHARES=$SH python scratch/exp_r2.py "cfg4 100k x 1024" "cfg4-general 100k x 1024" 2>&1 | grep -v "amdgpu.ids\|p4general"
  done
done
