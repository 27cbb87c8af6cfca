#!/bin/bash
# cfg4, helper form: scanner shares by AGE of the waves on their SIMD (the youngest - waves 12..15 - end their scans 0.5 us
# after the oldest: oldest-first issue arbitration).  LR_P4_SHARES12 = trips per 9 for the pairs (4,5) (6,7) (8,9) (10,11) (12,13) (14,15)
for s in "0,0,0,0,0,0" "1,1,0,0,-1,-1" "1,0,0,0,0,-1" "0,1,0,0,-1,0" "1,1,1,1,-2,-2" "0,0,1,1,-1,-1" "0,0,0,0,0,0" "2,2,0,0,-2,-2"; do
  echo "== LR_P4_SHARES12=$s"
  LR_EXP_WARM=3000 LR_P4_SHARES12=$s python scratch/exp_r2.py "cfg4 100k x 1024" 2>&1 | grep -v amdgpu.ids
done
