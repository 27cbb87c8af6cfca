#!/bin/bash
# cfg4 under unequal scanner shares of the four-chain kernel's HELPER form: LR_P4_SHARES12 = trips (per 9) that the wave pairs
# (4,5) (6,7) (8,9) (10,11) (12,13) (14,15) scan beyond / short of the equal share - pairs (4,5), (8,9), (12,13) sit on the
# steppers' SIMDs, the others beside the helpers.   bash scratch/exp_shares12.sh > gpurun_out/shares12.txt
for s in "0,0,0,0,0,0" "-1,1,0,0,0,0" "-1,1,0,1,-1,0" "-2,2,0,1,-1,0" "-1,1,-1,1,-1,1" "-2,1,0,1,0,0" "0,0,0,0,0,0" "-1,0,0,1,0,0" "-2,1,-1,1,0,1" "1,-1,0,0,0,0"; do
  echo "== LR_P4_SHARES12=$s"
  LR_EXP_WARM=3000 LR_P4_SHARES12=$s python scratch/exp_r2.py "cfg4 100k x 1024" 2>&1 | grep -v amdgpu.ids
done
