#!/bin/bash
# A/B on ONE box: builds scratch/ab/draw0.so (draws ahead by waves 4, 5: round 4) and scratch/ab/new.so (by waves 6, 7), each with
# and without speculation on rejection (LR_P4_SPEC, latched at init), cfg4 and cfg4 model 3 from the initial state (WARM 300:
# the bench's regime) and after 3000 iterations, then the helper scan share swept.
lib=literate_amd/csrc/libliterate_hip.so
for warm in 300 3000; do
  for rep in 1 2; do
    for b in draw0 new; do
      cp scratch/ab/$b.so $lib
      for v in 0 1; do
        echo "== build=$b LR_P4_SPEC=$v warm=$warm (rep $rep)"
        LR_EXP_WARM=$warm LR_P4_SPEC=$v python scratch/exp_r2.py "cfg4 100k x 1024" "cfg4 model3" 2>&1 | grep -v amdgpu.ids
      done
    done
  done
done
cp scratch/ab/new.so $lib
for v in 0 1; do
  for t in 0 1 2 3 4 5; do
    echo "== build=new LR_P4_SPEC=$v LR_P4_HELP_TRIPS=$t warm=3000"
    LR_EXP_WARM=3000 LR_P4_HELP_TRIPS=$t LR_P4_SPEC=$v python scratch/exp_r2.py "cfg4 100k x 1024" 2>&1 | grep -v amdgpu.ids
  done
done
