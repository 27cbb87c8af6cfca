#!/bin/bash
# A/B on ONE box: death-side gathers of the cached-birth path batched by 2 / 4 / 8 chains (scratch/ab/b2.so, b4.so, b8.so)
lib=literate_amd/csrc/libliterate_hip.so
for rep in 1 2; do
  for v in b4 b8 b2; do
    cp scratch/ab/$v.so $lib
    echo "== $v (rep $rep)"
    LR_EXP_C=8,16,256 python scratch/exp_loglik_big.py 2>&1 | grep -v amdgpu.ids
  done
done
