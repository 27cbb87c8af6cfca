#!/bin/bash
# A/B on ONE box: unit-resolution scan without (u0.so) / with (u1.so) the wave-uniform birth entries in scalar registers:
# the launch-based engine on 16 chains x 1e7 / 3e7 / 1e8 lineages and cfg-sized pipelined engines
lib=literate_amd/csrc/libliterate_hip.so
for rep in 1 2; do
  for v in u0 u1; do
    cp scratch/ab/$v.so $lib
    echo "== $v (rep $rep)"
    LR_EXP_ENGINES=auto python scratch/exp_stream_engine.py 2>&1 | grep -v amdgpu.ids
  done
done
