#!/bin/bash
# A/B on ONE box: the unit-resolution scan with one (depth1.so) or two (depth2.so) 32-byte (ts, te) pairs in flight per thread
lib=literate_amd/csrc/libliterate_hip.so
for rep in 1 2; do
  for v in depth1 depth2; do
    cp scratch/ab/$v.so $lib
    echo "== $v (rep $rep)"
    python scratch/exp_stream_engine.py 2>&1 | grep -v amdgpu.ids
  done
done
