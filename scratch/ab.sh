#!/bin/bash
# A/B of builds of the library on ONE box (boxes of the pool differ by ~5 %): scratch/ab/<name>.so for the names in
# $VARIANTS (default "base new"), alternating, exp_r2.py cases given as arguments.  Leaves the last variant in place.
lib=literate_amd/csrc/libliterate_hip.so
for rep in 1 2; do
  for v in ${VARIANTS:-base new}; do
    cp scratch/ab/$v.so $lib
    echo "== $v (rep $rep)"
    python scratch/exp_r2.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
