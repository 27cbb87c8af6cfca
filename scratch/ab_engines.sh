#!/bin/bash
# like ab.sh, over exp_engines.py (two-chain / four-chain / speculative kernels at 256 chain pairs)
lib=literate_amd/csrc/libliterate_hip.so
for v in ${VARIANTS:-base new}; do
  cp scratch/ab/$v.so $lib
  echo "== $v"
  python scratch/exp_engines.py 2>&1 | grep -v amdgpu.ids | grep "C=512"
done
