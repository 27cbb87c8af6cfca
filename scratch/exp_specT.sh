#!/bin/bash
# spec kernel block size sweep: rebuild the library per size on the box, run the given exp_r2 cases
for T in ${SPEC_T:-768 512}; do
  LR_EXTRA_FLAGS=-DLR_SPEC_THREADS=$T python -m literate_amd.build > /dev/null 2>&1 || { echo "build $T failed"; exit 1; }
  echo "== LR_SPEC_THREADS=$T"
  python scratch/exp_r2.py "$@" 2>&1 | grep -v amdgpu.ids
done
