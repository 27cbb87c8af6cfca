#!/bin/bash
# rebuild the library on the box with each flag set in FLAGSETS (separated by ';'), run the given exp_r2 cases
IFS=';' read -ra SETS <<< "$FLAGSETS"
for F in "${SETS[@]}"; do
  LR_EXTRA_FLAGS="$F" python -m literate_amd.build > /dev/null 2>&1 || { echo "build [$F] failed"; exit 1; }
  echo "== [$F]"
  python scratch/exp_r2.py "$@" 2>&1 | grep -v amdgpu.ids
done
