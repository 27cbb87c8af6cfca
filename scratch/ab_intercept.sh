#!/bin/bash
# A/B of two builds (scratch/ab/base.so, new.so) on ONE box: per-launch intercept of the four-chain kernel and the bench
# line at the driver's arguments
lib=literate_amd/csrc/libliterate_hip.so
for rep in 1 2; do
  for v in base new; do
    cp scratch/ab/$v.so $lib
    echo "== $v (rep $rep)"
    python scratch/exp_intercept.py 2>&1 | grep -v amdgpu.ids | grep "n=   1\|n=   4\|n=  20\|n= 512"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-abi --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench20: value %.4g  ms_per_step %.5f  wall/device %.3f  us_per_iter_device %.3f' % (b['value'], b['ms_per_step'], b['config']['wall_over_device'], b['roofline']['us_per_iter_device']))"
  done
done
cp scratch/ab/new.so $lib
