#!/bin/bash
# wall / device of bench.py's timed region at the driver's arguments (--steps 20 --warmup 5) under the runtime's two ways of
# waiting for a completion signal: interrupts (default) and polling (HSA_ENABLE_INTERRUPT=0)
for rep in 1 2 3; do
  for v in default 0; do
    if [ $v = default ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=$v; fi
    echo "== HSA_ENABLE_INTERRUPT=$v (rep $rep)"
    LR_BENCH_DETAIL=/tmp/d.json python bench.py --steps 20 --warmup 5 --no-configs --no-pmc --no-abi 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3e  ms_per_step %.5f  wall_over_device %s' % (d['value'], d['ms_per_step'], d['config'].get('wall_over_device')))"
  done
done
