#!/bin/bash
# sustained shader clock / power while the cfg4 workload runs: rocm-smi sampled beside a ~25 s run
python scratch/soak.py > gpurun_out/soak_probe.log 2>&1 < /dev/null &
PID=$!
sleep 8
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | head -6
  echo "--"
  sleep 2
done
wait $PID
tail -2 gpurun_out/soak_probe.log
