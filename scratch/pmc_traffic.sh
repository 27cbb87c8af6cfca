#!/bin/bash
# separate PMC passes (FETCH_SIZE / WRITE_SIZE) + a kernel-trace stats pass over the same bench command
export TMPDIR=/tmp
out=$1; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err < /dev/null || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/$c.json 2> $out/$c.err < /dev/null || echo "$c pass failed"
done
python3 scratch/pmc_summary.py $out < /dev/null
