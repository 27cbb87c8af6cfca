#!/bin/bash
# Round-5 profile set, on the MI355X box: bash scratch/collect_r05.sh gpurun_out/r05
export TMPDIR=/tmp
out=$1; mkdir -p $out
LR_BENCH_DETAIL=$out/bench_default_detail.json python3 bench.py > $out/bench_default.out 2> $out/bench_default.err < /dev/null || echo "bench default failed"
tail -c 300 $out/bench_default.out; echo
LR_BENCH_DETAIL=$out/bench_driver_args_detail.json python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.out 2> $out/bench_driver_args.err < /dev/null || echo "bench driver args failed"
tail -c 300 $out/bench_driver_args.out; echo
LR_BENCH_DETAIL=$out/pmc/stats_bench_detail.json bash scratch/pmc_persist.sh $out/pmc > $out/pmc.log 2>&1
bash scratch/pmc_persist_long.sh $out/pmc_long > $out/pmc_long.log 2>&1
bash scratch/pmc_spec.sh $out/pmc_spec 10000 256 > $out/pmc_spec.log 2>&1
# the HBM-streaming entry points: kernel trace + stats of three calls each (+ three of the read-only yardstick) at 1e8 and 3e7
for n in 100000000 30000000; do
  for k in "lr_bin_unit_events 0" "lr_bd_loglik_batch 1" "lr_bd_loglik_batch 8" "lr_bd_loglik_batch 16"; do
    set -- $k
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/abi_${1}_c${2}_n${n} -- python3 bench.py --abi-child --abi-kernel $1 --abi-n $n --chains $2 > /dev/null 2>&1 < /dev/null || echo "abi trace $1 $2 $n failed"
  done
done
echo collected
