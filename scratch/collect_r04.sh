#!/bin/bash
# Round-4 profile set, on the MI355X box: bash scratch/collect_r04.sh gpurun_out/r04
export TMPDIR=/tmp
out=$1; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err < /dev/null || echo "bench default failed"
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err < /dev/null || echo "bench driver args failed"
bash scratch/pmc_persist.sh $out/pmc > $out/pmc.log 2>&1
bash scratch/pmc_persist_long.sh $out/pmc_long > $out/pmc_long.log 2>&1
bash scratch/pmc_spec.sh $out/pmc_spec 10000 256 > $out/pmc_spec.log 2>&1
# the HBM-streaming entry points: kernel trace + stats of three calls each at 3e7 and 1e7 lineages
for n in 30000000 10000000; do
  for k in "lr_bin_unit_events 0" "lr_bd_loglik_batch 1" "lr_bd_loglik_batch 8"; do
    set -- $k
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/abi_${1}_c${2}_n${n} -- python3 bench.py --abi-child --abi-kernel $1 --abi-n $n --chains $2 > /dev/null 2>&1 < /dev/null || echo "abi trace $1 $2 $n failed"
  done
done
echo collected
