#!/bin/bash
# all micro-benchmarks of scratch/ubench in one go (on the MI355X box; binaries built in the container)
out=$1; mkdir -p $out
for b in valu_rate vgpr_bank; do echo "== $b" >> $out/ubench.txt; timeout -k 5 60 scratch/ubench/$b >> $out/ubench.txt 2>&1; done
echo "== lds_gather" >> $out/ubench.txt; timeout -k 5 60 scratch/ubench/lds_gather >> $out/ubench.txt 2>&1
echo "== lds_gather -probe" >> $out/ubench.txt; timeout -k 5 60 scratch/ubench/lds_gather -probe >> $out/ubench.txt 2>&1
echo "== lds_gather -mates" >> $out/ubench.txt; timeout -k 5 100 scratch/ubench/lds_gather -mates >> $out/ubench.txt 2>&1
echo "== lds_gather idx8_cfg4.bin" >> $out/ubench.txt; timeout -k 5 60 scratch/ubench/lds_gather scratch/ubench/idx8_cfg4.bin >> $out/ubench.txt 2>&1
echo "== scan_lab idx8_cfg4.bin" >> $out/ubench.txt; timeout -k 5 100 scratch/ubench/scan_lab scratch/ubench/idx8_cfg4.bin >> $out/ubench.txt 2>&1
echo "== log_check" >> $out/ubench.txt; timeout -k 5 200 scratch/ubench/log_check >> $out/ubench.txt 2>&1
