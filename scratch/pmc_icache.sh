#!/bin/bash
# instruction-cache behaviour of the persistent kernel (one 1000-iteration launch)
export TMPDIR=/tmp
out=$1; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH --kernel-trace --output-format csv -d $out/icache -- python3 scratch/prof_persist.py > $out/icache.log 2>&1 < /dev/null || echo "icache pass failed"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/icache/*/*counter_collection.csv")
d = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        d[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in d.items():
    if "persist" in k:
        print(k, dict(v))
PY
