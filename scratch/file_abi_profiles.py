"""File the ABI parts of a collected set (bash scratch/collect_r04.sh gpurun_out/<dir>) under profiles/:
    python scratch/file_abi_profiles.py gpurun_out/<dir> r04
-> profiles/<tag>_abi.json (the `abi` section of the default bench line), profiles/<tag>_abi_kernel_stats.json (the kernel
stats of the rocprofv3 traces of `bench.py --abi-child ...`, three calls each), profiles/<tag>_bench_cfg4_1gpu.json and
profiles/<tag>_bench_cfg4_1gpu_driver_args.json (the two bench lines)."""
import csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
def line(p):
    """bench_detail.json of a run (one document), or a captured stdout whose last line is the (compact) bench line"""
    txt = open(p).read().strip()
    try:
        return json.loads(txt)
    except ValueError:
        return json.loads(txt.splitlines()[-1])


b = line(os.path.join(src, "bench_default_detail.json"))
d = line(os.path.join(src, "bench_driver_args_detail.json"))
# the compact lines as the driver sees them (the last stdout line of each run)
json.dump({"default": json.loads(open(os.path.join(src, "bench_default.out")).read().strip().splitlines()[-1]),
           "driver_args": json.loads(open(os.path.join(src, "bench_driver_args.out")).read().strip().splitlines()[-1])},
          open(os.path.join(root, "profiles/%s_bench_compact_lines.json" % tag), "w"), indent=1)
json.dump(b, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu.json" % tag), "w"), indent=1)
json.dump(d, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu_driver_args.json" % tag), "w"), indent=1)
json.dump(b["abi"], open(os.path.join(root, "profiles/%s_abi.json" % tag), "w"), indent=1)
rows = []
for run in sorted(glob.glob(os.path.join(src, "abi_*"))):
    for f in glob.glob(os.path.join(run, "*", "*kernel_stats.csv")) + glob.glob(os.path.join(run, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if not r["Name"].startswith(("lr_", "void lr_")):          # the library's kernels (torch's while the input is made: not filed)
                continue
            rows.append({"run": os.path.basename(run), "kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                         "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])})
json.dump({"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --abi-child --abi-kernel <k> "
                     "--abi-n <n> --chains <c> (three calls each, then three of the read-only yardstick on the same arrays; scratch/collect_r05.sh)", "rows": rows},
          open(os.path.join(root, "profiles/%s_abi_kernel_stats.json" % tag), "w"), indent=1)
print(len(rows), "kernel-stat rows;", "value %.3e" % b["value"], "driver args %.3e" % d["value"])
