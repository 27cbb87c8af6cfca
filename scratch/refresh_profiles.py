"""Turn the outputs of `bash scratch/pmc_persist.sh gpurun_out/<dir>` + a default `python bench.py` line into the
files under profiles/ (run in the build container after the gpurun call):
    python scratch/refresh_profiles.py gpurun_out/<dir> gpurun_out/<bench>.json r02"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, bench_json, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ks = sorted(glob.glob(os.path.join(src, "stats/*/*_kernel_stats.csv")), key=os.path.getmtime)[-1]
kt = ks.replace("_kernel_stats", "_kernel_trace")
shutil.copy(ks, os.path.join(root, "profiles/%s_bench_cfg4_kernel_stats.csv" % tag))
rows = list(csv.DictReader(open(kt)))
prof_bench = json.loads(open(os.path.join(src, "stats_bench.json")).read().strip().splitlines()[-1])
kname = prof_bench["roofline"]["kernel"]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if kname.split("<")[0] in r["Kernel_Name"])
json.dump(prof_bench, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu_profiled.json" % tag), "w"), indent=1)
launch = {"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2000 --warmup 200 "
                    "--no-cpu-baseline --no-configs --no-pmc (kernel_trace.csv of the run whose summary is %s_bench_cfg4_kernel_stats.csv)" % tag,
          "kernel": kname,
          "note": "the stats file averages launches of different iteration counts (device spin-up: 256 iterations each; "
                  "warm-up: 200; timed region: 2000): the longest launch is the timed region",
          "launch_durations_ms_sorted": [round(x, 4) for x in d],
          "timed_2000_iteration_launch_ms_kernel_trace": d[-1],
          "bench_hip_event_ms_for_the_timed_2000_iteration_launch": prof_bench["roofline"]["kernel_ms"]}
json.dump(launch, open(os.path.join(root, "profiles/%s_launch_durations.json" % tag), "w"), indent=1)
# duration of the profiled launch in the SQ-counter pass (its kernel trace): the clock of the busy fractions below.
# (GRBM_GUI_ACTIVE / 8 served as that clock in round 2; on round 3's box the counter read 1.48 x the cycles the trace's
# duration allows at 2.4 GHz, so the duration is used and the counter kept for the record.)
sq_rows = []
for f in glob.glob(os.path.join(src, "SQ_LDS_IDX_ACTIVE", "*", "*kernel_trace.csv")):
    sq_rows += [r for r in csv.DictReader(open(f)) if kname.split("<")[0] in r["Kernel_Name"]]
launch_ns = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sq_rows)
cycles = launch_ns * 2.4                    # shader clock 2.4 GHz
p = json.load(open(os.path.join(src, "pmc_summary.json")))
k = [x for x in p if kname.split("<")[0] in x][0]
v = {a: b["mean"] for a, b in p[k].items()}
summary = {"kernel": kname, "workload": "cfg4: 1024 chains x 100k lineages", "iterations_in_profiled_launch": 1000,
           "counters_mean_per_launch": v,
           "hbm_bytes_per_iteration": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1000,
           "correction": "gfx950: FETCH_SIZE counts half the bytes of 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact; both in KiB",
           "profiled_launch_ms": launch_ns / 1e6, "cycles_at_2.4GHz": cycles,
           "lds_busy_fraction": v["SQ_LDS_IDX_ACTIVE"] / cycles / 256,
           "valu_busy_fraction": v["SQ_ACTIVE_INST_VALU"] / cycles / 256,
           "lds_bank_conflict_fraction": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
           # wave64 vector instructions (VALU + LDS) per SIMD x 4.18 cycles each (scratch/ubench/valu_rate.hip) over the
           # cycles of the launch: how full the SIMDs' issue ports are
           "vector_issue_fraction": (v["SQ_INSTS_VALU"] + v["SQ_INSTS_LDS"]) / (256 * 4) * 4.18 / cycles,
           "vector_instructions_per_CU_per_iteration": (v["SQ_INSTS_VALU"] + v["SQ_INSTS_LDS"]) / 256 / 1000,
           "wave_cycles_split": {x: v[x] / v["SQ_WAVE_CYCLES"] for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if x in v},
           "source": "rocprofv3 --pmc <counter group> --kernel-trace (separate passes) -- python3 scratch/prof_persist.py (one launch of "
                     "1000 iterations)"}
json.dump(summary, open(os.path.join(root, "profiles/%s_pmc_1000it.json" % tag), "w"), indent=1)
b = json.loads(open(bench_json).read().strip().splitlines()[-1])
json.dump(b, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu.json" % tag), "w"), indent=1)
print(json.dumps(launch, indent=1)); print(json.dumps(summary, indent=1))
print("value %.4e  kernel_ms %.3f  cpu %s" % (b["value"], b["roofline"]["kernel_ms"], b.get("cpu_baseline") and b["cpu_baseline"]["value"]))
