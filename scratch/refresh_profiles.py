"""Turn the outputs of `bash scratch/pmc_persist.sh gpurun_out/<dir>` + a default `python bench.py` line into the
files under profiles/ (run in the build container after the gpurun call):
    python scratch/refresh_profiles.py gpurun_out/r01g gpurun_out/bench_default.json v7"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, bench_json, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ks = sorted(glob.glob(os.path.join(src, "stats/*/*_kernel_stats.csv")), key=os.path.getmtime)[-1]
kt = ks.replace("_kernel_stats", "_kernel_trace")
shutil.copy(ks, os.path.join(root, "profiles/r01_%s_bench_cfg4_kernel_stats.csv" % tag))
rows = list(csv.DictReader(open(kt)))
prof_bench = json.loads(open(os.path.join(src, "stats_bench.json")).read().strip().splitlines()[-1])
kname = prof_bench["roofline"]["kernel"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if kname.split("<")[0] in r["Kernel_Name"]]
json.dump(prof_bench, open(os.path.join(root, "profiles/r01_bench_cfg4_1gpu_profiled.json"), "w"), indent=1)
spin = [x for x in d if 3.8 < x < 6]
launch = {"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2000 --warmup 200 "
                    "--no-cpu-baseline (kernel_trace.csv of the run whose summary is r01_%s_bench_cfg4_kernel_stats.csv)" % tag,
          "kernel": kname,
          "note": "the stats file averages launches of different iteration counts (device spin-up: 256 iterations each; "
                  "warm-up: 200; timed region: 2000).  Per-launch durations by iteration count:",
          "launches_256_iterations": {"n": len(spin), "mean_ms": sum(spin) / max(1, len(spin))},
          "launch_200_iterations_ms": [x for x in d if x <= 3.8],
          "launches_2000_iterations_ms": [x for x in d if x > 20],
          "bench_hip_event_ms_for_the_timed_2000_iteration_launch": prof_bench["roofline"]["kernel_ms"]}
json.dump(launch, open(os.path.join(root, "profiles/r01_%s_launch_durations.json" % tag), "w"), indent=1)
p = json.load(open(os.path.join(src, "pmc_summary.json")))
json.dump(p, open(os.path.join(root, "profiles/r01_%s_pmc_1000it.json" % tag), "w"), indent=1)
k = [x for x in p if kname.split("<")[0] in x][0]
v = {a: b["mean"] for a, b in p[k].items()}
tr = {"workload": "cfg4", "chains": 1024, "kernel": kname, "iterations_in_profiled_launch": 1000,
      "fetch_size_kib_per_launch": v["FETCH_SIZE"], "write_size_kib_per_launch": v["WRITE_SIZE"],
      "correction": "gfx950: FETCH_SIZE counts half the bytes of 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact",
      "hbm_bytes_per_iteration": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1000,
      "lds_busy_fraction": v["SQ_LDS_IDX_ACTIVE"] / v["GRBM_GUI_ACTIVE"] / 32,
      "valu_busy_fraction": v["SQ_ACTIVE_INST_VALU"] / v["GRBM_GUI_ACTIVE"] / 32,
      "lds_bank_conflict_fraction": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
      "source": "rocprofv3 --pmc <counter> --kernel-trace (separate passes) -- python3 scratch/prof_persist.py  (one launch of "
                "1000 iterations, 1024 chains x 100k lineages); full table in r01_%s_pmc_1000it.json" % tag}
json.dump(tr, open(os.path.join(root, "profiles/scan_traffic.json"), "w"), indent=1)
b = json.loads(open(bench_json).read().strip().splitlines()[-1])
if b["roofline"].get("traffic") is None:
    b["roofline"]["traffic"] = tr["hbm_bytes_per_iteration"] * b["roofline"]["iterations_per_launch"]
json.dump(b, open(os.path.join(root, "profiles/r01_bench_cfg4_1gpu.json"), "w"), indent=1)
print(json.dumps(launch, indent=1)); print(json.dumps(tr, indent=1))
print("value %.4e  kernel_ms %.3f  cpu %s" % (b["value"], b["roofline"]["kernel_ms"], b.get("cpu_baseline") and b["cpu_baseline"]["value"]))
