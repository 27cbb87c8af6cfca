"""Turn the outputs of `bash scratch/pmc_persist.sh gpurun_out/<dir>` + a default `python bench.py` line into the
files under profiles/ (run in the build container after the gpurun call):
    python scratch/refresh_profiles.py gpurun_out/<dir> gpurun_out/<bench>.json r02"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, bench_json, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def load(path):
    """a bench DETAIL file (bench_detail.json: one JSON document) or a captured stdout (the compact line is its last line)"""
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except ValueError:
        return json.loads(txt.splitlines()[-1])



ks = sorted(glob.glob(os.path.join(src, "stats/*/*_kernel_stats.csv")), key=os.path.getmtime)[-1]
kt = ks.replace("_kernel_stats", "_kernel_trace")
shutil.copy(ks, os.path.join(root, "profiles/%s_bench_cfg4_kernel_stats.csv" % tag))
rows = list(csv.DictReader(open(kt)))
prof_bench = load(os.path.join(src, "stats_bench_detail.json") if os.path.exists(os.path.join(src, "stats_bench_detail.json")) else os.path.join(src, "stats_bench.json"))
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
           # VALU issue cycles over the cycles of the launch, bracketed by the two issue classes of profiles/r04_ubench.txt
           # (every instruction full rate: 2.35 cycles; every instruction half rate: 4.2)
           "valu_issue_fraction_bounds": [v["SQ_INSTS_VALU"] / (256 * 4) * 2.35 / cycles, v["SQ_INSTS_VALU"] / (256 * 4) * 4.2 / cycles],
           "vector_instructions_per_CU_per_iteration": (v["SQ_INSTS_VALU"] + v["SQ_INSTS_LDS"]) / 256 / 1000,
           "wave_cycles_split": {x: v[x] / v["SQ_WAVE_CYCLES"] for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if x in v},
           "source": "rocprofv3 --pmc <counter group> --kernel-trace (separate passes) -- python3 scratch/prof_persist.py (one launch of "
                     "1000 iterations)"}
json.dump(summary, open(os.path.join(root, "profiles/%s_pmc_1000it.json" % tag), "w"), indent=1)
b = load(bench_json)
json.dump(b, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu.json" % tag), "w"), indent=1)
print(json.dumps(launch, indent=1)); print(json.dumps(summary, indent=1))
print("value %.4e  kernel_ms %.3f  cpu %s" % (b["value"], b["roofline"]["kernel_ms"], b.get("cpu_baseline") and b["cpu_baseline"]["value"]))

# optional: the same counters over one launch of the speculative kernel (bash scratch/pmc_spec.sh <dir> 10000 256)
if len(sys.argv) > 4:
    sp = sys.argv[4]
    ps = json.load(open(os.path.join(sp, "pmc_summary.json")))
    ksp = [x for x in ps if "lr_spec_kernel" in x][0]
    vs = {a: b["mean"] for a, b in ps[ksp].items()}
    rows = []
    for f in glob.glob(os.path.join(sp, "SQ_LDS_IDX_ACTIVE", "*", "*kernel_trace.csv")):
        rows += [r for r in csv.DictReader(open(f)) if "lr_spec_kernel" in r["Kernel_Name"]]
    ns = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    cyc, n_it, blocks = ns * 2.4, 2000, 256
    spec = {"kernel": ksp.replace("void ", ""),
            "workload": "cfg3: 256 chains x 10k lineages, a team per chain (256 blocks), one launch of 2000 iterations from the initial state",
            "counters_mean_per_launch": vs, "profiled_launch_ms": ns / 1e6, "us_per_iteration_in_profiled_launch": ns / 1e3 / n_it,
            "valu_busy_fraction": vs["SQ_ACTIVE_INST_VALU"] / cyc / blocks, "lds_busy_fraction": vs["SQ_LDS_IDX_ACTIVE"] / cyc / blocks,
            "valu_issue_fraction_bounds": [vs["SQ_INSTS_VALU"] / (blocks * 4) * 2.35 / cyc, vs["SQ_INSTS_VALU"] / (blocks * 4) * 4.2 / cyc],
            "vector_instructions_per_CU_per_iteration": (vs["SQ_INSTS_VALU"] + vs["SQ_INSTS_LDS"]) / blocks / n_it,
            "salu_instructions_per_CU_per_iteration": vs["SQ_INSTS_SALU"] / blocks / n_it,
            "note": "2 candidate + 2 helper + 8 scanner waves per CU; the iteration is the serial chain candidate (to the hand-over) -> "
                    "helper (table) -> barrier -> scanners (pair planes, scan, decision): the CU issues about half of what it could",
            "source": "rocprofv3 --pmc <counter group> --kernel-trace (separate passes) -- python3 scratch/prof_spec.py 10000 256 "
                      "(bash scratch/pmc_spec.sh <dir> 10000 256)"}
    json.dump(spec, open(os.path.join(root, "profiles/%s_pmc_spec_cfg3.json" % tag), "w"), indent=1)
    print(json.dumps({k: v for k, v in spec.items() if k != "counters_mean_per_launch"}, indent=1))
# optional: the bench line at the driver's arguments
if len(sys.argv) > 5:
    b20 = load(sys.argv[5])
    json.dump(b20, open(os.path.join(root, "profiles/%s_bench_cfg4_1gpu_driver_args.json" % tag), "w"), indent=1)
    print("driver args: value %.4e  wall/device %.3f" % (b20["value"], b20["ms_per_step"] * b20["steps"] / b20["roofline"]["kernel_ms"]))
# optional: the SQ counters over one 4000-iteration launch (bash scratch/pmc_persist_long.sh <dir>)
if len(sys.argv) > 6:
    lg = sys.argv[6]
    pl = json.load(open(os.path.join(lg, "pmc_summary.json")))
    kl = [x for x in pl if kname.split("<")[0] in x][0]
    vl = {a_: b_["mean"] for a_, b_ in pl[kl].items()}
    rows = []
    for f in glob.glob(os.path.join(lg, "SQ_LDS_IDX_ACTIVE", "*", "*kernel_trace.csv")):
        rows += [r for r in csv.DictReader(open(f)) if kname.split("<")[0] in r["Kernel_Name"]]
    ns = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    n_it, cyc, us_bench = 4000, ns * 2.4, b["ms_per_step"] * 1e3
    valu, lds = vl["SQ_INSTS_VALU"] / 256 / n_it, vl["SQ_INSTS_LDS"] / 256 / n_it
    long_ = {"kernel": kl.replace("void ", ""),
             "workload": "cfg4: 1024 chains x 100k lineages, ONE launch of 4000 iterations from the initial state (the chains reach their "
                         "working number of shifts after ~500)",
             "iterations_in_profiled_launch": n_it, "counters_mean_per_launch": vl, "profiled_launch_ms": ns / 1e6,
             "us_per_iteration_profiled": ns / 1e3 / n_it, "us_per_iteration_unprofiled_bench": us_bench,
             "per_CU_per_iteration": {"valu": valu, "lds": lds, "salu": vl["SQ_INSTS_SALU"] / 256 / n_it, "vmem_rd": vl["SQ_INSTS_VMEM_RD"] / 256 / n_it},
             "valu_busy_fraction_profiled": vl["SQ_ACTIVE_INST_VALU"] / cyc / 256, "lds_busy_fraction_profiled": vl["SQ_LDS_IDX_ACTIVE"] / cyc / 256,
             "lds_bank_conflict_fraction": vl["SQ_LDS_BANK_CONFLICT"] / vl["SQ_LDS_IDX_ACTIVE"],
             "valu_issue_fraction_at_bench_speed_bounds": [valu * 2.35 / 4 / (us_bench * 2400), valu * 4.2 / 4 / (us_bench * 2400)],
             "scan_loop_valu_issue_fraction_at_bench_speed": 2 * 7143 / 64.0 * 92.55 / 4 / (us_bench * 2400),
             "lds_data_cycles_fraction_at_bench_speed": vl["SQ_LDS_IDX_ACTIVE"] / 256 / n_it / (us_bench * 2400),
             "note": "the counter passes slow the launch (us_per_iteration_profiled against the unprofiled bench line); the *_at_bench_speed "
                     "fractions put the counts of this launch over the unprofiled iteration time.  VALU issue: a wave64 instruction costs "
                     "2.35 (plain 32-bit) or 4.2 cycles (fp64, SDWA / DPP, three-operand integer, compares) of its SIMD (profiles/r04_ubench.txt) - "
                     "the bounds take all instructions as one class; the scan loop's own share is exact (2 pairs x 7143 groups / 64 lanes x 92.55 cycles per trip "
                     "over 4 SIMDs; rounds 3 and 4 filed twice that - the pairs counted as four chains).  SQ_LDS_IDX_ACTIVE = cycles the LDS data path is busy, per CU",
             "source": "bash scratch/pmc_persist_long.sh <dir>: rocprofv3 --pmc <group> --kernel-trace (separate passes) -- python3 "
                       "scratch/prof_persist.py with LR_PROF_ITERS=4000"}
    json.dump(long_, open(os.path.join(root, "profiles/%s_pmc_4000it.json" % tag), "w"), indent=1)
    print(json.dumps({k_: v_ for k_, v_ in long_.items() if k_ != "counters_mean_per_launch"}, indent=1))
