#!/usr/bin/env python3
"""bench.py - RJMCMC birth-death likelihood loop on MI355X: lineage-log-lik evals/s.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 without WORLD_SIZE: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one RJMCMC iteration of every chain on this GPU: one scan of the lineages scoring all pending proposals
+ the per-chain accept / trace / next-proposal step, all inside the persistent engine kernel (one launch runs the K
timed iterations).  Workload (config.workload) = BASELINE.json configs[3] ("cfg4"): synthetic 100k lineages, 128 unit
bins, 20 true shifts per process, 1024 chains per GPU; chains shard across ranks with no data-path collective (weak
scaling), lineage arrays are replicated; the trace rows sampled inside the timed region (every --sample-every
iterations) are gathered to rank 0 over RCCL before the clock stops.  Inputs are resident in HBM before timing.

value = iterations x lineages x chains / time (one unit = one lineage's contribution to one chain's proposed-state
log-likelihood: "iters/sec x lineages" of BASELINE.json, summed over chains).

Beside the headline the line carries
  roofline      - the dominant kernel against its physical bound (the LDS gather rate), with the
                  HBM figures SURVEY 8(d) asks for as side fields and the HBM traffic measured by a rocprofv3 --pmc
                  child run of the same launch (null + the reason when the profiler is not usable);
  configs       - the other BASELINE configurations (cfg2, cfg3, cfg5), cfg4 on general (non-integer) lineage times,
                  and cfg4 exactly as BASELINE.json words it (1024 chains over 8 GPUs = a 128-chain shard);
  strong_scaling- for N > 1: the same 1024 chains in total, sharded over the N ranks, timed the same way;
  cpu_baseline  - the numpy port (oracle/) on the host cores of the same box;
  abi           - the kernels HBM bounds, at sizes where it does (1e7 / 3e7 lineages): lr_bin_unit_events and
                  lr_bd_loglik_batch against the 8 TB/s peak with the FETCH_SIZE traffic of the same call, the
                  launch-based ENGINE on 16 chains x 1e7 / 3e7 lineages (the RJMCMC loop streaming ts / te every
                  iteration), and the cost of the calc_likelihood seam per call.
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_POLLING_SET_HERE = False   # main() switched the runtime to polling for completion signals (children get the default back)

WORKLOADS = {
    # name: (lineages, n_bins, true shifts, chains per GPU, model, general times)
    "cfg4": (100_000, 128, 20, 1024, 0, False),
    "cfg3": (10_000, 128, 20, 256, 0, False),
    # BASELINE.json configs[1]: the shipped metal_bands lineages (30,217; the fixture holds the parsed file), 128 chains,
    # model_BDI 2 as in the reference's tutorial run
    "cfg2": (30_217, 32, 0, 128, 2, False),
    # BASELINE.json configs[4]: the DDRate.py sampler (model "dd": -m_birth 2 -m_death 2) on 50k lineages, 256 chains
    "cfg5": (50_000, 64, 6, 256, "dd", False),
    # cfg4 with continuous (non-integer) birth / death times: the per-lineage form of BDIx:124-160 proper
    "cfg4_general": (100_000, 128, 20, 1024, 0, True),
    # BASELINE.json configs[3] as worded: 1024 chains over 8 GPUs = 128 chains on each
    "cfg4_shard128": (100_000, 128, 20, 128, 0, False),
}
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LDS_PEAK_GBS = 256 * 256 * 2.4e9 / 1e9     # 256 B/clk/CU x 256 CUs x 2.4 GHz = 157,286 GB/s (MI355X_MICROARCH.md, LDS)
# Vector-instruction ISSUE cost, cycles per wave64 instruction per SIMD at >= 2 waves per SIMD (ONE table for every class:
# scratch/ubench/issue_rate.hip -> profiles/r04_ubench.txt).  Two classes: 2.3-2.4 for the plain 32-bit VOP1 / VOP2 forms
# (v_and_b32 - also with a literal -, v_lshrrev_b32, v_add_u32, v_xor_b32, v_mov_b32, v_add_f32, v_fma_f32); 4.2 for every
# fp64 instruction (add, mul, fma, max, compare, v_cvt_f64_u32), every SDWA and DPP form, v_lshlrev_b32, the three-operand
# integer forms (v_bfe_u32, v_and_or_b32, v_perm_b32, v_lshl_add_u32, v_mad_u32_u24, v_med3_i32), v_mul_lo_u32, v_cndmask
# under an SGPR mask, and any VALU instruction with an SGPR source operand.
ISSUE_CYC_FULL = 2.35
ISSUE_CYC_HALF = 4.2
# the compiled scan loops (ISA of lr_persist4_kernel<136, .>): (half-rate instructions, full-rate instructions) per trip
# of one 16-byte group = 14 lineages x 2 chains
SCAN_LOOP_MIX = {True: (16 + 1, 9),        # unit resolution: 14 v_add_f64 + 2 v_fmac_f64, v_cvt_f64_u32; 4 v_and + 5 v_lshrrev / v_and
                 False: (32 + 8, 9)}       # general times: 18 v_fmac_f64 + 14 v_add_f64, 8 v_cvt_f64_u32; 9 v_and / v_lshrrev


def make_workload(name):
    """-> (ts, te, n_bins requested, model, label)"""
    n_lin, n_bins, n_shifts, _, model, general = WORKLOADS[name]
    from literate_amd import synth
    if name == "cfg2":
        G = np.load(os.path.join(ROOT, "tests", "golden", "binning_lik.npz"))
        return G["metal_bands/ts"], G["metal_bands/te"], model, "shipped metal_bands_1.tsv, 30217 lineages"
    ts, te, _ = synth.make_lineages(n_lin, n_bins=n_bins, n_shifts=n_shifts, seed=0)   # same on every rank
    label = "synthetic %d lineages, %d unit bins, %d true shifts" % (n_lin, n_bins, n_shifts)
    if general:
        # continuous times: births anywhere inside their year, deaths anywhere inside theirs (same bins, same events)
        rng = np.random.default_rng(7)
        ts = ts + rng.uniform(0.0, 1.0, len(ts)) * 0.999
        te = np.maximum(np.ceil(te) - 1.0 + rng.uniform(1e-3, 0.999, len(te)), ts + 1e-3)
        label += ", continuous times"
    return ts, te, model, label


def make_engine(name, ts, te, model, chains, chain_offset, s_freq, n_slots, engine="auto"):
    if model == "dd":
        from literate_amd.ddrate import DDRateEngine
        return DDRateEngine(ts, te, float(ts.min()), float(te.max()), chains, m_birth=2, m_death=2, seed=2026,
                            s_freq=s_freq, n_trace_slots=n_slots, chain_offset=chain_offset, engine=engine)
    from literate_amd.engine import ChainEngine
    return ChainEngine(ts, te, chains, model=model, seed=2026, s_freq=s_freq, n_trace_slots=n_slots,
                       chain_offset=chain_offset, engine=engine)


def kernel_figures(eng, n_lin, chains, n_iters, kernel_ms):
    """LDS-roofline figures of `n_iters` iterations that took `kernel_ms` of device time."""
    unit = bool(eng.unit_resolution)
    if eng.layout.persistent:
        # persistent engines: groups of up to 14 lineages of one birth bin in 7 slots (lr_pack.hip; a slot = one lineage or a
        # pair through a pair-sum plane): per group and chain PAIR one gather of the birth entry + 7 of the slots' entries,
        # 16 B each (unit resolution); value and slope entry each on general times.  The speculative kernel with a team
        # per CHAIN scores ONE chain per gather (the entry's second half is unused): twice the bytes per eval
        lds_bytes_per_eval = (1 + 7) * (16 if unit else 32) / (14.0 * chains_per_gather(eng))
    else:
        lds_bytes_per_eval = 16 if unit else 32      # launch-based scan: two 8-byte / two 16-byte entries per (lineage, chain)
    evals = float(n_iters) * n_lin * chains
    achieved = evals * lds_bytes_per_eval / (kernel_ms * 1e-3) / 1e9
    out = dict(kernel=eng.kernel_name(), us_per_iter=kernel_ms / n_iters * 1e3, evals_per_s=evals / (kernel_ms * 1e-3),
               lds_bytes_per_eval=lds_bytes_per_eval, lds_GBs=achieved, lds_frac=achieved / LDS_PEAK_GBS,
               unit_resolution_tables=unit, persistent=int(eng.layout.persistent),
               threads_per_block=int(eng.layout.reserved1))
    out.update(eval_cost(eng))
    # SURVEY 8(d)'s un-amortised "effective GB/s": 16 B (fp64 ts + te) per (lineage, chain) evaluation
    out["effective_GBs_16B_per_eval"] = 16.0 * evals / (kernel_ms * 1e-3) / 1e9
    if eng.layout.persistent:
        # What the scan loop is really bound by (profiles/r04_ubench.txt): vector instruction ISSUE.  The SIMD-cycles one
        # trip (14 lineages x chains_per_gather chains x 64 lanes) costs = its half-rate instructions x 4.2 + its full-rate
        # ones x 2.35; the LDS gathers (8 / 16 ds_read_b128 per trip) and the group load issue beside them.  The chain
        # steps share the same SIMDs, so the fraction below is the share of the chip's issue CYCLES spent on SCAN-LOOP
        # vector instructions.
        n_half, n_full = SCAN_LOOP_MIX[unit]
        cyc_trip = n_half * ISSUE_CYC_HALF + n_full * ISSUE_CYC_FULL
        evals_trip = 14.0 * chains_per_gather(eng) * 64
        peak = 256 * 4 * 2.4e9 / cyc_trip * evals_trip
        out["issue"] = dict(half_rate_instr_per_trip=n_half, full_rate_instr_per_trip=n_full, cycles_half=ISSUE_CYC_HALF,
                            cycles_full=ISSUE_CYC_FULL, simd_cycles_per_trip=cyc_trip, evals_per_trip=evals_trip,
                            vector_instr_per_eval=(n_half + n_full) / (14.0 * chains_per_gather(eng)),
                            peak_evals_per_s=peak, frac=out["evals_per_s"] / peak)
    return out


def chains_per_gather(eng):
    """chains one 16-byte table gather of the persistent scans serves: a pair - except under the speculative kernel with
    a team per chain (lr_spec_kernel<..., MODE 1|2|3>), whose table entries are (chain, unused)"""
    return 1 if (int(eng.layout.persistent) == 3 and int(eng.layout.spec_chains_per_team) == 1) else 2


def eval_cost(eng):
    """What one counted eval costs in the scan loop, per (lineage, chain), from the group format (csrc/lr_scan.h):
    a 16-byte group = up to 14 lineages of one birth bin in 7 slots, scored for a chain PAIR with 1 birth gather
    (applied `count` times by one fma) + 7 slot gathers (a slot = one lineage's death entry or the pre-summed entry of
    two neighbouring lineages, planes derived from the chain's own table every iteration).  The aggregation level is
    frozen at this: run-length on the birth side, pairs on the death side - no wider slots, no count-weighted slots."""
    unit = bool(eng.unit_resolution)
    if not eng.layout.persistent:
        return dict(gathers_per_eval=2.0, fp64_ops_per_eval=3.0 if unit else 4.0, aggregation="none: launch-based scan, "
                    "two table gathers per (lineage, chain)")
    cpg = chains_per_gather(eng)
    if cpg == 1:
        # a team per chain: the same 8 gathers - and the same pair arithmetic, its second half on zeros - serve 14
        # lineages of ONE chain
        return dict(gathers_per_eval=(8 if unit else 16) / 14.0, fp64_ops_per_eval=(17 if unit else 40) / 14.0,
                    aggregation="as the pair form (run-length on the birth side, pairs on the death side), one chain per "
                                "gather: a team of blocks per chain, table entries (chain, unused)")
    if unit:
        # 8 ds_read_b128 serve 14 lineages x 2 chains; 6 + 6 adds, 2 fma, 2 accumulates, 1 conversion of the count
        return dict(gathers_per_eval=8 / 28.0, fp64_ops_per_eval=17 / 28.0,
                    aggregation="per group of <= 14 lineages of one birth bin: birth entry gathered once and multiplied by "
                                "the run count; death entries gathered per slot of one lineage or a pre-summed pair")
    # general times: value + slope entry per gather site (16 reads), per slot one conversion + 2 fma, 12 adds, 4 fma on the
    # birth side, 2 accumulates, 1 conversion of the count
    return dict(gathers_per_eval=16 / 28.0, fp64_ops_per_eval=(7 * 3 + 12 + 4 + 2 + 1) / 28.0,
                aggregation="as at unit resolution, pairs only for two lineages that die in the same bin; every lineage "
                            "keeps its own in-bin fractions (32-bit fixed point)")


def side_config(name, steps, warmup):
    """One of the other configurations, ~1 s: device time of `steps` iterations (HIP events on the launch stream)."""
    import torch
    ts, te, model, label = make_workload(name)
    chains = WORKLOADS[name][3]
    eng = make_engine(name, ts, te, model, chains, 0, 100, (steps + warmup) // 100 + 2)
    eng.init()
    eng.steps(warmup)
    torch.cuda.synchronize()
    ms = eng.timed_steps(steps)
    snap = eng.snapshot()
    assert np.all(snap["it"] == steps + warmup) and np.all(np.isfinite(snap["likA"])), name
    out = dict(workload="%s: %s, %d chains, %s" % (name, label, chains,
                                                    "DDRate sampler -m_birth 2 -m_death 2" if model == "dd" else "model_BDI %d" % model),
               lineages=len(ts), chains=chains, steps=steps, iters_per_s_per_chain=steps / (ms * 1e-3))
    out.update(kernel_figures(eng, len(ts), chains, steps, ms))
    out["chains_per_gather"] = chains_per_gather(eng)
    eng.close()
    return out


# ---- CPU legs (oracle/ = the checker, timed as the reported baseline) -------------------------------------------
def _per_lineage_worker(job):
    """One host core: the numpy per-lineage evaluator on the same lineages for `budget_s` seconds."""
    ts, te, t0, n_bins, br, budget_s, seed = job
    from oracle import literate_oracle as lo
    rng = np.random.default_rng(seed)
    pre = lo.lineage_bins(ts, te, t0, n_bins)          # index/fraction pass is data-only: not re-timed
    n_eval, t_start = 0, time.perf_counter()
    while True:
        lam = np.exp(rng.uniform(np.log(.05), np.log(.6), n_bins))
        mu = np.exp(rng.uniform(np.log(.02), np.log(.3), n_bins))
        lo.per_lineage_loglik(ts, te, t0, lam, mu, 0, br, pre=pre)
        n_eval += 1
        el = time.perf_counter() - t_start
        if el > budget_s:
            return n_eval, el


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            return next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        return "unknown"


def cpu_baseline(ts, te, t0, n_bins, stats, start_time, end_time, budget_s=8.0):
    """CPU numbers beside the GPU one, all from the numpy port under oracle/, bounded to ~25 s:
    value      : per-lineage evaluator (oracle.per_lineage_loglik: O(N) gather form of get_BDlik) on ONE core,
    all_cores  : the same evaluator, one process per host core (fresh interpreters, no GPU),
    reference_loop : the reference's own algorithm - the whole RJMCMC iteration on binned sufficient statistics,
                 O(n_bins) per iteration, one chain on one core - as iterations/s and iterations/s x lineages."""
    import multiprocessing as mp
    br = stats["br"]
    n_eval, el = _per_lineage_worker((ts, te, t0, n_bins, br, budget_s, 0))
    out = dict(value=n_eval * len(ts) / el, unit="lineage-log-lik evals/s", cores=1, kind="port",
               cpu_model=cpu_model_name(),
               sample="%d chain states x %d lineages (same synthetic lineages, model 0), %.1f s of numpy on 1 core"
                      % (n_eval, len(ts), el))
    try:
        cores = len(os.sched_getaffinity(0))           # every core this process may run on (all host cores)
    except AttributeError:
        cores = os.cpu_count() or 1
    if cores > 1:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_per_lineage_worker, [(ts, te, t0, n_bins, br, 5.0, 100 + i) for i in range(cores)], chunksize=1)
        out["all_cores"] = dict(value=sum(n for n, _ in res) * len(ts) / max(e for _, e in res), cores=cores,
                                sample="one process per core of sched_getaffinity (%d), 5 s each" % cores)
    out["effective_GBs_16B_per_eval"] = 16.0 * out["value"] / 1e9
    from oracle import mcmc_oracle as mo
    n_it = 20000
    t_start = time.perf_counter()
    mo.run_mcmc(stats, start_time, end_time, mo.Settings(model_BDI=0),
                mo.PhiloxDraws(2026, 0), n_it, 100)
    el = time.perf_counter() - t_start
    out["reference_loop"] = dict(iters_per_s=n_it / el, iters_per_s_x_lineages=n_it / el * len(ts), cores=1,
                                 sample="%d RJMCMC iterations of one chain on binned statistics, %.1f s" % (n_it, el))
    return out


def cpu_baseline_dd(ts, te, budget_s=4.0):
    """cfg5's CPU leg: DDRate.py's own loop (one chain, binned create_bins statistics; oracle/dd_mcmc_oracle.py) on
    one core, as iterations/s and iterations/s x lineages."""
    from oracle import dd_mcmc_oracle as do
    from oracle import literate_oracle as lo
    origin, present, n_spec, n_exti, DT, n_time_bins, time_range = lo.create_bins(float(ts.min()), float(te.max()), ts, te, 0)
    n_it = 400
    t_start = time.perf_counter()
    done = 0
    while time.perf_counter() - t_start < budget_s:
        do.run_dd_mcmc(n_spec, n_exti, DT, time_range, origin, present, 2, 2, do.NumpyLegacyDraws(), n_it, 100)
        done += n_it
    el = time.perf_counter() - t_start
    return dict(value=done / el * len(ts), unit="lineage-log-lik evals/s", cores=1, kind="port",
                iters_per_s=done / el, cpu_model=cpu_model_name(),
                sample="%d iterations of DDRate.py's loop (one chain, binned statistics, -m_birth 2 -m_death 2), %.1f s of "
                       "numpy on 1 core; value = iterations/s x %d lineages" % (done, el, len(ts)))


# ---- HBM traffic of the dominant kernel: rocprofv3 --pmc child runs of the same launch ----------------------------
def pmc_child(args):
    """--pmc-child: exactly one engine launch of `steps` iterations on the named workload, nothing else."""
    import torch
    ts, te, model, _ = make_workload(args.workload)
    chains = args.chains or WORKLOADS[args.workload][3]
    eng = make_engine(args.workload, ts, te, model, chains, 0, args.sample_every,
                      args.steps // args.sample_every + 2, engine=args.engine)
    eng.init()
    eng.steps(args.steps)
    torch.cuda.synchronize()
    eng.close()


def measure_traffic(workload, chains, steps, kname, engine, sample_every):
    """HBM bytes of ONE launch of the dominant kernel, the way MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE and
    WRITE_SIZE in separate --pmc passes, FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B), both in KiB.
    Returns (bytes or None, note)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    steps = min(steps, 4096)                      # one launch
    vals = {}
    base = kname.split("<")[0]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lr_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", workload, "--chains", str(chains),
               "--steps", str(steps), "--engine", engine, "--sample-every", str(sample_every)]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + (("HSA_ENABLE_INTERRUPT",) if _POLLING_SET_HERE else ()):
            env.pop(k, None)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=150)
        except (subprocess.TimeoutExpired, OSError) as ex:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 --pmc %s: %s" % (counter, type(ex).__name__)
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                rows += [x for x in csv.DictReader(fh) if x.get("Counter_Name") == counter and base in x.get("Kernel_Name", "")]
        shutil.rmtree(d, ignore_errors=True)
        if r.returncode != 0 or not rows:
            return None, "rocprofv3 --pmc %s: rc %d, %d rows for %s" % (counter, r.returncode, len(rows), base)
        # the engine's launch of `steps` iterations is the longest dispatch of that kernel: the largest counter value
        vals[counter] = max(float(x["Counter_Value"]) for x in rows)
    hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return hbm * 1.0, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over one %d-iteration launch of %s: "
                       "FETCH_SIZE %.1f KiB x 2 (gfx950 correction) + WRITE_SIZE %.1f KiB"
                       % (steps, kname, vals["FETCH_SIZE"], vals["WRITE_SIZE"]))


# ---- the ABI-level streaming kernels against the HBM roofline ----------------------------------------------------
# lr_bin_unit_events (precompute_events / get_br for every unit window: lib:74-85, loop LRF:519-523) and lr_bd_loglik_batch
# (per-lineage get_BDlik, BDIx:124-146 = the calc_likelihood seam INTEGRATION.md binds) are the kernels that stream
# ts / te from HBM: 16 B per lineage and pass.  The engines never re-read ts / te, so these two are where
# the north star's HBM roofline is testable - at lineage counts where HBM matters (1e7, 3e7: 160 / 480 MB per pass).
ABI_SIZES = (10_000_000, 30_000_000, 100_000_000)      # 160 MB, 480 MB, 1.6 GB per pass: the last is 6 x the Infinity Cache
ABI_BINS = 128
ABI_ROTATE_BYTES = 1.2e9          # a timed call never finds its input in the 256 MiB Infinity Cache: >= this many bytes of
                                  # OTHER lineage arrays are streamed between two uses of one (ts, te) pair


def abi_lineages(n, general, order="sorted"):
    """n synthetic lineages in HBM: cfg4's generator output tiled to n; `general` moves every time off the year grid
    (own in-bin fractions); order = 'sorted' by birth time (how the reference's input files are written) or 'shuffled'."""
    import torch
    from literate_amd import synth
    ts0, te0, _ = synth.make_lineages(100_000, n_bins=ABI_BINS, n_shifts=20, seed=0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    reps = -(-n // len(ts0))
    ts = torch.as_tensor(ts0, device="cuda").repeat(reps)[:n].contiguous()
    te = torch.as_tensor(te0, device="cuda").repeat(reps)[:n].contiguous()
    if general:
        ts = ts + torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 0.999
        te = torch.maximum(torch.ceil(te) - 1.0 + 1e-3 + 0.998 * torch.rand(n, generator=g, device="cuda", dtype=torch.float64),
                           ts + 1e-3)
    if order == "sorted":
        ts, perm = torch.sort(ts, stable=True)
        te = te[perm].contiguous()
    else:
        perm = torch.randperm(n, generator=g, device="cuda")
        ts, te = ts[perm].contiguous(), te[perm].contiguous()
    return ts, te


def abi_calls(kernel, ts, te, n_chains, model=2):
    """-> (closure that enqueues ONE ABI call on torch's current stream with everything pre-marshalled, outputs, info)"""
    import torch
    from literate_amd import _hip
    lib = _hip.load()
    n = ts.numel()
    dev = ts.device
    stream = _hip.stream_ptr(dev)
    if kernel == "lr_bin_events":
        lo = torch.arange(ABI_BINS, dtype=torch.float64, device=dev)
        hi = lo + 1.0
        sp = torch.empty(ABI_BINS, dtype=torch.int64, device=dev)
        ex = torch.empty_like(sp)
        br = torch.empty(ABI_BINS, dtype=torch.float64, device=dev)
        ws = torch.empty(int(lib.lr_bin_events_workspace_bytes(n, ABI_BINS)), dtype=torch.uint8, device=dev)
        args = (_hip.ptr(ts), _hip.ptr(te), n, _hip.ptr(lo), _hip.ptr(hi), ABI_BINS, _hip.ptr(sp), _hip.ptr(ex),
                _hip.ptr(br), _hip.ptr(ws), ws.numel(), stream)
        keep = (lo, hi, sp, ex, br, ws)

        def call():
            rc = lib.lr_bin_events(*args)
            assert rc == 0, rc
        return call, (sp, ex, br), dict(passes=1, windows=ABI_BINS), keep
    if kernel == "lr_bin_unit_events":
        sp = torch.empty(ABI_BINS, dtype=torch.int64, device=dev)
        ex = torch.empty_like(sp)
        br = torch.empty(ABI_BINS, dtype=torch.float64, device=dev)
        ws = torch.empty(int(lib.lr_bin_unit_events_workspace_bytes(n, ABI_BINS)), dtype=torch.uint8, device=dev)
        args = (_hip.ptr(ts), _hip.ptr(te), n, 0.0, ABI_BINS, _hip.ptr(sp), _hip.ptr(ex), _hip.ptr(br), _hip.ptr(ws),
                ws.numel(), stream)
        keep = (sp, ex, br, ws)

        def call():
            rc = lib.lr_bin_unit_events(*args)
            assert rc == 0, rc
        return call, (sp, ex, br), dict(passes=1, windows=ABI_BINS), keep
    rng = np.random.default_rng(5)
    lam = torch.as_tensor(np.exp(rng.uniform(np.log(.05), np.log(.6), (n_chains, ABI_BINS))), device=dev)
    mu = torch.as_tensor(np.exp(rng.uniform(np.log(.02), np.log(.3), (n_chains, ABI_BINS))), device=dev)
    out = torch.empty(n_chains, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.lr_bd_loglik_workspace_bytes(n, ABI_BINS, n_chains, model)), dtype=torch.uint8, device=dev)
    plan = (_hip.c_i32 * 4)()
    rc = lib.lr_bd_loglik_plan(n, ABI_BINS, n_chains, model, plan)
    assert rc == 0, rc
    cb, tiles, H = int(plan[0]), int(plan[1]), int(plan[2])
    args = (_hip.ptr(ts), _hip.ptr(te), n, 0.0, ABI_BINS, _hip.ptr(lam), _hip.ptr(mu), n_chains, model, None, 0.0,
            _hip.ptr(out), _hip.ptr(ws), ws.numel(), stream)
    keep = (lam, mu, out, ws)

    def call():
        rc = lib.lr_bd_loglik_batch(*args)
        assert rc == 0, rc
    return call, (out, lam, mu), dict(passes=-(-n_chains // cb), Cb=cb, tiles=tiles, H=H), keep


def abi_time(calls, reps):
    """average device time in ms of one ABI call: HIP events on the stream the calls enqueue on, `reps` calls back to
    back.  `calls` = one closure, or a list of closures over DIFFERENT copies of the input that are used in turn (so that
    no call re-reads what the one before it left in the Infinity Cache)."""
    import torch
    if callable(calls):
        calls = [calls]
    for c in calls[:2] + calls[:1]:
        c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        calls[i % len(calls)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def abi_rotation(ts, te):
    """copies of (ts, te) to rotate through: as many as it takes for ABI_ROTATE_BYTES to lie between two uses of one"""
    k = max(1, int(-(-ABI_ROTATE_BYTES // (16.0 * ts.numel()))))
    return [(ts, te)] + [(ts.clone(), te.clone()) for _ in range(k - 1)]


def abi_stream2_call(ts, te):
    """the yardstick: lr_debug_stream2 only READS the two arrays (csrc/lr_stats.hip)"""
    import torch
    from literate_amd import _hip
    lib = _hip.load()
    out = torch.zeros(1, dtype=torch.float64, device=ts.device)
    args = (_hip.ptr(ts), _hip.ptr(te), ts.numel(), _hip.ptr(out), _hip.stream_ptr(ts.device))

    def call():
        rc = lib.lr_debug_stream2(*args)
        assert rc == 0, rc
    return call, out


def abi_child(args):
    """--abi-child: `kernel` on n lineages, three calls, then three calls of the read-only yardstick on the same arrays,
    nothing else (the rocprofv3 --pmc / --kernel-trace target)."""
    import torch
    ts, te = abi_lineages(args.abi_n, args.abi_general, args.abi_order)
    call, _, _, keep = abi_calls(args.abi_kernel, ts, te, args.chains or 8)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    y, keep2 = abi_stream2_call(ts, te)
    for _ in range(3):
        y()
    torch.cuda.synchronize()


def abi_fetch_bytes(kernel, n, chains, general, order):
    """HBM bytes read by ONE call (all its kernels): rocprofv3 --pmc FETCH_SIZE child pass.  Returns (bytes per call with
    the guide's x 2 gfx950 correction, note, per-dispatch detail) - the detail lists every lr_* dispatch of the child with
    its raw FETCH_SIZE, and the same counter for the read-only yardstick on the same arrays (lr_debug_stream2_kernel: 16 B
    x n known bytes), which calibrates the counter for this access shape."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found", None
    d = tempfile.mkdtemp(prefix="lr_abi_pmc_", dir="/tmp")
    cmd = [rocprof, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--abi-child", "--abi-kernel", kernel, "--abi-n", str(n),
           "--chains", str(chains), "--abi-order", order] + (["--abi-general"] if general else [])
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + (("HSA_ENABLE_INTERRUPT",) if _POLLING_SET_HERE else ()):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=240)
    except (subprocess.TimeoutExpired, OSError) as ex:
        shutil.rmtree(d, ignore_errors=True)
        return None, "rocprofv3 --pmc FETCH_SIZE: %s" % type(ex).__name__, None
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += [x for x in csv.DictReader(fh) if x.get("Counter_Name") == "FETCH_SIZE"]
    shutil.rmtree(d, ignore_errors=True)
    if r.returncode != 0 or not rows:
        return None, "rocprofv3 --pmc FETCH_SIZE: rc %d, %d rows" % (r.returncode, len(rows)), None
    return abi_fetch_summary(rows, n)


def abi_fetch_summary(rows, n):
    """(bytes per call, note, detail) from the child's FETCH_SIZE rows (dicts with Kernel_Name, Counter_Value in KiB,
    Dispatch_Id).  Pure: tests/test_bench_contract.py feeds it rows."""
    per_kernel = {}
    for x in sorted(rows, key=lambda x: int(x.get("Dispatch_Id") or 0)):
        per_kernel.setdefault(x["Kernel_Name"].split("(")[0], []).append(float(x["Counter_Value"]))
    # the library's own kernels: names that START with lr_ (templates: "void lr_...").  Round 4 summed every kernel whose name
    # CONTAINED "lr_" - which __amd_rocclr_copyBuffer does: torch's 0.4 GB copy while the input is made was counted into the
    # three calls, 1/6 of a pass each: the "traffic = exactly 7/6 of the algorithmic bytes" of both kernels
    own = {k: v for k, v in per_kernel.items() if re.match(r"^(void )?lr_", k)}
    yard = [v for k, v in own.items() if "lr_debug_stream2" in k]
    call_kib = sum(sum(v) for k, v in own.items() if "lr_debug_stream2" not in k) / 3.0
    alg = 16.0 * n
    detail = {"algorithmic_bytes_per_pass": alg,
              "dispatches": {k: {"count": len(v), "FETCH_SIZE_KiB": v} for k, v in per_kernel.items()},
              "raw_bytes_per_call_over_algorithmic": call_kib * 1024.0 / alg}
    if yard and yard[0]:
        y = sum(yard[0]) / len(yard[0]) * 1024.0
        # the yardstick reads exactly 16 n bytes: alg / y is what one counted byte stands for in this access shape
        detail["yardstick_raw_bytes_over_algorithmic"] = y / alg
        detail["counter_bytes_per_counted_byte"] = alg / y
        detail["bytes_per_call_calibrated_on_yardstick"] = call_kib * 1024.0 * alg / y
        detail["traffic_over_algorithmic_calibrated"] = call_kib * 1024.0 / y
    return (2.0 * call_kib * 1024.0, "FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM), all lr_* kernels of three calls / 3; "
            "`fetch_detail` lists every dispatch and calibrates the counter on lr_debug_stream2_kernel", detail)


def abi_section(pmc=True, sizes=ABI_SIZES):
    """`abi`: the two HBM-streaming entry points at 1e7 / 3e7 / 1e8 lineages, and the cost of the calc_likelihood seam.
    Timed calls rotate through copies of the input (abi_rotation), so `ms` / `hbm_frac` are figures no Infinity Cache hit
    helps; `ms_same_buffer` is the back-to-back figure on ONE copy (what round 4 reported), and the read-only yardstick
    lr_debug_stream2 is timed the same way at every size (`stream2_GBs`, `frac_of_stream2`)."""
    import torch
    out = {"peak_GBs": HBM_PEAK_GBS, "bytes_per_lineage_pass": 16, "rotate_bytes": ABI_ROTATE_BYTES,
           "note": "achieved = 16 B x N x ceil(C / Cb) / device time of one ABI call (all its kernels; HIP events on the "
                   "call's stream; calls in turn on copies of the input totalling >= rotate_bytes); lineages = cfg4's "
                   "synthetic generator tiled to N, sorted by birth time as input files are (shuffled order beside it); "
                   "general = continuous times"}
    src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    copy_ms = abi_time(lambda: dst.copy_(src), 10)
    out["copy_GBs_measured"] = 2.0 * src.numel() / (copy_ms * 1e-3) / 1e9
    read_ms = abi_time(lambda: src.view(torch.int64).sum(), 10)
    out["read_GBs_measured_torch_sum"] = src.numel() / (read_ms * 1e-3) / 1e9
    del src, dst
    rows = []
    out["stream2_GBs"], out["stream2_same_buffer_GBs"] = {}, {}
    for n in sizes:
        for general in (False, True):
            for order in ("sorted", "shuffled"):
                if order == "shuffled" and (general or n != sizes[0]):
                    continue
                ts, te = abi_lineages(n, general, order)
                pairs = abi_rotation(ts, te)
                if n not in out["stream2_GBs"]:
                    ys = [abi_stream2_call(a, b) for a, b in pairs]
                    ms = abi_time([y[0] for y in ys], max(20, 2 * len(pairs)))
                    out["stream2_GBs"][n] = 16.0 * n / (ms * 1e-3) / 1e9
                    out["stream2_same_buffer_GBs"][n] = 16.0 * n / (abi_time(ys[0][0], 20) * 1e-3) / 1e9
                    del ys
                cases = [("lr_bin_unit_events", 0), ("lr_bin_events", 0)] + [("lr_bd_loglik_batch", c) for c in (1, 8, 16, 256)]
                for kernel, c in cases:
                    if kernel == "lr_bin_events" and (general or order != "sorted" or n != sizes[0]):
                        continue            # arbitrary windows, 8 per pass over the lineages: one row for comparison
                    made = [abi_calls(kernel, a, b, c) for a, b in pairs]
                    info = made[0][2]
                    reps = max(20, 2 * len(pairs)) if c <= 16 else max(3, len(pairs))
                    ms = abi_time([m[0] for m in made], reps)
                    ms_same = abi_time(made[0][0], min(reps, 20)) if len(pairs) > 1 else ms
                    if kernel == "lr_bin_events":
                        info["passes"] = -(-ABI_BINS // 8)      # LR_BW = 8 windows per block (csrc/lr_stats.hip)
                    gbs = 16.0 * n * info["passes"] / (ms * 1e-3) / 1e9
                    row = dict(kernel=kernel, lineages=n, general_times=general, order=order, chains=c, ms=ms,
                               ms_same_buffer=ms_same, rotated_copies=len(pairs),
                               achieved_GBs=gbs, hbm_frac=gbs / HBM_PEAK_GBS, frac_of_copy=gbs / out["copy_GBs_measured"],
                               frac_of_stream2=gbs / out["stream2_GBs"][n],
                               hbm_frac_same_buffer=16.0 * n * info["passes"] / (ms_same * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               lineages_per_s=n / (ms * 1e-3), **info)
                    if kernel == "lr_bd_loglik_batch":
                        row["evals_per_s"] = float(n) * c / (ms * 1e-3)
                    rows.append(row)
                    del made
                del ts, te, pairs
                torch.cuda.empty_cache()
    out["rows"] = rows

    # the summary objects: the one-pass figures at the LARGEST size (past the Infinity Cache), sorted unit-resolution input
    big = max(sizes)
    for kernel, key, pred in (("lr_bin_unit_events", "lr_bin_unit_events", lambda x: True),
                              ("lr_bd_loglik_batch", "lr_bd_loglik_batch", lambda x: x["chains"] == 1)):
        r = [x for x in rows if x["kernel"] == kernel and x["lineages"] == big and x["order"] == "sorted"
             and not x["general_times"] and pred(x)]
        if r:
            b = r[0]
            out[key] = dict(hbm_frac=b["hbm_frac"], achieved_GBs=b["achieved_GBs"], lineages=b["lineages"], chains=b["chains"],
                            general_times=b["general_times"], ms=b["ms"], frac_of_stream2=b["frac_of_stream2"], traffic=None)
            if pmc:
                traffic, note, detail = abi_fetch_bytes(kernel, b["lineages"], b["chains"], b["general_times"], "sorted")
                out[key]["traffic"], out[key]["traffic_note"], out[key]["fetch_detail"] = traffic, note, detail
                if traffic:
                    out[key]["traffic_over_algorithmic"] = traffic / (16.0 * b["lineages"])
                if detail and detail.get("traffic_over_algorithmic_calibrated"):
                    out[key]["traffic_over_algorithmic_calibrated"] = detail["traffic_over_algorithmic_calibrated"]
    out["engine_streaming"] = abi_engine_rows(sizes)
    out["seam"] = abi_seam()
    return out


def abi_engine_rows(sizes=ABI_SIZES, chains=16):
    """The RJMCMC loop itself on few chains x very many lineages (unit resolution), the engine sorting the lineages by
    (birth bin, death) as it does by default.  Two forms of the launch-based engine per size:
      * what the planner runs: the scan reads the PACKED lineages (lr_packscan.hip: 16 bytes per group of 14 lineages)
        once per iteration and scores every group against all chains - bound by the LDS gathers, not by HBM: us per
        iteration, evals/s, and the LDS fraction as in the headline's roofline (lds_bytes_per_eval x evals / t / LDS peak),
        for the whole iteration and for the scan kernel alone;
      * `ts_te` (engine="launch"): the scan re-reads ts / te, 16 bytes per lineage and pass of Cb chains, in every
        iteration - the HBM-bound form: 16 B x N x ceil(C / Cb) / t against the 8 TB/s peak, whole iteration and scan alone."""
    import torch
    from literate_amd.engine import ChainEngine
    rows = []
    for n in sizes:
        ts, te = abi_lineages(n, False, "sorted")
        row = dict(lineages=n, chains=chains)
        for form in ("auto", "launch"):
            eng = ChainEngine(ts, te, chains, model=0, seed=2026, s_freq=100, n_trace_slots=8, engine=form)
            eng.init()
            eng.steps(40)
            torch.cuda.synchronize()
            n_it = 100
            us = min(eng.timed_steps(n_it) for _ in range(2)) / n_it * 1e3
            r = dict(kernel=eng.kernel_name(), persistent=int(eng.layout.persistent), packed_scan=int(eng.layout.packed_scan),
                     us_per_iter=us, evals_per_s=float(n) * chains / (us * 1e-6))
            if not eng.layout.persistent:
                cb = int(eng.layout.chains_per_block)
                passes = 1 if eng.layout.packed_scan else -(-chains // cb)
                scan_us = eng.time_scan(20) * 1e3
                r.update(Cb=cb, passes=passes, scan_kernel_us=scan_us)
                if eng.layout.packed_scan:
                    # one birth + seven slot gathers of 16 bytes score 14 lineages x 2 chains (as the headline's kernel)
                    lds_b = 8 * 16 / 28.0
                    r.update(lds_bytes_per_eval=lds_b, lds_frac=r["evals_per_s"] * lds_b / 1e9 / LDS_PEAK_GBS,
                             scan_lds_frac=float(n) * chains / (scan_us * 1e-6) * lds_b / 1e9 / LDS_PEAK_GBS,
                             packed_bytes_per_lineage=16.0 / 14.0)
                else:
                    r.update(hbm_GBs=16.0 * n * passes / (us * 1e-6) / 1e9, scan_hbm_GBs=16.0 * n * passes / (scan_us * 1e-6) / 1e9)
                    r["hbm_frac"] = r["hbm_GBs"] / HBM_PEAK_GBS
                    r["scan_hbm_frac"] = r["scan_hbm_GBs"] / HBM_PEAK_GBS
            if form == "auto":
                row.update(r)
            else:
                row["ts_te"] = r
            eng.close()
            del eng
        rows.append(row)
        del ts, te
        torch.cuda.empty_cache()
    return rows


def abi_seam():
    """What the drop-in seam costs: literate_library.BDI_partial_lik / BD_lik_Keiding (the calc_likelihood operator,
    LRF:137-162, 305-308) on the shipped metal_bands lineages - wall time per call with ONE state (what a maintainer who
    only swaps the operator pays per iteration: one launch + the result polled in pinned host memory) and with 1024 states
    per call (upload, three launches, read-back, synchronisation) - beside
    the reference's binned numpy expression timed in this process (cpu leg: oracle/)."""
    import literate_library as ll
    G = np.load(os.path.join(ROOT, "tests", "golden", "binning_lik.npz"))
    ts, te = G["metal_bands/ts"], G["metal_bands/te"]
    ll.bind_lineages(ts, te, model=0)
    nb = ll.n_bins
    rng = np.random.default_rng(3)
    L1, M1 = np.exp(rng.uniform(-3, -1, nb)), np.exp(rng.uniform(-3, -1, nb))
    LC, MC = np.exp(rng.uniform(-3, -1, (1024, nb))), np.exp(rng.uniform(-3, -1, (1024, nb)))
    out = {"dataset": "metal_bands_1.tsv, %d lineages, %d bins" % (len(ts), nb)}
    for name, fn in (("BDI_partial_lik", ll.BDI_partial_lik), ("BD_lik_Keiding", ll.BD_lik_Keiding)):
        for _ in range(20):
            fn(L1, M1)
        t = time.perf_counter()
        for _ in range(300):
            fn(L1, M1)
        us1 = (time.perf_counter() - t) / 300 * 1e6
        for _ in range(3):
            fn(LC, MC)
        t = time.perf_counter()
        for _ in range(30):
            fn(LC, MC)
        usC = (time.perf_counter() - t) / 30 * 1e6
        out[name] = {"us_per_call_1_state": us1, "us_per_call_1024_states": usC,
                     "states_per_s_at_1024": 1024 / (usC * 1e-6), "evals_per_s_at_1024": 1024.0 * len(ts) / (usC * 1e-6)}
    # cpu leg: the reference's own operator is an n_bins-long numpy expression on binned statistics (oracle restatement)
    from oracle import literate_oracle as lo
    stats = dict(sp=ll.sp_events_bin, ex=ll.ex_events_bin, br=ll.br_length_bin)
    for name, model in (("BDI_partial_lik", 0), ("BD_lik_Keiding", 2)):
        with np.errstate(all="ignore"):
            for _ in range(200):
                lo.calc_likelihood(model, L1, M1, stats)
            t = time.perf_counter()
            for _ in range(5000):
                lo.calc_likelihood(model, L1, M1, stats)
        out[name]["numpy_binned_us_per_call"] = (time.perf_counter() - t) / 5000 * 1e6
        out[name]["seam_over_numpy_at_1_state"] = out[name]["us_per_call_1_state"] / out[name]["numpy_binned_us_per_call"]
        # cost(C states per call) ~ a + b (C - 1) from the two measured points, numpy = n C: equal at C = (a - b) / (n - b)
        a, n_us = out[name]["us_per_call_1_state"], out[name]["numpy_binned_us_per_call"]
        b = (out[name]["us_per_call_1024_states"] - a) / 1023.0
        out[name]["states_per_call_to_break_even"] = (a - b) / (n_us - b) if n_us > b else None
    out["note"] = ("one state per call is latency bound - for few states on few lineages lr_bd_loglik_batch is ONE launch "
                   "(lr_loglik_small_kernel) and literate_amd.ops.LoglikSession lets the kernel read the rates from pinned host "
                   "memory and write its result there, polled by the host: no copy, no stream synchronisation - and still "
                   "slower than the reference's 24..32-element numpy expression on binned statistics; swapping only the "
                   "operator pays off from states_per_call_to_break_even states per call (chains evaluated together; beyond 16 "
                   "states per call: one upload, three launches, one read-back) - a maintainer who wants the speed binds "
                   "lr_mcmc_steps, which keeps the whole loop on the device")
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks ourselves - one process per GPU
    under torch.distributed.run - BEFORE this process has touched the GPU (it never does), pass their output through and
    exit with their code.  The child command is the one the driver uses for N > 1."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def roofline_object(fig, kname, kernel_ms, n_ev, n_lin, chains, passes, cb_pass):
    """The `roofline` object of one measured kernel: LDS gather bytes against the LDS rate (what the gathers cost), the
    issue-rate view (what the loop is really bound by) and SURVEY 8(d)'s HBM conventions as side fields."""
    conv_bytes = 16.0 * n_lin * passes                          # 16 B x N x ceil(C/Cb) per launch
    conv = conv_bytes / (kernel_ms * 1e-3) / 1e9
    return {"bound": "lds", "achieved": fig["lds_GBs"], "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": fig["lds_frac"],
            "traffic": None, "traffic_note": "not measured",
            "kernel": kname, "kernel_ms": kernel_ms, "iterations_per_launch": n_ev,
            "evals_per_launch": float(n_ev) * n_lin * chains,
            "lds_bytes_per_eval": fig["lds_bytes_per_eval"], "gathers_per_eval": fig["gathers_per_eval"],
            "fp64_ops_per_eval": fig["fp64_ops_per_eval"], "aggregation": fig["aggregation"],
            "kernel_evals_per_s": fig["evals_per_s"],
            "bound_note": "LDS gather bandwidth: per (lineage, chain) the scan gathers %.2f B of lookup-table entries from "
                          "LDS (256 B/clk/CU x 256 CU x 2.4 GHz); no MFMA in a gather/scan/reduce.  Micro-benchmarks "
                          "(scratch/ubench) show the loop is bound by vector instruction issue before LDS bandwidth: see "
                          "`issue`.  HBM is not the bound (the packed lineages and the tables are L2 / LDS resident): "
                          "`hbm` holds SURVEY 8(d)'s figures, which exceed the HBM peak for that reason"
                          % fig["lds_bytes_per_eval"],
            "issue": fig.get("issue"),
            "hbm": {"peak_GBs": HBM_PEAK_GBS,
                    "algorithmic_GBs_16B_convention": conv, "frac_of_peak_16B_convention": conv / HBM_PEAK_GBS,
                    "chains_per_pass_Cb": cb_pass,
                    "effective_GBs_16B_per_eval": fig["effective_GBs_16B_per_eval"],
                    "note": "SURVEY 8(d): a lineage pass priced at 16 B (fp64 ts + te) x N x ceil(C/Cb), and un-amortised at "
                            "16 B per (lineage, chain).  The engine never re-reads ts/te: it reads them once, packs them into "
                            "table indices that stay in L2 and scores Cb chains per pass from LDS, so both figures exceed "
                            "the HBM peak; measured_GBs is what really reaches HBM"}}


# ---- what is printed: ONE compact line (the driver parses the LAST stdout line) + the detail beside it -------------
COMPACT_LIMIT = 4096          # target size of the final line; tests/test_bench_contract.py holds it under 8192


def _sig(x, digits=5):
    """floats to `digits` significant digits (the line is for reading and parsing, not for archiving)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _pick(d, *keys):
    """nested lookup that tolerates absent sections: _pick(d, 'roofline', 'issue', 'frac')"""
    for k in keys:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return None
        d = d[k]
    return d


def _abi_row(rows, n, **match):
    """the abi row of `n` lineages whose fields equal `match` (None when the section was not run)"""
    for r in rows or ():
        if r.get("lineages") == n and all(r.get(k) == v for k, v in match.items()):
            return r
    return None


def compact_line(d):
    """The final stdout line: every field the driver's contract names + `roofline` + `cpu_baseline` + one-number
    summaries of the side sections - scalars and short names only, no prose; everything else stays in the detail
    (bench_detail.json, and `#detail` lines on stderr).  Pure function of the detail dict."""
    cfg, r, c = d["config"], d["roofline"], d.get("cpu_baseline")
    out = {k: (_sig(d[k], 9) if isinstance(d[k], float) else d[k]) for k in (
        "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data")}
    out["config"] = {"workload": cfg["workload"][:110]}
    for k in ("lineages", "chains_per_gpu", "chains_total", "n_bins", "sample_every", "trace_rows_gathered_in_region",
              "process_group", "wall_over_device", "host_wait"):
        out["config"][k] = _sig(cfg.get(k))
    out["config"]["gathers_per_eval"] = _sig(r.get("gathers_per_eval"))
    out["config"]["fp64_ops_per_eval"] = _sig(r.get("fp64_ops_per_eval"))
    # the driver's parser keeps scalars of `roofline` / `cpu_baseline` only: nested figures are flattened (issue_frac, ...)
    out["roofline"] = {k: _sig(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                                   "iterations_per_launch", "lds_bytes_per_eval", "frac_engine")}
    out["roofline"].update(issue_frac=_sig(_pick(r, "issue", "frac")), hbm_measured_GBs=_sig(_pick(r, "hbm", "measured_GBs")),
                           hbm_peak_GBs=HBM_PEAK_GBS, hbm_frac_16B_convention=_sig(_pick(r, "hbm", "frac_of_peak_16B_convention")),
                           us_per_iter_device=_sig(_pick(r, "engine", "us_per_iter_device")))
    if c:
        out["cpu_baseline"] = {k: _sig(c.get(k)) for k in ("value", "unit", "cores", "kind", "cpu_model")}
        out["cpu_baseline"]["sample"] = (c.get("sample") or "")[:110]
        out["cpu_baseline"].update(all_cores_value=_sig(_pick(c, "all_cores", "value")), all_cores=_pick(c, "all_cores", "cores"),
                                   reference_loop_iters_per_s=_sig(_pick(c, "reference_loop", "iters_per_s") or c.get("iters_per_s")))
    else:
        out["cpu_baseline"] = None
    co = d.get("co_headline")
    if co:
        out["co_headline"] = {"workload": "cfg4_general", "value": _sig(co["value"]), "us_per_iter": _sig(co["us_per_iter_device"]),
                              "frac": _sig(_pick(co, "roofline", "frac"))}
    if d.get("configs"):
        out["configs"] = {n: {"us_per_iter": _sig(s["us_per_iter"]), "evals_per_s": _sig(s["evals_per_s"]), "frac": _sig(s["lds_frac"])}
                          for n, s in d["configs"].items()}
    s = d.get("strong_scaling")
    if s:
        out["strong_scaling"] = {k: _sig(s.get(k), 9) for k in ("chains_total", "chains_per_gpu", "value", "ms_per_step")}
    a = d.get("abi")
    if a:
        rows = a.get("rows")
        ab = {"peak_GBs": a.get("peak_GBs"), "stream2_GBs": {str(k): _sig(v) for k, v in (a.get("stream2_GBs") or {}).items()}}
        for key, kern, ch in (("lr_bin_unit_events", "lr_bin_unit_events", 0), ("lr_bd_loglik_batch_c1", "lr_bd_loglik_batch", 1),
                              ("lr_bd_loglik_batch_c8", "lr_bd_loglik_batch", 8), ("lr_bd_loglik_batch_c16", "lr_bd_loglik_batch", 16),
                              ("lr_bd_loglik_batch_c256", "lr_bd_loglik_batch", 256)):
            per_n = {}
            for n in sorted({x["lineages"] for x in rows or ()}):
                x = _abi_row(rows, n, kernel=kern, chains=ch, general_times=False, order="sorted")
                if x:
                    per_n["%.0e" % n] = _sig(x["hbm_frac"], 3)
            if per_n:
                ab[key] = {"hbm_frac": per_n}
        for key, ckey in (("lr_bin_unit_events", "lr_bin_unit_events"), ("lr_bd_loglik_batch", "lr_bd_loglik_batch_c1")):
            if isinstance(a.get(key), dict) and ckey in ab:
                ab[ckey]["traffic_over_algorithmic"] = _sig(a[key].get("traffic_over_algorithmic"), 4)
                ab[ckey]["traffic_over_algorithmic_calibrated"] = _sig(a[key].get("traffic_over_algorithmic_calibrated"), 4)
                ab[ckey]["frac_of_stream2"] = _sig(a[key].get("frac_of_stream2"), 3)
        # (`ts_te`: the same loop with the scan re-reading ts / te - the HBM-bound form; rounds before 5 have only that one)
        ab["engine_streaming"] = [{"lineages": x["lineages"], "chains": x["chains"], "us_per_iter": _sig(x["us_per_iter"], 4),
                                   "evals_per_s": _sig(x.get("evals_per_s"), 4), "lds_frac": _sig(x.get("lds_frac"), 3),
                                   "us_per_iter_ts_te": _sig(_pick(x, "ts_te", "us_per_iter") if "ts_te" in x else x["us_per_iter"], 4),
                                   "hbm_frac_ts_te": _sig(_pick(x, "ts_te", "hbm_frac") if "ts_te" in x else x.get("hbm_frac"), 3),
                                   "scan_hbm_frac_ts_te": _sig(_pick(x, "ts_te", "scan_hbm_frac") if "ts_te" in x else x.get("scan_hbm_frac"), 3)}
                                  for x in a.get("engine_streaming") or ()]
        ab["seam"] = {"us_per_call_1_state": _sig(_pick(a, "seam", "BDI_partial_lik", "us_per_call_1_state"), 3),
                      "numpy_us_per_call": _sig(_pick(a, "seam", "BDI_partial_lik", "numpy_binned_us_per_call"), 3)}
        out["abi"] = ab
    out["detail"] = d.get("detail_file")
    return out


def emit(detail, stream=None, detail_stream=None):
    """Write the detail (bench_detail.json; `#detail <section> <json>` lines on STDERR) and print the compact line - the ONLY
    stdout line of the run, so that it is what the driver parses whichever end of stdout it keeps."""
    stream = stream or sys.stdout
    detail_stream = detail_stream or (sys.stderr if stream is sys.stdout else stream)
    path = os.environ.get("LR_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f)
        detail["detail_file"] = os.path.relpath(path, ROOT)
    except OSError:
        detail["detail_file"] = None
    for k in ("configs", "co_headline", "strong_scaling", "cpu_baseline", "abi", "config", "roofline"):
        if detail.get(k) is not None:
            detail_stream.write("#detail %s %s\n" % (k, json.dumps(detail[k])))
    detail_stream.flush()
    line = json.dumps(compact_line(detail), separators=(",", ":"))
    assert len(line) < 2 * COMPACT_LIMIT, len(line)
    stream.write(line + "\n")
    stream.flush()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: the workload's chains on EVERY GPU; strong: that many chains in total, sharded over the ranks")
    ap.add_argument("--engine", default="auto", choices=("auto", "launch", "persistent", "persistent4"))
    ap.add_argument("--sample-every", type=int, default=100, help="trace sampling frequency (-s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the side configurations (cfg2, cfg3, cfg5, ...)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-abi", action="store_true", help="skip the `abi` section (HBM-streaming entry points at 1e7 / 3e7 lineages)")
    ap.add_argument("--abi-only", action="store_true", help="print only the `abi` section")
    ap.add_argument("--abi-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--abi-kernel", default="lr_bd_loglik_batch", help=argparse.SUPPRESS)
    ap.add_argument("--abi-n", type=int, default=ABI_SIZES[0], help=argparse.SUPPRESS)
    ap.add_argument("--abi-general", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--abi-order", default="sorted", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.abi_child:
        return abi_child(args)
    if args.abi_only:
        a = abi_section(pmc=not args.no_pmc)
        path = os.environ.get("LR_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "abi_detail.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"abi": a}, f)
        print("#detail abi " + json.dumps(a), file=sys.stderr)
        print(json.dumps({"abi": compact_line({"config": {"workload": ""}, "roofline": {}, "abi": a, **{k: None for k in (
            "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")}})["abi"]}, separators=(",", ":")))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))           # nothing above has imported torch or touched the GPU

    # The host waits for completion signals by POLLING (the ROCm runtime's HSA_ENABLE_INTERRUPT=0; set before anything
    # initialises the GPU, and only if the caller has not chosen): woken by an interrupt, torch.cuda.synchronize() returned
    # 20 - 100 us after the kernel had ended - a 140-us region at --steps 20 read wall / device 1.22 - 1.79 from run to run,
    # polling 1.18 - 1.23 (scratch/exp_sync_latency.sh).  The profiler children keep the runtime's default.
    global _POLLING_SET_HERE
    if "HSA_ENABLE_INTERRUPT" not in os.environ:
        os.environ["HSA_ENABLE_INTERRUPT"] = "0"
        _POLLING_SET_HERE = True

    import torch
    import torch.distributed as dist
    from literate_amd.dist import gather_traces as gather_rows
    from literate_amd.dist import shard_chains

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # LR_DIST_BACKEND=gloo: rehearsal of the multi-rank path on ONE GPU (all ranks on device 0, host-staged gather, no
    # teams of CUs: the ranks' kernels share the device); the real thing is one rank per GPU over RCCL
    backend = os.environ.get("LR_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if world > 1:
            os.environ.setdefault("LR_SHARED_DEVICE", "1")
    torch.cuda.set_device(local_rank)
    # a process group whenever a launcher started us (RANK set) - also for ONE rank: `torch.distributed.run
    # --nproc-per-node 1 bench.py --gpus 1` then runs the whole RCCL path (communicator, barrier, the gather of the
    # sampled rows on the device) on the one GPU; plain `python bench.py` has no group and no collective
    dist_on = world > 1 or "RANK" in os.environ
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    ts, te, model, label = make_workload(args.workload)
    n_lin = len(ts)
    base_chains = args.chains or WORKLOADS[args.workload][3]

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(scaling):
        """W warm-up + exactly K timed iterations of this rank's shard; returns the figures of the region."""
        if scaling == "weak":
            offset, chains = rank * base_chains, base_chains
            total = base_chains * world
        else:
            offset, chains = shard_chains(base_chains, world, rank)
            total = base_chains
        n_slots = (args.steps + args.warmup) // args.sample_every + 2
        eng = make_engine(args.workload, ts, te, model, chains, offset, args.sample_every, n_slots, engine=args.engine)
        # bring the device out of its idle power state before anything is measured (a cold MI355X runs the first
        # tens of milliseconds at a fraction of its clock): ~0.4 s of throw-away iterations, then a fresh init so
        # that exactly W warm-up + K timed iterations follow
        eng.init()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.4:
            eng.steps(256)
            torch.cuda.synchronize()
        eng.init()
        # the W warm-up iterations go through the SAME call as the timed ones (timing events included), in up to three
        # calls, so that the one-off costs of that path - first use of the events, first elapsed-time query, cold host
        # code - are not charged to the region
        w_left = args.warmup
        for parts in (3, 2, 1):
            if w_left > 0:
                w = max(1, w_left // parts)
                eng.timed_steps(w)
                w_left -= w
        s0 = -(-args.warmup // args.sample_every)               # trace rows the warm-up has sampled (iterations 0, s, ...)
        s1 = -(-(args.warmup + args.steps) // args.sample_every)

        def gather_traces(a, b):
            # the log-posterior rows sampled in iterations [a, b) gathered to rank 0 over RCCL / xGMI
            # (literate_amd/dist.py; the same function runs under gloo in tests/test_host_cpu.py).  A region that
            # samples nothing gathers nothing - on every rank alike
            if b > a and dist_on:
                gather_rows(eng.trace[a:b, :, :13].contiguous(), total_chains=total)
            return b - a

        if dist_on:         # untimed: sets up the RCCL communicator and loads the copy kernels (one-off costs)
            gather_rows(eng.trace[0:1, :, :13].contiguous(), total_chains=total)
        timed_call = eng.prepared_timed_steps(args.steps)      # (arguments marshalled outside the region)
        sync, clock = torch.cuda.synchronize, time.perf_counter
        barrier()
        t_begin = clock()
        # the K timed iterations, bracketed by HIP events on the launch stream too (roofline.kernel_ms for the
        # persistent engine: the timed region IS its kernel, ceil(K/4096) launches); returns when they are done
        region_kernel_ms = timed_call()
        rows = gather_traces(s0, s1)
        sync()
        elapsed = clock() - t_begin
        if dist_on:
            dist.barrier()
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # sanity: the chains ran and hold finite posteriors
        snap = eng.snapshot()
        assert np.all(snap["it"] == args.steps + args.warmup) and np.all(np.isfinite(snap["likA"]))
        assert snap["accepted"].min() > 0
        return eng, chains, total, elapsed, region_kernel_ms, rows

    eng, chains, total_chains, elapsed, region_kernel_ms, rows_gathered = timed_region(args.scaling)
    value = args.steps * n_lin * total_chains / elapsed

    out = None
    if rank == 0:
        persistent = bool(eng.layout.persistent)
        unit = bool(eng.unit_resolution)
        cb = eng.layout.chains_per_block
        kname = eng.kernel_name()
        if persistent:
            # Dominant kernel = the persistent engine kernel: ONE launch runs n_ev iterations of every chain.  Timed
            # live with HIP events recorded on its stream around the launches of the timed region itself
            # (lr_mcmc_time_steps); a launch runs at most 4096 iterations.
            launches = -(-args.steps // 4096)
            n_ev = args.steps / launches                            # iterations per launch (average)
            kernel_ms = region_kernel_ms / launches
            cb_pass = chains_per_gather(eng)                        # chains one pass over the packed lineages scores
            passes = n_ev * (-(-chains // cb_pass))                 # lineage passes per launch
        else:
            # launch-based engine: the lineage scan runs as the scan blocks of lr_fused_iter_kernel, which HIP
            # events cannot bracket launch by launch under graph replay; the SAME block body is timed live as the
            # stand-alone lr_scan_*_kernel over all chains (lr_mcmc_time_scan, back-to-back launches)
            n_ev = 1
            kernel_ms = eng.time_scan(reps=50)
            kname = ("lr_scan_unit_kernel<%d, %d>" if unit else "lr_scan_fast_kernel<%d, %d>") % (
                cb, eng.layout.table_stride // (1 if unit else 2))
            passes = -(-chains // cb)
            cb_pass = cb
        fig = kernel_figures(eng, n_lin, chains, n_ev, kernel_ms)
        roof = roofline_object(fig, kname, kernel_ms, n_ev, n_lin, chains, passes, cb_pass)
        roof["frac_engine"] = value / world * fig["lds_bytes_per_eval"] / 1e9 / LDS_PEAK_GBS
        # yardstick beside the nominal 8 TB/s: device-to-device copy of 1 GiB, read + write bytes
        src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            dst.copy_(src)
        ev1.record()
        torch.cuda.synchronize()
        roof["hbm"]["copy_GBs_measured"] = 10 * 2.0 * src.numel() / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        del src, dst
        roof["engine"] = {"persistent": fig["persistent"], "threads_per_block": fig["threads_per_block"],
                          "chains_per_block": {1: 2, 2: 4, 3: chains_per_gather(eng)}[int(eng.layout.persistent)] if persistent else cb,
                          "team_blocks": int(eng.layout.team_blocks), "table_mode": int(eng.layout.table_mode),
                          "unit_resolution_tables": unit, "us_per_iter_device": fig["us_per_iter"]}
        out = {
            "metric": "RJMCMC iters/sec x lineages (lineage-log-lik evals/s, summed over chains)",
            "value": value, "unit": "lineage-log-lik evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "shipped metal_bands_1.tsv" if args.workload == "cfg2" else "synthetic",
            "config": {"workload": "%s: %s, %d chains %s, %s" % (
                           args.workload, label, base_chains, "per GPU" if args.scaling == "weak" else "in total",
                           "DDRate sampler -m_birth 2 -m_death 2" if model == "dd" else "model_BDI %d, RJ prior on shifts" % model),
                       "lineages": n_lin, "chains_per_gpu": chains, "chains_total": total_chains,
                       "n_bins": eng.n_bins, "sample_every": args.sample_every,
                       "trace_rows_gathered_in_region": rows_gathered,
                       "process_group": (dist.get_backend() if dist_on else None),
                       "iters_per_s_per_chain": args.steps / elapsed,
                       "wall_over_device": elapsed * 1e3 / region_kernel_ms,
                       "host_wait": "polling" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupt",
                       "eval_note": "one eval = one lineage scored under one chain's rates in one iteration.  The scan does NOT "
                                    "spend two gathers per eval: lineages are sorted, a run of up to 14 lineages of one birth "
                                    "bin shares ONE gather of the birth entry (multiplied by the run count - a per-birth-bin "
                                    "event count) and two neighbouring lineages may share ONE gather of a pre-summed death "
                                    "entry: %.2f gathers and %.2f fp64 operations per eval (roofline.gathers_per_eval, "
                                    ".fp64_ops_per_eval).  Every lineage's own (birth bin, death bin) is still read from "
                                    "the packed groups in every iteration and the cost stays O(N) per chain - the full "
                                    "per-bin collapse the reference evaluates (LRF:137-162) is not taken, and the "
                                    "aggregation is frozen at this level; co_headline (cfg4 on continuous times, every "
                                    "lineage with its own in-bin fractions) is the figure without year-resolution sharing"
                                    % (fig["gathers_per_eval"], fig["fp64_ops_per_eval"])},
            "roofline": roof,
        }
    eng.close()
    del eng

    # ---- N > 1: the configuration as BASELINE.json words it (the chains in total, sharded) -----------------------
    if world > 1 and args.scaling == "weak":
        eng2, chains2, total2, elapsed2, kms2, rows2 = timed_region("strong")
        if rank == 0:
            out["strong_scaling"] = {"chains_total": total2, "chains_per_gpu": chains2,
                                     "value": args.steps * n_lin * total2 / elapsed2, "unit": out["unit"],
                                     "ms_per_step": elapsed2 / args.steps * 1e3, "kernel": eng2.kernel_name(),
                                     "kernel_ms": kms2, "trace_rows_gathered_in_region": rows2}
        eng2.close()
        del eng2

    if rank == 0 and world == 1:
        r = out["roofline"]
        if not args.no_pmc:
            traffic, note = measure_traffic(args.workload, chains, args.steps if r["iterations_per_launch"] > 1 else 64,
                                            r["kernel"], args.engine, args.sample_every)
            if traffic is not None and r["iterations_per_launch"] > 1:
                traffic *= r["iterations_per_launch"] / min(args.steps, 4096)
            r["traffic"], r["traffic_note"] = traffic, note
            if traffic is not None:
                r["hbm"]["measured_GBs"] = traffic / (r["kernel_ms"] * 1e-3) / 1e9
                r["hbm"]["measured_frac_of_peak"] = r["hbm"]["measured_GBs"] / HBM_PEAK_GBS
        if not args.no_configs:
            cfgs = {}
            for name in ("cfg2", "cfg3", "cfg5", "cfg4_general", "cfg4_shard128"):
                if name == args.workload:
                    continue
                # (3000 iterations of warm-up: the chains start with one rate per process and the cost of a move grows with
                # the number of shifts - by then K has reached its stationary range)
                cfgs[name] = side_config(name, 2000, 3000)
            out["configs"] = cfgs
            if "cfg4_general" in cfgs:
                # the co-headline: the same workload on continuous times, where no two lineages share a fraction and
                # pairs form only inside a death bin - 2000 iterations, device time
                g = cfgs["cfg4_general"]
                ms = g["us_per_iter"] * g["steps"] * 1e-3
                out["co_headline"] = {
                    "workload": g["workload"], "value": g["evals_per_s"], "unit": out["unit"], "steps": g["steps"],
                    "us_per_iter_device": g["us_per_iter"], "timing": "HIP events around one 2000-iteration launch",
                    "roofline": roofline_object(g, g["kernel"], ms, g["steps"], g["lineages"], g["chains"],
                                                g["steps"] * (-(-g["chains"] // g["chains_per_gather"])), g["chains_per_gather"])}
        if not args.no_cpu_baseline:
            if model == "dd":
                out["cpu_baseline"] = cpu_baseline_dd(ts, te, budget_s=15.0)
            else:
                e0 = make_engine(args.workload, ts, te, model, 2, 0, 100, 2)
                stats = dict(sp=e0.sp_events.cpu().numpy(), ex=e0.ex_events.cpu().numpy(), br=e0.br_length.cpu().numpy())
                t0, nb, st, en = e0.t0, e0.n_bins, e0.start_time, e0.end_time
                e0.close()
                out["cpu_baseline"] = cpu_baseline(ts, te, t0, nb, stats, st, en)
            if "configs" in out and "cfg5" in out["configs"]:
                ts5, te5, _, _ = make_workload("cfg5")
                out["configs"]["cfg5"]["cpu_baseline"] = cpu_baseline_dd(ts5, te5)
        else:
            out["cpu_baseline"] = None
        if not args.no_abi:
            out["abi"] = abi_section(pmc=not args.no_pmc)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        emit(out)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
