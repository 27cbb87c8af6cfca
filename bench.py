#!/usr/bin/env python3
"""bench.py - RJMCMC birth-death likelihood loop on MI355X: lineage-log-lik evals/s.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one RJMCMC iteration of every chain on this GPU: one scan of the lineage arrays
scoring all pending proposals + the per-chain accept / trace / next-proposal step.  For this workload the
engine is the persistent kernel lr_persist4_kernel: one launch runs all K iterations, a 1024-thread block owns four
chains (two pairs in ping-pong: one pair's step hides under the other pair's scan).  Workload (config.workload) =
BASELINE.json configs[3] ("cfg4"): synthetic 100k lineages,
128 unit bins, 20 true shifts per process, 1024 chains per GPU; chains shard across ranks with no
data-path collective (weak scaling), lineage arrays are replicated; the sampled trace rows are
gathered over RCCL once at the end of the timed region.  Inputs are resident in HBM before timing.

value = iterations x lineages x chains / time (one unit = one lineage's contribution to one
chain's proposed-state log-likelihood: "iters/sec x lineages" of BASELINE.json, summed over chains).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (lineages, n_bins, true shifts, chains per GPU, model)
    "cfg4": (100_000, 128, 20, 1024, 0),
    "cfg3": (10_000, 128, 20, 256, 0),
    # BASELINE.json configs[1]: the shipped metal_bands lineages (30,217; the fixture holds the parsed file), 128 chains,
    # model_BDI 2 as in the reference's tutorial run
    "cfg2": (30_217, 32, 0, 128, 2),
    # BASELINE.json configs[4]: the DDRate.py sampler (model "dd": -m_birth 2 -m_death 2) on 50k lineages, 256 chains
    "cfg5": (50_000, 64, 6, 256, "dd"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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


def cpu_baseline(ts, te, t0, n_bins, stats, start_time, end_time, budget_s=10.0):
    """CPU numbers beside the GPU one, all from the numpy port under oracle/, bounded to ~25 s:
    value      : per-lineage evaluator (oracle.per_lineage_loglik: O(N) gather form of get_BDlik) on ONE core,
    all_cores  : the same evaluator, one process per host core (fresh interpreters, no GPU),
    reference_loop : the reference's own algorithm - the whole RJMCMC iteration on binned sufficient statistics,
                 O(n_bins) per iteration, one chain on one core - as iterations/s and iterations/s x lineages."""
    import multiprocessing as mp
    br = stats["br"]
    n_eval, el = _per_lineage_worker((ts, te, t0, n_bins, br, budget_s, 0))
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    out = dict(value=n_eval * len(ts) / el, unit="lineage-log-lik evals/s", cores=1, kind="port", cpu_model=cpu_model,
               sample="%d chain states x %d lineages (same synthetic lineages, model 0), %.1f s of numpy on 1 core"
                      % (n_eval, len(ts), el))
    try:
        cores = min(len(os.sched_getaffinity(0)), 16)
    except AttributeError:
        cores = min(os.cpu_count() or 1, 16)
    if cores > 1:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_per_lineage_worker, [(ts, te, t0, n_bins, br, 6.0, 100 + i) for i in range(cores)])
        out["all_cores"] = dict(value=sum(n for n, _ in res) * len(ts) / max(e for _, e in res), cores=cores,
                                sample="one process per core, 6 s each")
    from oracle import mcmc_oracle as mo
    n_it = 20000
    t_start = time.perf_counter()
    mo.run_mcmc(stats, start_time, end_time, mo.Settings(model_BDI=0),
                mo.PhiloxDraws(2026, 0), n_it, 100)
    el = time.perf_counter() - t_start
    out["reference_loop"] = dict(iters_per_s=n_it / el, iters_per_s_x_lineages=n_it / el * len(ts), cores=1,
                                 sample="%d RJMCMC iterations of one chain on binned statistics, %.1f s" % (n_it, el))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--sample-every", type=int, default=100, help="trace sampling frequency (-s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from literate_amd import synth
    from literate_amd.engine import ChainEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # LR_DIST_BACKEND=gloo: rehearsal of the multi-rank path on ONE GPU (all ranks on device 0, host-staged gather);
    # the real thing is one rank per GPU over RCCL
    backend = os.environ.get("LR_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n_lin, n_bins, n_shifts, chains, model = WORKLOADS[args.workload]
    if args.chains:
        chains = args.chains
    if args.workload == "cfg2":
        G = np.load(os.path.join(ROOT, "tests", "golden", "binning_lik.npz"))
        ts, te = G["metal_bands/ts"], G["metal_bands/te"]
        n_lin = len(ts)
    else:
        ts, te, _ = synth.make_lineages(n_lin, n_bins=n_bins, n_shifts=n_shifts, seed=0)   # same on every rank
    n_slots = (args.steps + args.warmup) // args.sample_every + 2
    if model == "dd":
        from literate_amd.ddrate import DDRateEngine
        eng = DDRateEngine(ts, te, float(ts.min()), float(te.max()), chains, m_birth=2, m_death=2, seed=2026,
                           s_freq=args.sample_every, n_trace_slots=n_slots, chain_offset=rank * chains)
    else:
        eng = ChainEngine(ts, te, chains, model=model, seed=2026, s_freq=args.sample_every, n_trace_slots=n_slots,
                          chain_offset=rank * chains)
    # bring the device out of its idle power state before anything is measured (a cold MI355X runs the first
    # tens of milliseconds at a fraction of its clock): ~0.4 s of throw-away iterations, then a fresh init so
    # that exactly W warm-up + K timed iterations follow
    eng.init()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.4:
        eng.steps(256)
        torch.cuda.synchronize()
    eng.init()
    eng.steps(args.warmup)

    from literate_amd.dist import gather_traces as gather_rows

    def gather_traces():
        # log-posterior trace rows sampled so far, gathered to rank 0 over RCCL / xGMI (literate_amd/dist.py; the same
        # function runs under gloo in tests/test_host_cpu.py)
        return gather_rows(eng.trace[:, :, :13], total_chains=chains * world)

    gather_traces()     # untimed: loads the copy kernel and sets up the RCCL communicator (one-off costs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t_begin = time.perf_counter()
    # the K timed iterations, also bracketed by HIP events on the launch stream (roofline.kernel_ms for the persistent
    # engine: the timed region IS its kernel, ceil(K/4096) launches)
    region_kernel_ms = eng.timed_steps(args.steps)
    gather_traces()
    barrier()
    elapsed = time.perf_counter() - t_begin
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the chains ran and hold finite posteriors
    snap = eng.snapshot()
    assert np.all(snap["it"] == args.steps + args.warmup) and np.all(np.isfinite(snap["likA"]))
    assert snap["accepted"].min() > 0

    if rank == 0:
        total_chains = chains * world
        value = args.steps * n_lin * total_chains / elapsed
        # ---- roofline ------------------------------------------------------------------------------------
        cb = eng.layout.chains_per_block
        n_parts, pipelined = eng.layout.n_parts, bool(eng.layout.pipelined)
        persistent = bool(eng.layout.persistent)
        unit = bool(eng.unit_resolution)
        H = eng.layout.table_stride // (1 if unit else 2)       # table half-stride the kernels are instantiated for
        # physical limiter: LDS gather rate, 256 B/clk/CU; bytes gathered per (lineage, chain) pair: 16 (unit) / 32
        lds_peak_pairs = 256 * 2.4e9 * 256 / (16 if unit else 32)
        scan_ms = eng.time_scan(reps=50)   # stand-alone tiled scan of all chains (launch-based engine's body)
        if persistent:
            # Dominant kernel = lr_persist_kernel: ONE launch runs n_ev iterations of every chain (a 512-thread
            # block owns two chains and reads the packed lineage indices, 2 B per lineage, once per iteration).
            # Timed live with HIP events recorded on its stream around the launches of the timed region itself
            # (lr_mcmc_time_steps); a launch runs at most 4096 iterations.
            launches = -(-args.steps // 4096)
            n_ev = args.steps / launches                            # iterations per launch (average)
            kernel_ms = region_kernel_ms / launches
            # layout.persistent == 2: four chains per 1024-thread block, the two pairs scanned in turn (each pair still
            # one pass over the packed indices per iteration); 1: two chains per 512-thread block
            kname = "lr_persist4_kernel<%d>" % H if eng.layout.persistent == 2 else "lr_persist_kernel<%d, %d>" % (H, eng.layout.reserved1)
            pairs_per_launch = float(n_ev) * n_lin * chains
            passes = n_ev * ((chains + 1) // 2)                     # lineage passes: one per block per iteration
            alg_bytes = 2.0 * n_lin * passes                        # bytes of lineage data the launch reads
            conv_bytes = 16.0 * n_lin * passes                      # SURVEY 8(d) convention: 16 B x N x ceil(C/Cb), Cb = 2
            ms_per_iter_ev = kernel_ms / n_ev
            cb_pass = 2
        else:
            # launch-based engine: the lineage scan runs as the scan blocks of lr_fused_iter_kernel, which HIP
            # events cannot bracket launch by launch under graph replay; the SAME block body is timed live as the
            # stand-alone lr_scan_*_kernel over all chains (lr_mcmc_time_scan, back-to-back launches)
            n_ev = 200
            ms_per_iter_ev = eng.timed_steps(n_ev) / n_ev
            kernel_ms = scan_ms
            kname = ("lr_scan_unit_kernel<%d,%d>" if unit else "lr_scan_fast_kernel<%d,%d>") % (cb, H)
            pairs_per_launch = float(n_lin) * chains
            alg_bytes = conv_bytes = 16.0 * n_lin * (-(-chains // cb))
            cb_pass = cb
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
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
        copy_gbs = 10 * 2.0 * src.numel() / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        del src, dst
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("workload") == args.workload and tj.get("chains") == chains and tj.get("kernel") == kname:
                traffic = tj.get("hbm_bytes_per_iteration", 0.0) * (n_ev if persistent else 1)
        out = {
            "metric": "RJMCMC iters/sec x lineages (lineage-log-lik evals/s, summed over chains)",
            "value": value, "unit": "lineage-log-lik evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "shipped metal_bands_1.tsv" if args.workload == "cfg2" else "synthetic",
            "config": {"workload": "%s: synthetic %d lineages, %d unit bins, %d true shifts, %d chains per GPU, "
                                   "%s" % (args.workload, n_lin, n_bins, n_shifts, chains,
                                         "DDRate sampler -m_birth 2 -m_death 2" if model == "dd"
                                         else "model_BDI %d, RJ prior on shifts" % model),
                       "lineages": n_lin, "chains_per_gpu": chains, "chains_total": total_chains,
                       "n_bins": eng.n_bins, "sample_every": args.sample_every,
                       "iters_per_s_per_chain": args.steps / elapsed},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "hbm_copy_GBs_measured": copy_gbs,
                         "kernel": kname, "kernel_ms": kernel_ms, "iterations_per_launch": n_ev if persistent else 1,
                         "pairs_per_launch": pairs_per_launch, "chains_per_pass_Cb": cb_pass,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_lineage_per_pass": 2 if persistent else 16,
                         "achieved_GBs_16B_convention": conv_bytes / (kernel_ms * 1e-3) / 1e9,
                         "effective_GBs_unamortised": 16.0 * pairs_per_launch / (kernel_ms * 1e-3) / 1e9,
                         "kernel_evals_per_s": pairs_per_launch / (kernel_ms * 1e-3),
                         "physical_bound": "lds", "lds_peak_evals_per_s": lds_peak_pairs,
                         "lds_frac_kernel": pairs_per_launch / (kernel_ms * 1e-3) / lds_peak_pairs,
                         "lds_frac_engine": value / world / lds_peak_pairs,
                         "engine": {"persistent": persistent, "chains_per_block": 2 * eng.layout.persistent if persistent else cb,
                                    "partitions_in_flight": 1 if persistent else n_parts,
                                    "device_ms_per_step_hip_events": ms_per_iter_ev,
                                    "unit_resolution_tables": unit,
                                    "tiled_scan_kernel_ms_all_chains": scan_ms}},
        }
        if not args.no_cpu_baseline and world == 1:
            if model == "dd":
                out["cpu_baseline"] = None     # the CPU legs below time the RJ sampler's path; cfg4 carries them
            else:
                stats = dict(sp=eng.sp_events.cpu().numpy(), ex=eng.ex_events.cpu().numpy(), br=eng.br_length.cpu().numpy())
                out["cpu_baseline"] = cpu_baseline(ts, te, eng.t0, eng.n_bins, stats, eng.start_time, eng.end_time)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
