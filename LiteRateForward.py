#!/usr/bin/env python
"""LiteRateForward.py - drop-in command line of the reference's RJMCMC sampler
(/root/reference/LiteRateForward.py:376-403: same flags, defaults and log files), running
`--chains` independent chains on MI355X GPUs through literate_amd.

Additive flags (not in the reference): --chains N (total chains; sharded over ranks when started
with torch.distributed.run), --init_shifts K (start every chain with K equally spaced shifts per
process and the CLI's Gamma(2,2) rates; SURVEY.md section 8c 'config-1 note').
With --chains 1 the log file names are exactly the reference's; with more, chain k >= 0 writes
<name>_c<k>_{mcmc,sp_rates,ex_rates}.log next to the shared _div.log.
"""
import argparse
import os
import sys
import time
from warnings import warn

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('-v', action='version', version='%(prog)s')
    p.add_argument('-d', type=str, help='data file', default="", metavar="")
    p.add_argument('-n', type=int, help='n. MCMC iterations', default=10000000, metavar=10000000)
    p.add_argument('-p', type=int, help='print frequency', default=1000, metavar=1000)
    p.add_argument('-s', type=int, help='sampling frequency', default=1000, metavar=1000)
    p.add_argument('-seed', type=int, help='seed (set to -1 to make it random)', default=-1, metavar=-1)
    p.add_argument('-const_rates', type=int, help="set to: 1 for constant B/I and D rates", default=0, metavar=0)
    p.add_argument('-const_death_rate', type=int, help="set to: 1 for constant D rates", default=0, metavar=0)
    p.add_argument('-model_BDI', type=int, help='0: birth-death; 1: immigration-death; 2 birth-death (Keiding likelihood); 3 Keiding likelihood, only no extant', default=0, metavar=0)
    p.add_argument('-TBP', help='Default is AD. Include for TBP.', default=False, action='store_true')
    p.add_argument('-pyrate_output', help='Make output PyRate-compatible', default=False, action='store_true')
    p.add_argument('-first_year', type=int, help='different start of the dataset (unspecified for TBP)', default=-1, metavar=-1)
    p.add_argument('-last_year', type=int, help='different end of the dataset (unspecified for TBP)', default=-1, metavar=-1)
    p.add_argument('-death_jitter', type=float, help='amount added to death times', default=.5, metavar=.5)
    p.add_argument('-use_rate_HP', type=int, help='0: no hyper-prior on rates, 1: hyper-prior on rates', default=1, metavar=1)
    p.add_argument('-Poisson_prior', type=float, help='0: use hyper-prior on n. shifts, >0:  fixed prior on n. shifts', default=0, metavar=0)
    p.add_argument('-rm_first_bin', type=float, help='if set to 1 it removes the first time bin', default=0, metavar=0)
    p.add_argument('-calc_adequacy', type=int, help='if set to 1 calculates and log to file adequacy', default=1, metavar=1)
    p.add_argument('-update_fraction', type=float, help='', default=0.75, metavar=0.75)
    p.add_argument('-out', type=str, help='output tag', default="", metavar="")
    p.add_argument('-rev_se', type=int, help='reversed order of ts and te in input file', default=0, metavar=0)
    p.add_argument('--chains', type=int, default=1, help='total number of independent chains')
    p.add_argument('--checkpoint', type=str, default="", help='file the run is saved to after every print block and '
                   'resumed from if it exists (same data and flags; extension to the reference, which cannot resume)')
    p.add_argument('--combine', type=float, default=-1.0, help='with --chains > 1: also write COMBINED_{mcmc,sp_rates,'
                   'ex_rates,div}.log with this burn-in fraction dropped per chain (plotRJforward.v3.py combine_logs)')
    p.add_argument('--init_shifts', type=int, default=0, help='initial number of rate shifts per process')
    p.add_argument('--block', type=int, default=0, help='iterations per device window: logs are written and flushed and '
                   'the state is printed once per window, while the next one runs (default: -p rounded up to ~50000)')
    return p


def parse_data(args):
    """LRF:438-474 (np.genfromtxt path), including its filters as written."""
    t_file = np.genfromtxt(args.d, skip_header=1)
    if t_file.shape[1] == 4:
        warn('Four column (with clade) LiteRate input is deprecated. Use three columns.', FutureWarning)
        ts_years, te_years = t_file[:, 2], t_file[:, 3]
    elif args.rev_se:
        ts_years, te_years = t_file[:, 2], t_file[:, 1]
    else:
        ts_years, te_years = t_file[:, 1], t_file[:, 2]
    if args.TBP:
        true_root_age = np.max(ts_years)
        ts, te = true_root_age - ts_years, true_root_age - te_years
    else:
        true_root_age = 0
        if args.first_year != -1:
            ts_years = ts_years[ts_years >= args.first_year]
            te_years = te_years[ts_years >= args.first_year]      # LRF:460-461 (filter on the filtered array)
        if args.last_year != -1:
            ts = ts_years[ts_years <= args.last_year]
            te = te_years[ts_years <= args.last_year]
            te[te > args.last_year] = args.last_year
        else:
            ts, te = ts_years, te_years
    te = te + args.death_jitter
    return np.asarray(ts, dtype=float), np.asarray(te, dtype=float), true_root_age


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("\n\n             LiteRate - 20200206 (MI355X engine)\n")
    import torch
    import torch.distributed as dist
    from literate_amd import dist as lrd
    from literate_amd import logs, ops
    from literate_amd.engine import ChainEngine

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        backend = os.environ.get("LR_DIST_BACKEND", "nccl")      # gloo: rehearsal of the sharded run on one GPU
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    rseed = np.random.randint(0, 9999) if args.seed == -1 else args.seed
    if world > 1:
        s = torch.tensor([rseed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.broadcast(s, 0)
        rseed = int(s.item())
    np.random.seed(rseed)

    ts, te, true_root_age = parse_data(args)
    model = args.model_BDI
    offset, n_local = lrd.shard_chains(args.chains, world, rank)
    n_samples = (args.n + args.s - 1) // args.s if args.n > 0 else 0
    eng = ChainEngine(ts, te, max(n_local, 1), model=model, seed=rseed, const_rates=args.const_rates,
                      const_death_rate=args.const_death_rate, use_rate_HP=args.use_rate_HP,
                      poisson_HP=args.Poisson_prior, update_fraction=args.update_fraction, s_freq=args.s,
                      n_trace_slots=n_samples, chain_offset=offset)
    sp, ex, br = [x.cpu().numpy() for x in (eng.sp_events, eng.ex_events, eng.br_length)]
    if args.rm_first_bin:
        raise SystemExit("-rm_first_bin 1 makes the reference's own sampler fail with IndexError at the first "
                         "K>=2 proposal (SURVEY.md section 8a A13); not supported")
    if rank == 0:
        print(ex.tolist())
        print(br.sum(), range(int(np.min(ts)), int(np.max(te))))
        out_dir, paths = logs.log_paths(args.d, model, args.out)
        try:
            os.mkdir(out_dir)
        except OSError as e:
            print(e)
        logs.write_div_log(paths["div"], sp, ex, br)
    emp = None
    if args.calc_adequacy:
        with np.errstate(all="ignore"):
            emp = (sp / br, ex / br)
        if rank == 0:
            print("EMPIRICAL BIRTH RATES:"), print(emp[0]), print("EMPIRICAL DEATH RATES:"), print(emp[1])

    if args.init_shifts > 0:
        k = args.init_shifts + 1
        t = np.linspace(eng.start_time, eng.end_time, k + 1)
        rng = np.random.default_rng(rseed)
        L = [rng.gamma(2, 2, k) for _ in range(eng.n_chains)]
        M = [rng.gamma(2, 2, k) for _ in range(eng.n_chains)]
        eng.init(L, M, [t] * eng.n_chains, [t] * eng.n_chains)
    else:
        eng.init()
    done = 0
    ckpt = (args.checkpoint + (".rank%d" % rank if world > 1 else "")) if args.checkpoint else ""
    if ckpt and not ckpt.endswith(".npz"):
        ckpt += ".npz"
    if ckpt and os.path.exists(ckpt):
        eng.load(ckpt)
        done = eng.iterations
        if rank == 0:
            print("resumed from %s at iteration %d" % (ckpt, done))
    # The logs are written as the run goes (the reference flushes every sample, LRF:334-359): a window's rows leave the
    # device on a side stream - gathered over RCCL when the chains are sharded - while the next window runs.  A resumed
    # run rewrites the rows the checkpoint holds first (they are all in the workspace), then appends.  A checkpoint per
    # window: the state is copied on the device at the window's end (checkpoint_begin) and written - with the window's
    # trace rows appended to <file>.trace - behind the next window, after collect() has shown that the run is not void.
    from literate_amd.engine import TraceStreamer
    streamer = TraceStreamer(eng, args.chains, n_local)
    writer = None
    if rank == 0 and n_samples:
        writer = logs.ChainLogWriter(args.d, model, args.out, args.chains, emp, eng.n_bins, args.pyrate_output, true_root_age)
    tickets = []                     # one per marked window: its end as a checkpoint ticket (None: nothing to write)
    if done > 0:
        streamer.mark()
        tickets.append(None)         # (the rows a resumed run re-reads are the checkpoint's own)

    def flush_window():
        rows, snap, (s0, s1, its) = streamer.collect()      # (raises when the run is void: no checkpoint of it is written)
        ticket = tickets.pop(0)
        if ticket is not None:
            eng.checkpoint_write(ticket, ckpt)              # the window's end, on disk while the next window runs
        if rank == 0:
            if writer is not None:
                writer.append(rows)
            print(its, snap["likA"][0], snap["priorA"][0])
            print("\tsp.times:", snap["tL"][0]), print("\tex.times:", snap["tM"][0])
            print("\tsp.rates:", snap["L"][0]), print("\tex.rates:", snap["M"][0])
            sys.stdout.flush()

    t_start = time.time()
    block = args.block if args.block > 0 else args.p * max(1, 50000 // max(args.p, 1))
    while done < args.n:
        n = min(block, args.n - done)
        eng.steps(n)
        streamer.mark()
        done += n
        tickets.append(eng.checkpoint_begin() if ckpt else None)    # a device-side copy of the state at the window's end
        if len(streamer.pending) > 1:
            flush_window()          # the window before this one, while this one runs
    while streamer.pending:
        flush_window()
    torch.cuda.synchronize()
    eng.check_status()
    if rank == 0 and args.n > 0:
        el = time.time() - t_start
        print("%d iterations x %d chains in %.2f s (%.0f iterations/s/chain)" % (args.n, args.chains, el, args.n / el))
    wtxt = eng.warning_text()
    if wtxt:
        print(wtxt, file=sys.stderr)
    if rank == 0 and n_samples and args.combine >= 0 and args.chains > 1:
        files = [logs.log_paths(args.d, model, args.out, c)[1]["mcmc"] for c in range(args.chains)]
        logs.combine_logs(files, os.path.dirname(files[0]), args.combine)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
