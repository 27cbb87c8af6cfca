#!/usr/bin/env python3
"""DDRate.py - drop-in CLI for the reference's DDRate.py (diversity-dependent birth/death rates), with the whole
Metropolis-Hastings loop (DDRate.py:124-241) running on the MI355X for any number of independent chains.

Same flags as the reference (core_arguments lib:291-308 + -m_birth / -m_death / -fix_death, DD:22-27) and the same
log file beside the data, `<data>_<seed><model suffix>.log` (DD:135-143), one per chain (`_c<i>` appended when
--chains > 1).  Extension: --chains.  Randomness is the engine's addressed Philox stream keyed by (seed, chain), so
trajectories are reproducible but are not numpy's.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from literate_amd.literate_library import core_arguments, parse_ts_te, print_empirical_rates, set_seed  # noqa: E402


def build_parser():
    p = core_arguments()
    p.add_argument('-m_birth', type=int, help='0) use const b rates 1) DD birth 2) niche dep DD b', default=2, metavar=2)
    p.add_argument('-m_death', type=int, help='-1) fixed d rate 0) use const d rates 1) DD death 2) niche dep DD d',
                   default=2, metavar=2)
    p.add_argument('-fix_death', type=float, help='Fix death rate (with -m_death -1)', default=0.1, metavar=0.1)
    p.add_argument('--chains', type=int, default=1, help='total number of independent chains (extension)')
    p.add_argument('--block', type=int, default=0, help='iterations per device window (logs are flushed once per window; '
                   'default: -p rounded up to ~50000)')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    import torch.distributed as dist
    from literate_amd import dist as lrd
    from literate_amd.ddrate import DDRateEngine, model_suffix

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    seed = set_seed(args.seed)
    if world > 1:
        s = torch.tensor([seed], device="cuda")
        dist.broadcast(s, 0)
        seed = int(s.item())
    TS, TE, PRESENT, ORIGIN = parse_ts_te(args.d, args.TBP, args.first_year, args.last_year, args.death_jitter)
    offset, n_local = lrd.shard_chains(args.chains, world, rank)
    n_samples = (args.n + args.s - 1) // args.s if args.n > 0 else 0
    eng = DDRateEngine(np.asarray(TS, dtype=float), np.asarray(TE, dtype=float), ORIGIN, PRESENT, max(n_local, 1),
                       m_birth=args.m_birth, m_death=args.m_death, init_death=args.fix_death, seed=seed, s_freq=args.s,
                       n_trace_slots=n_samples, chain_offset=offset, rm_first_bin=int(args.rm_first_bin))
    if rank == 0:
        print(eng.origin, eng.present)
    emp = None
    with np.errstate(all="ignore"):
        if rank == 0:
            emp = print_empirical_rates(eng.n_spec, eng.n_exti, eng.DT)
        else:
            emp = (eng.n_spec / eng.DT, eng.n_exti / eng.DT)
    eng.init()
    stem = "%s_%s%s" % (os.path.splitext(args.d)[0], seed, model_suffix(args.m_birth, args.m_death))
    paths = [stem + ("" if args.chains == 1 else "_c%d" % (offset + c)) + ".log" for c in range(n_local)]
    # every rank writes the logs of its own chains, window by window while the next window runs (the reference writes,
    # flushes and fsyncs every sample: DD:225-238 / trend_rate.py:183-195)
    from literate_amd.engine import TraceStreamer
    streamer = TraceStreamer(eng, gather=False)
    for path in paths:
        eng.start_log(path)

    def flush_window():
        rows, snap, (s0, s1, its) = streamer.collect()
        with torch.cuda.stream(streamer.side):          # the per-bin log columns are recomputed on the side stream
            eng.append_logs(paths, rows, emp)
        if rank == 0:
            print(its, snap["likA"][0], snap["L"][0][:8])
            sys.stdout.flush()

    t_start, done = time.time(), 0
    block = args.block if args.block > 0 else args.p * max(1, 50000 // max(args.p, 1))
    while done < args.n:
        n = min(block, args.n - done)
        eng.steps(n)
        streamer.mark()
        done += n
        if len(streamer.pending) > 1:
            flush_window()
    while streamer.pending:
        flush_window()
    torch.cuda.synchronize()
    eng.check_status()
    if rank == 0 and args.n > 0:
        el = time.time() - t_start
        print("%d iterations x %d chains in %.2f s (%.0f iterations/s/chain)" % (args.n, args.chains, el, args.n / el))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
