#!/usr/bin/env python3
"""trend_rate.py - drop-in CLI for the reference's trend_rate.py (birth/death rates driven by an environmental trend),
with the Metropolis-Hastings loop (trend_rate.py:102-196) on the MI355X for any number of independent chains.

Same flags as the reference (core_arguments lib:291-308 + -trend_data / -trend_index / -const_B / -const_D,
trend_rate.py:33-38; the reference's -no_death switch only renames the log and forces -const_D, :42-44) and the same
log file beside the data, `<data>_<seed><model suffix>_<trend_index>.trendrate.log` (:110), one per chain (`_c<i>`
inserted when --chains > 1).  Extension: --chains.  Randomness is the engine's addressed Philox stream.
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
    p.add_argument('-trend_data', metavar='<path to trend file>', type=str,
                   help='Input trend file should be columns tab-separated with headers. No missing values.', default="")
    p.add_argument('-trend_index', type=int, help='Column of trend in trend file.', default=0, metavar=0)
    p.add_argument('-const_B', type=bool, help='F) Vary rates with trend T) Constant rates', default=False, metavar=False)
    p.add_argument('-const_D', type=bool, help='F) Vary rates with trend T) Constant rates', default=False, metavar=False)
    p.add_argument('-no_death', type=bool, help='F) Calculate death rate T) Likelihood based on births only', default=False,
                   metavar=False)
    p.add_argument('--chains', type=int, default=1, help='total number of independent chains (extension)')
    p.add_argument('--block', type=int, default=0, help='iterations per device window (logs are flushed once per window; '
                   'default: -p rounded up to ~50000)')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("\n\n             TrendRate - 20190205 (MI355X engine)\n")
    import torch
    import torch.distributed as dist
    from literate_amd import dist as lrd
    from literate_amd.trendrate import TrendRateEngine, model_suffix, parse_trend_data

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    seed = set_seed(args.seed)
    if world > 1:
        s = torch.tensor([seed], device="cuda")
        dist.broadcast(s, 0)
        seed = int(s.item())
    const_death = True if args.no_death else args.const_D          # trend_rate.py:42-44
    TS, TE, PRESENT, ORIGIN = parse_ts_te(args.d, args.TBP, args.first_year, args.last_year, args.death_jitter)
    trend = parse_trend_data(args.trend_data, args.trend_index, args.rm_first_bin)
    offset, n_local = lrd.shard_chains(args.chains, world, rank)
    n_samples = (args.n + args.s - 1) // args.s if args.n > 0 else 0
    eng = TrendRateEngine(np.asarray(TS, dtype=float), np.asarray(TE, dtype=float), ORIGIN, PRESENT, trend, max(n_local, 1),
                          const_birth=args.const_B, const_death=const_death, seed=seed, s_freq=args.s,
                          n_trace_slots=n_samples, chain_offset=offset, rm_first_bin=int(args.rm_first_bin))
    with np.errstate(all="ignore"):
        if rank == 0:
            emp = print_empirical_rates(eng.n_spec, eng.n_exti, eng.DT)
            print("TREND", eng.trend)
        else:
            emp = (eng.n_spec / eng.DT, eng.n_exti / eng.DT)
    eng.init()
    stem = "%s_%s%s" % (os.path.splitext(args.d)[0], seed, model_suffix(args.const_B, const_death, args.no_death))
    paths = ["%s%s_%s.trendrate.log" % (stem, "" if args.chains == 1 else "_c%d" % (offset + c), args.trend_index)
             for c in range(n_local)]
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
            print(its, snap["likA"][0], snap["L"][0][:6])
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
