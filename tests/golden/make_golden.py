#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--only binning,proposals,ddrate,ratemaps,ddtraj,shipped,cfg1,trendtraj,traj,library,flags,marginal,parse]
    python tests/golden/make_chains.py <dataset> <model> <n> <s> <chains>   # long reference chains

The reference is executed unmodified: LiteRateForward.py / DDRate.py through
runpy with ``-n 0`` (module body runs, the MCMC loop does not), which yields the
CLI-local functions bound to the populated globals; literate_library is
imported directly.  Only inputs and outputs are stored (npz/json) - no
reference source text.
"""
import argparse
import contextlib
import io
import json
import os
import runpy
import shutil
import subprocess
import sys
import tempfile
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

DATASETS = {
    # name: (path relative to REF, extra CLI flags)
    "example_TBP": ("example_data/example_dataTBP.txt", ["-TBP"]),
    "example_TAD": ("example_data/example_dataTAD.txt", []),
    "metal_bands": ("example_data/metal_bands/single_run/metal_bands_1.tsv", []),
    "simulated": ("simulated_results.tsv", []),
}


def run_cli(script, data_rel, flags, workdir, tolerate=()):
    """runpy the reference script on a writable copy of the data; return its globals.
    tolerate: exception types the script's own main may die with AFTER its functions exist (DDRatev2.py:152 uses an
    undefined name in __main__); the module body is then executed in a dict that survives the exception."""
    dst = os.path.join(workdir, os.path.basename(data_rel))
    if not os.path.exists(dst):
        shutil.copy(os.path.join(REF, data_rel), dst)
    argv = [script, "-d", dst] + flags
    old_argv, old_path = sys.argv, list(sys.path)
    sys.argv = argv
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            if tolerate:
                path = os.path.join(REF, script)
                g = {"__name__": "__main__", "__file__": path}
                try:
                    with open(path) as f:
                        exec(compile(f.read(), path, "exec"), g)
                except tolerate:
                    pass
            else:
                g = runpy.run_path(os.path.join(REF, script))
    finally:
        sys.argv, sys.path[:] = old_argv, old_path
    return g


def random_state(rng, start, end, kmax=12):
    """A random valid (rates, times) pair: K in 1..kmax, interior shifts >1 apart."""
    K = int(rng.integers(1, kmax + 1))
    span = end - start
    while True:
        shifts = np.sort(rng.uniform(start, end, K - 1))
        t = np.concatenate([[start], shifts, [end]])
        if K == 1 or (np.min(np.diff(t)) > 1.0 and span > K):
            break
        K = max(1, K - 1)
    rates = np.exp(rng.uniform(np.log(0.02), np.log(1.5), K))
    return rates, t


def make_binning_and_lik(work):
    out = {}
    for name, (rel, flags) in DATASETS.items():
        rng = np.random.default_rng(abs(hash(name)) % 2**31 if False else sum(map(ord, name)))
        per_model = {}
        for model in (0, 1, 2, 3):
            g = run_cli("LiteRateForward.py", rel, flags + ["-n", "0", "-seed", "1", "-model_BDI", str(model)], work)
            if model == 0:
                out[name + "/ts"] = np.asarray(g["ts"], dtype=float)
                out[name + "/te"] = np.asarray(g["te"], dtype=float)
                out[name + "/sp"] = np.asarray(g["sp_events_bin"], dtype=np.int64)
                out[name + "/ex"] = np.asarray(g["ex_events_bin"], dtype=np.int64)
                out[name + "/br"] = np.asarray(g["br_length_bin"], dtype=float)
                out[name + "/start_end"] = np.array([g["start_time"], g["end_time"]], dtype=float)
                out[name + "/B_EMP"] = np.asarray(g["B_EMP"], dtype=float)
                out[name + "/D_EMP"] = np.asarray(g["D_EMP"], dtype=float)
                # states shared by all four models
                states = [random_state(rng, g["start_time"], g["end_time"]) + random_state(rng, g["start_time"], g["end_time"])
                          for _ in range(24)]
                # the state quoted in SURVEY.md section 8c (example TBP only makes sense there)
                if name == "example_TBP":
                    states.insert(0, (np.array([.6, .2]), np.array([0, 4.55, 24.5]),
                                      np.array([.15, .19]), np.array([0, 16.682, 24.5])))
                kmax = max(max(len(s[0]), len(s[2])) for s in states)
                S = len(states)
                KL = np.array([len(s[0]) for s in states])
                KM = np.array([len(s[2]) for s in states])
                Lr = np.zeros((S, kmax)); Lt = np.zeros((S, kmax + 1))
                Mr = np.zeros((S, kmax)); Mt = np.zeros((S, kmax + 1))
                for i, (l, tl, m, tm) in enumerate(states):
                    Lr[i, :len(l)] = l; Lt[i, :len(tl)] = tl
                    Mr[i, :len(m)] = m; Mt[i, :len(tm)] = tm
                out[name + "/state_KL"], out[name + "/state_KM"] = KL, KM
                out[name + "/state_L"], out[name + "/state_tL"] = Lr, Lt
                out[name + "/state_M"], out[name + "/state_tM"] = Mr, Mt
                n_bins = g["n_bins"]
                indL = np.zeros((S, n_bins), dtype=np.int64)
                indM = np.zeros((S, n_bins), dtype=np.int64)
                for i, (l, tl, m, tm) in enumerate(states):
                    indL[i] = g["get_rate_index"](np.floor(tl))
                    indM[i] = g["get_rate_index"](np.floor(tm))
                out[name + "/state_indL"], out[name + "/state_indM"] = indL, indM
            if model == 3:
                out[name + "/ex_dead"] = np.asarray(g["ex_events_bin_dead"], dtype=np.int64)
                out[name + "/br_dead"] = np.asarray(g["br_length_bin_dead"], dtype=float)
            liks = []
            with np.errstate(all="ignore"):
                for (l, tl, m, tm) in states:
                    iL = g["get_rate_index"](np.floor(tl))
                    iM = g["get_rate_index"](np.floor(tm))
                    liks.append(float(g["calc_likelihood"](l[iL], m[iM])))
            per_model[model] = np.array(liks)
        for model, v in per_model.items():
            out["%s/lik_model%d" % (name, model)] = v

        # library path (DDRate / create_bins): parse_ts_te + create_bins
        sys.path.insert(0, REF)
        import literate_library as lib
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            TS, TE, PRESENT, ORIGIN = lib.parse_ts_te(os.path.join(REF, rel), "-TBP" in flags, -1, -1, .5)
        # what the library's own parser produced (it differs from the CLI's on files with
        # trailing tab columns, e.g. example_dataTAD.txt: pandas sees 6 columns there)
        out[name + "/lib_ts"] = np.asarray(TS, dtype=float)
        out[name + "/lib_te"] = np.asarray(TE, dtype=float)
        for rm in (0, 1):
            o, p, nsp, nex, dt, nb, tr = lib.create_bins(ORIGIN, PRESENT, TS, TE, rm)
            out["%s/lib_bins_rm%d_sp" % (name, rm)] = np.asarray(nsp, dtype=np.int64)
            out["%s/lib_bins_rm%d_ex" % (name, rm)] = np.asarray(nex, dtype=np.int64)
            out["%s/lib_bins_rm%d_dt" % (name, rm)] = np.asarray(dt, dtype=float)
            out["%s/lib_bins_rm%d_meta" % (name, rm)] = np.array([o, p, nb], dtype=float)
        # single arbitrary windows through lib.precompute_events (non-integer edges)
        wins = []
        for _ in range(8):
            a = rng.uniform(ORIGIN - 2, PRESENT)
            b = a + rng.uniform(0.1, 9.0)
            s_, e_, br_ = lib.precompute_events(np.asarray(TS, float), np.asarray(TE, float), a, b)
            wins.append([a, b, s_, e_, br_])
        out[name + "/lib_windows"] = np.array(wins, dtype=float)
        sys.path.pop(0)
    np.savez_compressed(os.path.join(HERE, "binning_lik.npz"), **out)
    print("binning_lik.npz:", len(out), "arrays")


def make_proposals(work):
    """Proposal scorers with the draws they consumed (replayed from the same seed)."""
    g = run_cli("LiteRateForward.py", DATASETS["example_TBP"][0], ["-TBP", "-n", "0", "-seed", "1"], work)
    sys.path.insert(0, REF)
    import literate_library as lib
    rng = np.random.default_rng(7)
    recs = {"add": [], "remove": [], "mult": [], "mult_scalar": [], "rjmcmc": []}
    start, end = g["start_time"], g["end_time"]
    for case in range(40):
        rates, times = random_state(rng, start, end, kmax=8)
        if case == 0:
            rates, times = np.array([.6, .2]), np.array([0, 4.55, 24.5])
        seed = 5 if case == 0 else int(rng.integers(0, 2**31 - 1))
        # add
        np.random.seed(seed)
        r_, t_, s_ = g["add_shift_RJ_weighted_mean"](rates, times)
        np.random.seed(seed)
        ind = np.random.choice(range(len(times) - 1))
        delta = np.random.uniform(0, np.diff(times)[ind])
        u = np.random.beta(10., 10.)
        recs["add"].append(dict(seed=seed, rates=rates.tolist(), times=times.tolist(), ind=int(ind), delta=float(delta),
                                u=float(u), out_rates=r_.tolist(), out_times=t_.tolist(), score=float(s_)))
        # remove (needs K>1)
        if len(rates) > 1:
            np.random.seed(seed)
            r_, t_, s_ = g["remove_shift_RJ_weighted_mean"](rates, times)
            np.random.seed(seed)
            idx = np.random.choice(range(1, len(times) - 1))
            recs["remove"].append(dict(seed=seed, rates=rates.tolist(), times=times.tolist(), idx=int(idx),
                                       out_rates=r_.tolist(), out_times=t_.tolist(), score=float(s_)))
        # vector multiplier (CLI-local and library versions are the same function)
        for f in (0.75, 1.0, 0.3):
            np.random.seed(seed)
            q_, h_ = g["update_multiplier_freq"](rates, f=f)
            np.random.seed(seed)
            q2_, h2_ = lib.update_multiplier_proposal_vec(rates, 1.1, f)
            assert np.array_equal(q_, q2_) and h_ == h2_
            np.random.seed(seed)
            ff = np.random.binomial(1, f, np.shape(rates))
            uu = np.random.uniform(0, 1, np.shape(rates))
            recs["mult"].append(dict(seed=seed, q=rates.tolist(), f=f, ff=ff.tolist(), u=uu.tolist(),
                                     out=q_.tolist(), hastings=float(h_)))
        np.random.seed(seed)
        q_, h_ = lib.update_multiplier_proposal(rates[0], 1.1)
        np.random.seed(seed)
        u1 = np.random.random()
        recs["mult_scalar"].append(dict(seed=seed, q=float(rates[0]), u=float(u1), out=float(q_), hastings=float(h_)))
        # full RJ dispatcher
        ratesM, timesM = random_state(rng, start, end, kmax=8)
        np.random.seed(seed)
        nL, ntL, nM, ntM, hq, upd = g["RJMCMC"]([rates, ratesM, times, timesM], 0.5)
        np.random.seed(seed)
        r = np.random.random(2)
        recs["rjmcmc"].append(dict(L=rates.tolist(), M=ratesM.tolist(), tL=times.tolist(), tM=timesM.tolist(),
                                   seed=seed, r=r.tolist(), out_L=np.asarray(nL).tolist(), out_tL=np.asarray(ntL).tolist(),
                                   out_M=np.asarray(nM).tolist(), out_tM=np.asarray(ntM).tolist(),
                                   score=float(hq), update_L=int(upd)))
    # priors on grids
    pri = {"gamma": [], "poisson": [], "sym_beta": []}
    for case in range(30):
        x = np.exp(rng.uniform(np.log(1e-3), np.log(20), int(rng.integers(1, 9))))
        b = float(np.exp(rng.uniform(np.log(0.05), np.log(30))))
        pri["gamma"].append(dict(x=x.tolist(), a=2.0, b=b, out=float(g["prior_gamma"](x, 2.0, b))))
    pri["gamma"].append(dict(x=[0.7], a=2.0, b=2.0, out=float(g["prior_gamma"](np.array([0.7])))))
    for k in range(1, 33):
        for rate in (1, 0.37, 5.96):
            pri["poisson"].append(dict(k=k, rate=rate, out=float(g["Poisson_prior"](k, rate))))
    for x in np.linspace(0.02, 0.98, 25):
        pri["sym_beta"].append(dict(x=float(x), a=10.0, out=float(g["prior_sym_beta"](x, 10.))))
    # adequacy statistic and HPD
    adq = []
    for case in range(6):
        eb, ed = rng.uniform(0.05, 1, 24), rng.uniform(0.05, 1, 24)
        sb, sd = rng.uniform(0.05, 1, 24), rng.uniform(0.05, 1, 24)
        adq.append(dict(eb=eb.tolist(), ed=ed.tolist(), sb=sb.tolist(), sd=sd.tolist(),
                        out=[float(v) for v in lib.calculate_r_squared(eb, ed, sb, sd)]))
    hpd = []
    for case in range(6):
        d = rng.gamma(2, 1, int(rng.integers(20, 400)))
        hpd.append(dict(d=d.tolist(), out=lib.calcHPD(d, 0.95).tolist()))
    sys.path.pop(0)
    with open(os.path.join(HERE, "proposals_priors.json"), "w") as f:
        json.dump(dict(proposals=recs, priors=pri, adequacy=adq, hpd=hpd), f)
    print("proposals_priors.json written")


def make_ddrate(work):
    out = {}
    rng = np.random.default_rng(11)
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, 0), (1, 2)):
        g = run_cli("DDRate.py", DATASETS["metal_bands"][0],
                    ["-n", "0", "-seed", "3", "-m_birth", str(mb), "-m_death", str(md)], work)
        key = "mb%d_md%d" % (mb, md)
        if "N_SPEC" not in out:
            out["N_SPEC"] = np.asarray(g["N_SPEC"], dtype=np.int64)
            out["N_EXTI"] = np.asarray(g["N_EXTI"], dtype=np.int64)
            out["DT"] = np.asarray(g["DT"], dtype=float)
            out["TIME_RANGE"] = np.asarray(g["TIME_RANGE"], dtype=float)
            out["meta"] = np.array([g["ORIGIN"], g["PRESENT"], g["PRIOR_K0_L"]], dtype=float)
        A, LK, BR, DR, NI, NF, PR = [], [], [], [], [], [], []
        for case in range(20):
            args = np.array([np.exp(rng.uniform(np.log(.05), np.log(1.5))), rng.normal(0, 1.5),
                             rng.uniform(0, 40), rng.uniform(1, 200), rng.uniform(500, 40000),
                             np.exp(rng.uniform(np.log(.02), np.log(.6))), abs(rng.normal(1, .5)) + .05,
                             abs(rng.normal(1, .5)) + .05])
            if case == 0:
                args = np.array([0.5, 1.5, 16.25, 10, 20000, 0.1, 1., 1.])
            with np.errstate(all="ignore"):
                lik, br, dr, ni, nf = g["likelihood_function"](args)
                pr = g["calc_prior"](args)
            A.append(args); LK.append(lik); BR.append(br); DR.append(dr); NI.append(ni); NF.append(nf); PR.append(pr)
        out[key + "/args"] = np.array(A); out[key + "/lik"] = np.array(LK)
        out[key + "/birth"] = np.array(BR); out[key + "/death"] = np.array(DR)
        out[key + "/niche"] = np.array(NI); out[key + "/niche_frac"] = np.array(NF)
        out[key + "/prior"] = np.array(PR, dtype=float)
    np.savez_compressed(os.path.join(HERE, "ddrate.npz"), **out)
    print("ddrate.npz:", len(out), "arrays")


def make_ratemaps(work):
    """DDRatev2.py and trend_rate.py likelihood_function / calc_prior on random parameter vectors (module bodies run
    with -n 0, like DDRate.py).  The trend column is synthetic (written here, stored in the fixture as given)."""
    out = {}
    rng = np.random.default_rng(23)
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, -1), (1, 2)):
        g = run_cli("DDRatev2.py", DATASETS["metal_bands"][0],
                    ["-n", "0", "-seed", "3", "-m_birth", str(mb), "-m_death", str(md)], work, tolerate=(NameError,))
        key = "ddv2_mb%d_md%d" % (mb, md)
        if "N_SPEC" not in out:
            out["N_SPEC"] = np.asarray(g["N_SPEC"], dtype=np.int64)
            out["N_EXTI"] = np.asarray(g["N_EXTI"], dtype=np.int64)
            out["DT"] = np.asarray(g["DT"], dtype=float)
            out["TIME_RANGE"] = np.asarray(g["TIME_RANGE"], dtype=float)
        A, LK, BR, DR, NI, NF = [], [], [], [], [], []
        for case in range(20):
            # [l_f, l_mul, k, x0, div_0, L, m_mul, nuB, nuD]
            args = np.array([np.exp(rng.uniform(np.log(.02), np.log(.8))), rng.uniform(0, 3), rng.normal(0, 1.5),
                             rng.uniform(0, 40), rng.uniform(1, 200), rng.uniform(500, 40000), rng.uniform(0, 1.3),
                             abs(rng.normal(1, .5)) + .05, abs(rng.normal(1, .5)) + .05])
            with np.errstate(all="ignore"):
                lik, br, dr, ni, nf = g["likelihood_function"](args)
            A.append(args); LK.append(lik); BR.append(br); DR.append(dr); NI.append(ni); NF.append(nf)
        out[key + "/args"] = np.array(A); out[key + "/lik"] = np.array(LK)
        out[key + "/birth"] = np.array(BR); out[key + "/death"] = np.array(DR)
        out[key + "/niche"] = np.array(NI); out[key + "/niche_frac"] = np.array(NF)
    n_rows = len(out["DT"]) + 1                      # parse_trend_data drops the last row
    raw = np.round(50 + 30 * np.sin(np.arange(n_rows) / 5.0) + rng.normal(0, 4, n_rows) + np.arange(n_rows), 3)
    trend_file = os.path.join(work, "trend.tsv")
    with open(trend_file, "w") as f:
        f.write("year\ttrend\n")
        for i, v in enumerate(raw):
            f.write("%d\t%r\n" % (i, float(v)))
    out["trend_raw"] = raw
    for cb, cd in ((0, 0), (1, 0), (0, 1)):
        flags = ["-n", "0", "-seed", "3", "-trend_data", trend_file, "-trend_index", "1"]
        flags += (["-const_B", "1"] if cb else []) + (["-const_D", "1"] if cd else [])
        g = run_cli("trend_rate.py", DATASETS["metal_bands"][0], flags, work)
        key = "trend_cb%d_cd%d" % (cb, cd)
        out["TREND"] = np.asarray(g["TREND"], dtype=float)
        A, LK, BR, DR, PR = [], [], [], [], []
        for case in range(20):
            # [l_min, m_min, alpha, beta, delta, gamma]
            args = np.array([np.exp(rng.uniform(np.log(.01), np.log(.5))), np.exp(rng.uniform(np.log(.01), np.log(.5))),
                             rng.normal(0, .4), rng.normal(0, .4), rng.gamma(3, .5) + .05, rng.gamma(3, .5) + .05])
            with np.errstate(all="ignore"):
                lik, br, dr = g["likelihood_function"](args)
                pr = g["calc_prior"](args)
            A.append(args); LK.append(lik); BR.append(np.array(br)); DR.append(np.array(dr)); PR.append(pr)
        out[key + "/args"] = np.array(A); out[key + "/lik"] = np.array(LK)
        out[key + "/birth"] = np.array(BR); out[key + "/death"] = np.array(DR)
        out[key + "/prior"] = np.array(PR, dtype=float)
    np.savez_compressed(os.path.join(HERE, "ratemaps.npz"), **out)
    print("ratemaps.npz:", len(out), "arrays")


def make_dd_trajectories(work):
    """Full DDRate.py runs (reference process, fixed seed): every sampled row of its log is kept so that the
    restated loop (oracle/dd_mcmc_oracle.py) fed the same MT19937 stream must reproduce them."""
    out = {}
    suffix_b = {0: "_LL", 1: "_LDD", 2: "_LDDN"}
    for mb, md, seed, n, s in ((2, 2, 4, 4000, 10), (1, 1, 5, 3000, 10), (0, 0, 6, 3000, 10), (2, -1, 7, 3000, 10),
                               (2, 0, 8, 3000, 10)):
        rel, flags = DATASETS["metal_bands"]
        d = tempfile.mkdtemp(dir=work)
        dst = os.path.join(d, os.path.basename(rel))
        shutil.copy(os.path.join(REF, rel), dst)
        cmd = [sys.executable, "-B", os.path.join(REF, "DDRate.py"), "-d", dst, "-n", str(n), "-s", str(s),
               "-seed", str(seed), "-m_birth", str(mb), "-m_death", str(md)] + flags
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        suffix_d = "_ML" if md <= 0 else ("_MDD" if md == 1 else "_MDDN")
        log = "%s_%s%s%s.log" % (os.path.splitext(dst)[0], seed, suffix_b[mb], suffix_d)
        rows = np.loadtxt(log, skiprows=1, ndmin=2)
        key = "mb%d_md%d_s%d" % (mb, md, seed)
        # all sampled rows: the 14 scalar columns and the 3 adequacy columns; the 4 x n_bins per-bin columns
        # (functions of the 8 parameters alone) only for the first 25 rows
        out[key + "/head"] = rows[:, :14]
        out[key + "/adequacy"] = rows[:, -3:]
        out[key + "/full25"] = rows[:25]
        out[key + "/meta"] = np.array([mb, md, seed, n, s], dtype=float)
        print(key, rows.shape)
    np.savez_compressed(os.path.join(HERE, "dd_trajectories.npz"), **out)


def make_cfg1(work):
    """BASELINE.json configs[0]: example_dataTBP, ONE chain, fixed 2 rate shifts.  The CLI cannot start with
    shifts (SURVEY 8c, config-1 note), so the reference's own runMCMC (LRF:216-373) is called through the runpy
    harness with a 3-rate initial state and -const_rates 1; its three logs are kept row by row."""
    d = tempfile.mkdtemp(dir=work)
    seed, n, s_freq = 19, 6000, 10
    g = run_cli("LiteRateForward.py", DATASETS["example_TBP"][0],
                ["-TBP", "-n", "0", "-seed", str(seed), "-const_rates", "1", "-s", str(s_freq)], d)
    fn = g["runMCMC"]
    G_ = fn.__globals__
    G_["n_iterations"] = n
    start, end = G_["start_time"], G_["end_time"]
    t = np.linspace(start, end, 4)
    L0, M0 = np.array([0.4, 0.15, 0.6]), np.array([0.1, 0.3, 0.2])
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fn([L0, M0, t + 0., t + 0.])
    for name in ("mcmc_logfile", "sp_logfile", "ex_logfile"):
        if name in G_:
            G_[name].flush()
    logdir = os.path.join(d, "literate_mcmc_logs")
    stem = os.path.splitext(os.path.basename(DATASETS["example_TBP"][0]))[0] + "_BD"
    mc, sp, ex = parse_logs(logdir, stem)
    out = {"mcmc": mc, "sp": pack_rows(sp), "ex": pack_rows(ex), "meta": np.array([seed, n, s_freq], dtype=float),
           "L0": L0, "M0": M0, "times0": t}
    np.savez_compressed(os.path.join(HERE, "cfg1_fixed_shifts.npz"), **out)
    print("cfg1_fixed_shifts.npz:", mc.shape)


def make_trend_trajectories(work):
    """Full trend_rate.py runs (reference process, fixed seed) on metal_bands with the synthetic trend column of
    ratemaps.npz: every sampled log row."""
    R = np.load(os.path.join(HERE, "ratemaps.npz"))
    out = {}
    for cb, cd, seed, n, s in ((0, 0, 4, 3000, 10), (1, 0, 5, 2000, 10), (0, 1, 6, 2000, 10)):
        rel, flags = DATASETS["metal_bands"]
        d = tempfile.mkdtemp(dir=work)
        dst = os.path.join(d, os.path.basename(rel))
        shutil.copy(os.path.join(REF, rel), dst)
        trend_file = os.path.join(d, "trend.tsv")
        with open(trend_file, "w") as f:
            f.write("year\ttrend\n")
            for i, v in enumerate(R["trend_raw"]):
                f.write("%d\t%r\n" % (i, float(v)))
        cmd = [sys.executable, "-B", os.path.join(REF, "trend_rate.py"), "-d", dst, "-n", str(n), "-s", str(s),
               "-seed", str(seed), "-trend_data", trend_file, "-trend_index", "1"] + flags
        cmd += (["-const_B", "1"] if cb else []) + (["-const_D", "1"] if cd else [])
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        suffix = ("_CONB" if cb else "_EXPB") + ("_COND" if cd else "_EXPD")
        log = "%s_%s%s_%s.trendrate.log" % (os.path.splitext(dst)[0], seed, suffix, 1)
        rows = np.loadtxt(log, skiprows=1, ndmin=2)
        key = "cb%d_cd%d_s%d" % (cb, cd, seed)
        out[key + "/head"] = rows[:, :12]
        out[key + "/adequacy"] = rows[:, -3:]
        out[key + "/full25"] = rows[:25]
        out[key + "/meta"] = np.array([cb, cd, seed, n, s], dtype=float)
        with open(log) as f:
            out[key + "/header"] = np.array(f.readline().rstrip("\n"))
        print(key, rows.shape)
    np.savez_compressed(os.path.join(HERE, "trend_trajectories.npz"), **out)


def make_shipped(work):
    """Data files the reference ships with its example run: the single-run _div.log statistics
    and the first rows of the shipped DDRate log with its header line."""
    base = os.path.join(REF, "example_data/metal_bands/single_run")
    div = np.loadtxt(os.path.join(base, "metal_bands_1_div.log"), skiprows=1)
    np.savez_compressed(os.path.join(HERE, "shipped_metal_bands.npz"), sp=div[:, 0].astype(np.int64),
                        ex=div[:, 1].astype(np.int64), br=div[:, 2])
    log = os.path.join(base, "DD_Rate/all_bands_1_8898_LDDN_MDDN.log")
    with open(log) as f:
        header = f.readline().rstrip("\n")
    rows = np.loadtxt(log, skiprows=1)[:12]
    np.savez_compressed(os.path.join(HERE, "shipped_ddrate_log.npz"), header=np.array(header), rows=rows)
    # the shipped posterior summary of the 100-chain analysis (vectors of the R plotting script the reference ships)
    import re
    txt = open(os.path.join(REF, "example_data/metal_bands/combined_runs/metal_bands.100chains_RTT_plots.r")).read()
    summ = {}
    for name in ("time", "birth_rate", "birth_minHPD", "birth_maxHPD", "death_rate", "death_minHPD", "death_maxHPD",
                 "unique", "counts"):
        m = re.findall(r"\n" + name + r"=c\(([^)]*)\)", txt)
        summ[name] = np.array([float(x) for x in m[0].split(",")])
    np.savez_compressed(os.path.join(HERE, "shipped_metal_bands_100chains.npz"), **summ)
    print("shipped: div", div.shape, "ddrate rows", rows.shape)


def parse_logs(logdir, stem):
    mc = np.loadtxt(os.path.join(logdir, stem + "_mcmc.log"), skiprows=1, ndmin=2)
    rows = {}
    for kind in ("sp", "ex"):
        with open(os.path.join(logdir, "%s_%s_rates.log" % (stem, kind))) as f:
            rows[kind] = [np.array(l.split(), dtype=float) for l in f if l.strip()]
    return mc, rows["sp"], rows["ex"]


def pack_rows(rows):
    width = max(len(r) for r in rows)
    m = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        m[i, :len(r)] = r
    return m


def make_trajectories(work):
    """Short full CLI runs (reference process, fixed seed): every sampled row is kept so
    the oracle loop fed the same MT19937 stream must reproduce them."""
    out = {}
    runs = [("example_TBP", 0, 42, 20000, 10, []), ("example_TBP", 2, 7, 20000, 10, []),
            ("example_TBP", 1, 3, 10000, 10, []), ("example_TBP", 3, 9, 10000, 10, []),
            ("example_TBP", 0, 11, 10000, 10, ["-const_rates", "1"]),
            ("example_TBP", 0, 12, 10000, 10, ["-const_death_rate", "1"]),
            ("example_TBP", 0, 13, 10000, 10, ["-use_rate_HP", "0", "-Poisson_prior", "2.5"]),
            ("metal_bands", 2, 5, 6000, 10, [])]
    suffix = {0: "_BD", 1: "_ID", 2: "_BDk", 3: "_BDd"}
    for name, model, seed, n, s, extra in runs:
        rel, flags = DATASETS[name]
        d = tempfile.mkdtemp(dir=work)
        dst = os.path.join(d, os.path.basename(rel))
        shutil.copy(os.path.join(REF, rel), dst)
        cmd = [sys.executable, "-B", os.path.join(REF, "LiteRateForward.py"), "-d", dst, "-n", str(n), "-s", str(s),
               "-p", str(10**9), "-seed", str(seed), "-model_BDI", str(model)] + flags + extra
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        stem = os.path.splitext(os.path.basename(rel))[0] + suffix[model]
        mc, sp, ex = parse_logs(os.path.join(d, "literate_mcmc_logs"), stem)
        key = "%s_m%d_s%d%s" % (name, model, seed, "".join(extra).replace("-", "_"))
        out[key + "/mcmc"] = mc
        out[key + "/sp"] = pack_rows(sp)
        out[key + "/ex"] = pack_rows(ex)
        out[key + "/meta"] = np.array([model, seed, n, s], dtype=float)
        print(key, mc.shape)
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **out)


def make_library_surface(work):
    """The host-side names of literate_library.py that the reference's own callers import (DDRate.py:110-122, 195-207):
    priors on grids, the proposal helpers under fixed numpy seeds (outputs only; the functions consume np.random in the
    reference's call order, so a restatement seeded the same way must reproduce them bit for bit)."""
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import literate_library as ref
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(31)
    out = {"prior_gamma": [], "prior_norm": [], "prior_beta": [], "prior_sym_beta": [], "logPoisson_pmf": [],
           "approx_log_fact": [], "seeded": []}
    for _ in range(40):
        x = float(rng.uniform(-1, 30)); a = float(rng.uniform(0.5, 5)); sc = float(rng.uniform(0.1, 12)); loc = float(rng.choice([0.0, 0.001, 1.5]))
        with np.errstate(all="ignore"):
            out["prior_gamma"].append(dict(x=x, a=a, s=sc, l=loc, out=float(ref.prior_gamma(x, a, sc, loc))))
        out["prior_norm"].append(dict(x=x, l=loc, s=sc, out=float(ref.prior_norm(x, loc, sc))))
        u = float(rng.uniform(-0.2, 1.2)); b = float(rng.uniform(0.5, 12))
        with np.errstate(all="ignore"):
            out["prior_beta"].append(dict(x=u, a=a, b=b, out=float(ref.prior_beta(u, a, b))))
            out["prior_sym_beta"].append(dict(x=u, a=a, out=float(ref.prior_sym_beta(u, a))))
    xs = np.array([0.3, 2.0, 7.5])
    out["prior_gamma"].append(dict(x=xs.tolist(), a=1.0, s=10.0, l=0.0, out=ref.prior_gamma(xs, a=1, s=10, l=0).tolist()))
    out["prior_norm"].append(dict(x=xs.tolist(), l=0.0, s=1.0, out=ref.prior_norm(xs).tolist()))
    for n in (100, 150, 1000):                       # (n < 100 needs scipy.misc.factorial, gone from scipy: SURVEY 8c)
        out["approx_log_fact"].append(dict(n=n, out=float(ref.approx_log_fact(n))))
        out["logPoisson_pmf"].append(dict(x=n, l=87.5, out=float(ref.logPoisson_pmf(n, 87.5))))
    for seed in range(12):
        rec = dict(seed=seed)
        np.random.seed(seed)
        rec["sliding_win"] = float(ref.update_sliding_win(0.4 + seed, m=0, M=6.0, d=1.5))
        rec["sliding_win_m1"] = float(ref.update_sliding_win(0.98, m=0.5, M=1.0, d=0.2))
        rec["sliding_win_log"] = float(ref.update_sliding_win_log(1.05 + 0.1 * seed, m=1, M=np.e, d=0.5))
        rec["normal_nobound"] = float(ref.update_normal_nobound(1.5, d=0.2))
        v, h = ref.update_normal_nobound_vec(np.array([.1, .2, .3, .4, .5, .6]), d=0.001, f=np.array([0, 0, .5, .5, 0, 0]))
        rec["normal_nobound_vec"] = [np.asarray(v).tolist(), float(h)]
        q, U = ref.update_multiplier_proposal_vec(np.array([.5, 1.5, 3., 10., 2e4, .1, 1., 1.]), d=1.1, f=np.array([1, 1, 0, 1, 1, 1, 1, 1]) / 7.)
        rec["multiplier_vec"] = [np.asarray(q).tolist(), float(U)]
        q, U = ref.update_multiplier_proposal(0.37, d=1.2)
        rec["multiplier"] = [float(q), float(U)]
        out["seeded"].append(rec)
    with open(os.path.join(HERE, "library_surface.json"), "w") as f:
        json.dump(out, f)
    print("library surface:", {k: len(v) for k, v in out.items()})


def make_flag_paths(work):
    """CLI flag paths around the hot loop: -rev_se, -first_year / -last_year (parsed arrays and unit-bin
    statistics through the reference body with -n 0) and -pyrate_output (a short seeded run: the three logs)."""
    out = {}
    rel, flags = DATASETS["metal_bands"]
    raw = np.genfromtxt(os.path.join(REF, rel), skip_header=1)
    for tag, extra in (("last_year", ["-last_year", "2010"]), ("first_last", ["-first_year", "1900", "-last_year", "2005"]),
                       ("first_year_filtering", ["-first_year", "1985"])):
        d = tempfile.mkdtemp(dir=work)
        try:
            g = run_cli("LiteRateForward.py", rel, flags + ["-n", "0", "-seed", "1"] + extra, d)
            out["%s/ts" % tag], out["%s/te" % tag] = np.asarray(g["ts"], float), np.asarray(g["te"], float)
            out["%s/sp" % tag] = np.asarray(g["sp_events_bin"]); out["%s/ex" % tag] = np.asarray(g["ex_events_bin"])
            out["%s/br" % tag] = np.asarray(g["br_length_bin"], float)
            out["%s/error" % tag] = np.array("")
        except Exception as ex:                      # LRF:460-461 filters te with the already filtered ts
            out["%s/error" % tag] = np.array(type(ex).__name__)
        out["%s/flags" % tag] = np.array(extra)
        print(tag, str(out["%s/error" % tag]) or "ok")
    # -rev_se 1: the same TBP data with the two time columns swapped in the file
    rel_t, flags_t = DATASETS["example_TBP"]
    d = tempfile.mkdtemp(dir=work)
    src = os.path.join(REF, rel_t)
    lines = open(src).read().splitlines()
    swapped = os.path.join(d, "swapped.txt")
    rows = [l.split() for l in lines[1:] if l.strip()]
    with open(swapped, "w") as f:                    # three columns (id, te, ts): -rev_se only acts on three-column input
        f.write("species\tte\tts\n")
        for r in rows:
            f.write("\t".join([r[1], r[3], r[2]]) + "\n")
    old_ds = dict(DATASETS)
    # run_cli copies REF-relative paths: call the body directly here
    argv = ["LiteRateForward.py", "-d", swapped, "-TBP", "-rev_se", "1", "-n", "0", "-seed", "1"]
    old_argv, old_path = sys.argv, list(sys.path)
    sys.argv = argv
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            g = runpy.run_path(os.path.join(REF, "LiteRateForward.py"))
    finally:
        sys.argv, sys.path[:] = old_argv, old_path
    out["rev_se/file_cols"] = np.array([[float(r[2]), float(r[3])] for r in rows])      # (ts, te) as the shipped file has them
    out["rev_se/ts"], out["rev_se/te"] = np.asarray(g["ts"], float), np.asarray(g["te"], float)
    out["rev_se/sp"], out["rev_se/ex"] = np.asarray(g["sp_events_bin"]), np.asarray(g["ex_events_bin"])
    out["rev_se/br"] = np.asarray(g["br_length_bin"], float)
    # -pyrate_output: full reference run, logs kept
    d = tempfile.mkdtemp(dir=work)
    dst = os.path.join(d, os.path.basename(rel_t))
    shutil.copy(src, dst)
    seed, n, s_freq = 21, 3000, 10
    cmd = [sys.executable, "-B", os.path.join(REF, "LiteRateForward.py"), "-d", dst, "-n", str(n), "-s", str(s_freq),
           "-p", str(10**9), "-seed", str(seed), "-model_BDI", "0", "-pyrate_output"] + flags_t
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    stem = os.path.splitext(os.path.basename(rel_t))[0] + "_BD"
    mc, sp, ex = parse_logs(os.path.join(d, "literate_mcmc_logs"), stem)
    out["pyrate/mcmc"], out["pyrate/sp"], out["pyrate/ex"] = mc, pack_rows(sp), pack_rows(ex)
    out["pyrate/meta"] = np.array([seed, n, s_freq], dtype=float)
    out["pyrate/header"] = np.array(open(os.path.join(d, "literate_mcmc_logs", stem + "_mcmc.log")).readline().rstrip("\n"))
    print("pyrate_output", mc.shape)
    np.savez_compressed(os.path.join(HERE, "flag_paths.npz"), **out)


def make_marginal_rates(work):
    """The parity metric's definition: get_marginal_rates (plotRJforward.v3.py:92-139) executed on the shipped
    metal_bands sp / ex rate logs.  The function body is compiled from the reference file at run time (nothing of it
    is stored); the fixture holds its outputs."""
    path = os.path.join(REF, "plotRJforward.v3.py")
    src = open(path).read().splitlines()
    first = next(i for i, l in enumerate(src) if l.startswith("def get_marginal_rates("))
    last = next(i for i in range(first + 1, len(src)) if src[i].startswith("def ") or (src[i] and not src[i][0].isspace() and not src[i].startswith("#")))
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import literate_library as ref
    finally:
        sys.path.pop(0)
    ns = {"np": np, "calcHPD": ref.calcHPD}
    exec(compile("\n".join(src[first:last]), path, "exec"), ns)
    base = os.path.join(REF, "example_data/metal_bands/single_run")
    mc = np.loadtxt(os.path.join(base, "metal_bands_1_mcmc.log"), skiprows=1)
    start_age, end_age = float(np.max(mc[:, 9])), float(np.min(mc[:, 8]))      # as plotRJforward.v3.py reads them (max age, min age)
    out = {"ages": np.array([start_age, end_age])}
    for kind in ("sp", "ex"):
        f = os.path.join(base, "metal_bands_1_%s_rates.log" % kind)
        res = ns["get_marginal_rates"](f, start_age, end_age, nbins=0, burnin=0.2)
        out[kind + "/time_frames"], out[kind + "/mean"] = np.asarray(res[0], float), np.asarray(res[1], float)
        out[kind + "/hpd_lo"], out[kind + "/hpd_hi"] = np.asarray(res[2], float), np.asarray(res[3], float)
        out[kind + "/n_samples"] = np.array(res[5])
        rows = [np.array(l.split(), float) for l in open(f)]
        out[kind + "/rows"] = pack_rows(rows)
        print("marginal", kind, res[1].shape, res[5])
    np.savez_compressed(os.path.join(HERE, "marginal_rates.npz"), **out)


def make_parse_paths(work):
    """lib parse_ts_te (lib:196-229: the DDRate / trend_rate CLIs' reader) on the shipped example files and on a
    four-column variant, for every filter path: no filter, -first_year, -last_year (TBP: the reference indexes the deaths
    with a mask built from the already filtered births - as soon as the filter removes a lineage numpy raises), both.
    Inputs: the file tables; outputs: (ts, te, present, origin) or the name of the exception the reference dies with."""
    sys.path.insert(0, REF)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import literate_library as ref
    finally:
        sys.path.pop(0)
    out = {}
    cases = []
    for name, tbp, filters in (("example_TBP", True, [(-1, -1), (20, -1), (-1, 0), (-1, 5), (18, 3), (30, -1)]),
                               ("example_TAD", False, [(-1, -1), (1998, -1), (-1, 2030), (-1, 2012), (1999, 2011)]),
                               ("metal_bands", False, [(-1, -1), (1985, 2005)])):
        rel = DATASETS[name][0]
        src = os.path.join(REF, rel)
        for fy, ly in filters:
            for jitter in (0.5, 0.0):
                tag = "%s/fy%d_ly%d_j%g" % (name, fy, ly, jitter)
                try:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        ts, te, present, origin = ref.parse_ts_te(src, tbp, fy, ly, jitter)
                    out[tag + "/ts"], out[tag + "/te"] = np.asarray(ts, float), np.asarray(te, float)
                    out[tag + "/present_origin"] = np.array([present, origin], float)
                    err = ""
                except Exception as ex:
                    err = type(ex).__name__
                out[tag + "/error"] = np.array(err)
                cases.append([name, str(int(tbp)), str(fy), str(ly), repr(jitter), tag])
                print(tag, err or "ok")
    # the tables themselves (numeric columns of the shipped files), so that the test can write them back as TSV files
    import pandas as pd
    for name in ("example_TBP", "example_TAD", "metal_bands"):
        rel = DATASETS[name][0]
        t = pd.read_csv(os.path.join(REF, rel), delimiter="\t")
        out[name + "/table"] = t.to_numpy().astype(float)            # (the TAD file carries two empty trailing columns: NaN)
        out[name + "/header"] = np.array([str(c) for c in t.columns])
    # round 5: lineages with a MISSING death year (NaN): the reference's `te_years[te_years < last_year] = last_year` is a
    # strict comparison and leaves them NaN.  Three-column tables made from the example files, every 7th death blanked
    for name, base, tbp, filters in (("nan_TBP", "example_TBP", True, [(-1, -1), (-1, 0), (30, -1)]),
                                     ("nan_TAD", "example_TAD", False, [(-1, -1), (-1, 2012), (1998, 2030)])):
        tab = out[base + "/table"][:, :3].copy()
        tab[::7, 2] = np.nan
        out[name + "/table"], out[name + "/header"] = tab, out[base + "/header"][:3]
        src = os.path.join(work, name + ".tsv")
        t = pd.DataFrame(tab, columns=[str(c) for c in out[name + "/header"]])
        for c in t.columns:
            if not t[c].isna().any():
                t[c] = t[c].astype(np.int64)
        t.to_csv(src, sep="\t", index=False, na_rep="")
        for fy, ly in filters:
            tag = "%s/fy%d_ly%d_j0.5" % (name, fy, ly)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ts, te, present, origin = ref.parse_ts_te(src, tbp, fy, ly, 0.5)
                out[tag + "/ts"], out[tag + "/te"] = np.asarray(ts, float), np.asarray(te, float)
                out[tag + "/present_origin"] = np.array([present, origin], float)
                err = ""
            except Exception as ex:
                err = type(ex).__name__
            out[tag + "/error"] = np.array(err)
            cases.append([name, str(int(tbp)), str(fy), str(ly), repr(0.5), tag])
            print(tag, err or "ok")
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "parse_paths.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    work = tempfile.mkdtemp(prefix="lr_golden_")
    steps = dict(binning=make_binning_and_lik, proposals=make_proposals, ddrate=make_ddrate, ratemaps=make_ratemaps, ddtraj=make_dd_trajectories, shipped=make_shipped, cfg1=make_cfg1, trendtraj=make_trend_trajectories,
                 traj=make_trajectories, library=make_library_surface, flags=make_flag_paths, marginal=make_marginal_rates,
                 parse=make_parse_paths)
    for name, fn in steps.items():
        if args.only and name not in args.only.split(","):
            continue
        fn(work)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
