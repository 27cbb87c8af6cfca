"""CPU-side tests of the host surface around the hot loop, against fixtures made by RUNNING THE REFERENCE
(tests/golden/make_golden.py --only library,flags,marginal):

* the literate_library.py names the reference's own callers import (priors, scalar proposals; lib:124-193);
* the CLI flag paths -rev_se, -first_year / -last_year, -pyrate_output (LRF:324-341, 446-468);
* get_marginal_rates (plotRJforward.v3.py:92-139), the definition of the posterior-parity metric;
* the reference's DDRate.py body started against THIS repo's literate_library (star import, signatures).
"""
import inspect
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def S(golden_dir):
    with open(os.path.join(golden_dir, "library_surface.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def F(golden_dir):
    return np.load(os.path.join(golden_dir, "flag_paths.npz"))


def _arr(x):
    return np.array(x) if isinstance(x, list) else x


def test_library_priors_match_reference(S):
    import literate_library as ll
    for r in S["prior_gamma"]:
        assert np.allclose(ll.prior_gamma(_arr(r["x"]), r["a"], r["s"], r["l"]), r["out"], rtol=1e-12, atol=1e-12)
    for r in S["prior_norm"]:
        assert np.allclose(ll.prior_norm(_arr(r["x"]), r["l"], r["s"]), r["out"], rtol=1e-12, atol=1e-12)
    for r in S["prior_beta"]:
        assert np.allclose(ll.prior_beta(r["x"], r["a"], r["b"]), r["out"], rtol=1e-11, atol=1e-11)
    for r in S["prior_sym_beta"]:
        assert np.allclose(ll.prior_sym_beta(r["x"], r["a"]), r["out"], rtol=1e-11, atol=1e-11)
    for r in S["approx_log_fact"]:
        assert ll.approx_log_fact(r["n"]) == pytest.approx(r["out"], rel=1e-14)
    for r in S["logPoisson_pmf"]:
        assert ll.logPoisson_pmf(r["x"], r["l"]) == pytest.approx(r["out"], rel=1e-13)
    # keyword forms as DDRate.py:112-118 writes them
    assert ll.prior_gamma(0.3, a=1, s=10, l=0) == pytest.approx(-0.03 - np.log(10.0), rel=1e-14)
    assert ll.prior_norm(0.0) == pytest.approx(-0.5 * np.log(2 * np.pi), rel=1e-14)
    # below 100 the reference's get_log_factorial needs scipy.misc.factorial (gone): log n! it is
    assert ll.get_log_factorial(5) == pytest.approx(np.log(120.0), rel=1e-14)


def test_library_scalar_proposals_replay_reference_draws(S):
    """Same np.random call order as lib:124-146: seeded alike, the outputs are the reference's bit for bit."""
    import literate_library as ll
    for r in S["seeded"]:
        np.random.seed(r["seed"])
        assert ll.update_sliding_win(0.4 + r["seed"], m=0, M=6.0, d=1.5) == r["sliding_win"]
        assert ll.update_sliding_win(0.98, m=0.5, M=1.0, d=0.2) == r["sliding_win_m1"]
        assert ll.update_sliding_win_log(1.05 + 0.1 * r["seed"], m=1, M=np.e, d=0.5) == r["sliding_win_log"]
        assert ll.update_normal_nobound(1.5, d=0.2) == r["normal_nobound"]
        v, h = ll.update_normal_nobound_vec(np.array([.1, .2, .3, .4, .5, .6]), d=0.001, f=np.array([0, 0, .5, .5, 0, 0]))
        assert v.tolist() == r["normal_nobound_vec"][0] and h == r["normal_nobound_vec"][1]


def test_library_star_import_surface():
    """`from literate_library import *` must hand over what the reference's does (lib:8-21): numpy's namespace, np,
    scipy, stats, pd, csv, random, warn, and every function DDRate.py / trend_rate.py call."""
    ns = {}
    exec("from literate_library import *", ns)
    for name in ("np", "log", "exp", "array", "sum", "mean", "argparse", "os", "sys", "csv", "random", "stats", "scipy", "warn",
                 "pd", "gamma", "f_beta", "gdtr", "gdtrix", "betainc",
                 "calcHPD", "print_R_vec", "approx_log_fact", "get_log_factorial", "random_choice", "get_br",
                 "precompute_events", "get_rate_index", "BD_lik_Keiding", "BDI_partial_lik", "update_sliding_win",
                 "update_sliding_win_log", "update_normal_nobound", "update_normal_nobound_vec", "update_poisson_proposal",
                 "update_multiplier_proposal_vec", "update_multiplier_proposal", "logPoisson_pmf", "prior_gamma", "prior_norm",
                 "prior_sym_beta", "prior_beta", "parse_ts_te", "create_bins", "print_empirical_rates",
                 "calculate_r_squared", "set_seed", "core_arguments"):
        assert name in ns, name
    assert ns["sum"] is np.sum and ns["log"] is np.log


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_signatures_match_the_reference_module():
    """Every public function of the reference's literate_library.py exists here with the same parameter names and
    defaults (the CLI-local forms live under other names: prior_gamma_LRF, Poisson_prior)."""
    import warnings
    import literate_library as ours
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_literate_library", os.path.join(REF, "literate_library.py"))
        ref = importlib.util.module_from_spec(spec)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec.loader.exec_module(ref)
    finally:
        sys.path.pop(0)
    checked = 0
    for name, fn in vars(ref).items():
        if not isinstance(fn, types.FunctionType) or fn.__module__ != "ref_literate_library":
            continue
        assert hasattr(ours, name), name
        a, b = inspect.signature(fn), inspect.signature(getattr(ours, name))
        assert list(a.parameters) == list(b.parameters), (name, a, b)
        for p in a.parameters:
            da, db = a.parameters[p].default, b.parameters[p].default
            assert (da is inspect._empty) == (db is inspect._empty), (name, p)
            if da is not inspect._empty:
                assert np.all(da == db), (name, p, da, db)
        checked += 1
    assert checked >= 25


def test_cli_parse_data_flag_paths(F, tmp_path):
    """-last_year, -first_year + -last_year, -rev_se through LiteRateForward.parse_data: the arrays the reference body
    holds after parsing; where the reference itself dies (-first_year that really filters: LRF:460-461 indexes the
    unfiltered te with the filtered mask) this CLI dies the same way."""
    import LiteRateForward as cli
    from oracle import literate_oracle as lo
    base = ["-d", os.path.join(ROOT, "tests", "golden", "_unused")]
    src = os.path.join(tmp_path, "metal.tsv")
    B = np.load(os.path.join(ROOT, "tests", "golden", "binning_lik.npz"))
    # the shipped file's (ts, te) columns, as the fixture of the binning tests holds them (te carries the 0.5 jitter)
    ts0, te0 = B["metal_bands/ts"], B["metal_bands/te"] - 0.5
    with open(src, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts0, te0)):
            f.write("%d\t%r\t%r\n" % (i, float(a), float(b)))
    for tag in ("last_year", "first_last", "first_year_filtering"):
        flags = [str(x) for x in F[tag + "/flags"]]
        args = cli.build_parser().parse_args(["-d", src] + flags)
        err = str(F[tag + "/error"])
        if err:
            with pytest.raises(IndexError):
                cli.parse_data(args)
            continue
        ts, te, root = cli.parse_data(args)
        assert np.array_equal(ts, F[tag + "/ts"]) and np.array_equal(te, F[tag + "/te"]) and root == 0
        t0, sp, ex, br = lo.bin_events_cli(ts, te)            # and the unit-bin statistics the body builds from them
        assert np.array_equal(sp, F[tag + "/sp"]) and np.array_equal(ex, F[tag + "/ex"]) and np.array_equal(br, F[tag + "/br"])
    # -rev_se 1: second and third column swapped in the file
    cols = F["rev_se/file_cols"]
    swapped = os.path.join(tmp_path, "swapped.txt")
    with open(swapped, "w") as f:
        f.write("id\tte\tts\n")
        for i, (a, b) in enumerate(cols):
            f.write("%d\t%r\t%r\n" % (i, float(b), float(a)))
    args = cli.build_parser().parse_args(["-d", swapped, "-TBP", "-rev_se", "1"])
    ts, te, root = cli.parse_data(args)
    assert np.array_equal(ts, F["rev_se/ts"]) and np.array_equal(te, F["rev_se/te"]) and root == np.max(cols[:, 0])
    t0, sp, ex, br = lo.bin_events_cli(ts, te)
    assert np.array_equal(sp, F["rev_se/sp"]) and np.array_equal(ex, F["rev_se/ex"]) and np.array_equal(br, F["rev_se/br"])
    assert base


def test_pyrate_output_logs_match_reference_run(F, golden_dir, tmp_path):
    """-pyrate_output (LRF:324-341): a seeded reference run's three logs against the log writer fed the restated loop's
    rows (the loop itself is pinned on eight other reference runs)."""
    from literate_amd import _hip, logs
    from oracle import mcmc_oracle as mo
    G = np.load(os.path.join(golden_dir, "binning_lik.npz"))
    name = "example_TBP"
    seed, n, s = [int(v) for v in F["pyrate/meta"]]
    stats = dict(sp=G[name + "/sp"], ex=G[name + "/ex"], br=G[name + "/br"])
    start, end = G[name + "/start_end"]
    emp = (G[name + "/B_EMP"], G[name + "/D_EMP"])
    np.random.seed(seed)
    with np.errstate(all="ignore"):
        out = mo.run_mcmc(stats, start, end, mo.Settings(model_BDI=0), mo.NumpyLegacyDraws(), n, s, emp=emp)
    K = _hip.LR_KMAX
    rows = np.full((len(out["mcmc"]), _hip.LR_TRACE_W), np.nan)
    for i, (m, sp, ex) in enumerate(zip(out["mcmc"], out["sp"], out["ex"])):
        rows[i, :13] = m[:13]
        kl, km = int(m[6]), int(m[7])
        rl = rows[i, 13:13 + 2 * K - 1]; rm = rows[i, 13 + 2 * K - 1:]
        rl[:kl] = sp[:kl]; rl[K:K + kl - 1] = sp[kl:]
        rm[:km] = ex[:km]; rm[K:K + km - 1] = ex[km:]
    # TBP data: true_root_age = max(ts_years); the engine's time axis is root_age - years, so start_time = 0
    true_root_age = float(np.max(F["rev_se/file_cols"][:, 0]))
    data = tmp_path / "example_dataTBP.txt"
    data.write_text("x\n")
    out_dir, paths = logs.log_paths(str(data), 0, "")
    os.mkdir(out_dir)
    logs.write_chain_logs(paths, rows, emp, len(stats["sp"]), pyrate_output=True, true_root_age=true_root_age)
    lines = open(paths["mcmc"]).read().splitlines()
    assert lines[0] == str(F["pyrate/header"])
    mc = np.array([l.split("\t") for l in lines[1:]], dtype=float)
    ref = F["pyrate/mcmc"]
    assert mc.shape == ref.shape
    assert np.array_equal(mc[:, [0, 6, 7]], ref[:, [0, 6, 7]])
    assert np.allclose(mc[:, :13], ref[:, :13], rtol=1e-9, atol=1e-9)          # root_age / death_age columns flipped
    assert np.all(mc[:, 8] == true_root_age)
    assert np.allclose(mc[:, 13:], ref[:, 13:], rtol=1e-6, atol=1e-8, equal_nan=True)
    for kind in ("sp", "ex"):
        back = [np.array(l.split(), float) for l in open(paths[kind + "_rates"])]
        R = F["pyrate/" + kind]
        for i, row in enumerate(back):
            assert np.allclose(row, R[i, :len(row)], rtol=1e-10)               # shift times as root_age - t
            assert np.all(np.isnan(R[i, len(row):]))


def test_marginal_rates_pinned_on_reference_get_marginal_rates(golden_dir):
    """logs.marginal_rates = plotRJforward.v3.py:92-139 on the shipped metal_bands rate logs: per-bin mean and 95 % HPD
    as the reference's own function returns them."""
    from literate_amd import logs
    from oracle import literate_oracle as lo
    M = np.load(os.path.join(golden_dir, "marginal_rates.npz"))
    start_age, end_age = M["ages"]
    for kind in ("sp", "ex"):
        rows = [r[~np.isnan(r)] for r in M[kind + "/rows"]]
        frames, mean, lo_, hi_, mat = logs.marginal_rates(rows, start_age, end_age)
        assert np.array_equal(frames, M[kind + "/time_frames"])
        assert np.allclose(mean, M[kind + "/mean"], rtol=1e-13)
        assert np.array_equal(lo_, M[kind + "/hpd_lo"]) and np.array_equal(hi_, M[kind + "/hpd_hi"])
        assert mat.shape[0] == int(M[kind + "/n_samples"])
        o = lo.marginal_rates_from_rows(rows, start_age, end_age)            # the oracle's copy is pinned by the same fixture
        assert np.allclose(o[0], M[kind + "/mean"], rtol=1e-13) and np.array_equal(o[1], M[kind + "/hpd_lo"])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_reference_ddrate_body_runs_on_this_library(tmp_path, monkeypatch):
    """the reference's DDRate.py, unmodified, with THIS repo's literate_library first on the
    path.  `-n 0` runs its whole body up to the sampling loop: the star import, parse_ts_te, create_bins, the initial
    likelihood_function and calc_prior (prior_gamma(x, a=, s=, l=), prior_norm).  No GPU here, so the binning kernel
    behind create_bins is stood in for by the oracle's (the checker); everything else is the product's host code."""
    import runpy
    import shutil
    import literate_library as ll
    from literate_amd import ops
    from oracle import literate_oracle as lo

    class _T:                                   # minimal tensor stand-in for what create_bins reads back
        def __init__(self, a):
            self.a = np.asarray(a)

        def cpu(self):
            return self

        def numpy(self):
            return self.a

    def fake_bin_events(ts, te, win_lo, win_hi):
        ts, te = np.asarray(ts, float), np.asarray(te, float)
        res = [lo.precompute_events(ts, te, a, b) for a, b in zip(np.asarray(win_lo, float), np.asarray(win_hi, float))]
        return (_T(np.array([r[0] for r in res], dtype=np.int64)), _T(np.array([r[1] for r in res], dtype=np.int64)),
                _T(np.array([r[2] for r in res], dtype=float)))

    monkeypatch.setattr(ops, "bin_events", fake_bin_events)
    data = os.path.join(tmp_path, "metal_bands_1.tsv")
    shutil.copy(os.path.join(REF, "example_data/metal_bands/single_run/metal_bands_1.tsv"), data)
    monkeypatch.setattr(sys, "argv", ["DDRate.py", "-d", data, "-n", "0", "-seed", "3", "-m_birth", "2", "-m_death", "2"])
    monkeypatch.setattr(sys, "path", [ROOT] + [p for p in sys.path if p != REF])
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    assert sys.modules.get("literate_library") is None or sys.modules["literate_library"].__file__.startswith(ROOT)
    g = runpy.run_path(os.path.join(REF, "DDRate.py"), run_name="__main__")
    assert g["create_bins"] is ll.create_bins and g["prior_gamma"] is ll.prior_gamma
    # the body got as far as the initial state: its statistics are the golden ones of the oracle tests
    assert g["N_TIME_BINS"] == len(g["DT"]) and g["PRIOR_K0_L"] == np.max(g["DT"])
    p = g["calc_prior"](np.array([0.5, 1.5, 5.0, 10., 20000., 0.1, 1., 1.]))
    assert np.isfinite(p)
