"""The numpy oracle against vectors produced by the real reference (tests/golden/make_golden.py)
and against the reference's own known-answer values."""
import json
import os

import numpy as np
import pytest

from oracle import literate_oracle as lo
from oracle import mcmc_oracle as mo

DATASETS = ["example_TBP", "example_TAD", "metal_bands", "simulated"]


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "binning_lik.npz"))


@pytest.fixture(scope="module")
def P(golden_dir):
    with open(os.path.join(golden_dir, "proposals_priors.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", DATASETS)
def test_cli_binning_bit_exact(G, name):
    ts, te = G[name + "/ts"], G[name + "/te"]
    t0, sp, ex, br = lo.bin_events_cli(ts, te)
    assert np.array_equal(sp, G[name + "/sp"])
    assert np.array_equal(ex, G[name + "/ex"])
    assert np.array_equal(br, G[name + "/br"])       # half-integer data: sums are exact
    assert t0 == int(G[name + "/start_end"][0])
    exd, brd = lo.bin_events_dead(ts, te, G[name + "/start_end"][1])
    assert np.array_equal(exd, G[name + "/ex_dead"])
    assert np.array_equal(brd, G[name + "/br_dead"])


@pytest.mark.parametrize("name", DATASETS)
def test_lib_create_bins_and_windows(G, name):
    ts, te = G[name + "/lib_ts"], G[name + "/lib_te"]
    for rm in (0, 1):
        o, p, nb = G["%s/lib_bins_rm%d_meta" % (name, rm)]
        origin0 = o - rm
        o2, p2, sp, ex, dt, nb2, tr = lo.create_bins(origin0, p, ts, te, rm)
        assert (o2, p2, nb2) == (o, p, nb)
        assert np.array_equal(sp, G["%s/lib_bins_rm%d_sp" % (name, rm)])
        assert np.array_equal(ex, G["%s/lib_bins_rm%d_ex" % (name, rm)])
        assert np.array_equal(dt, G["%s/lib_bins_rm%d_dt" % (name, rm)])
    for a, b, s_, e_, br_ in G[name + "/lib_windows"]:
        s2, e2, br2 = lo.precompute_events(ts, te, a, b)
        assert (s2, e2) == (s_, e_)
        assert br2 == pytest.approx(br_, rel=1e-13)


def _states(G, name):
    KL, KM = G[name + "/state_KL"], G[name + "/state_KM"]
    for i in range(len(KL)):
        yield (i, G[name + "/state_L"][i, :KL[i]], G[name + "/state_tL"][i, :KL[i] + 1],
               G[name + "/state_M"][i, :KM[i]], G[name + "/state_tM"][i, :KM[i] + 1])


@pytest.mark.parametrize("name", DATASETS)
def test_rate_index_and_binned_likelihoods(G, name):
    stats = dict(sp=G[name + "/sp"], ex=G[name + "/ex"], br=G[name + "/br"],
                 ex_dead=G[name + "/ex_dead"], br_dead=G[name + "/br_dead"])
    n_bins = len(stats["sp"])
    for i, L, tL, M, tM in _states(G, name):
        iL = lo.get_rate_index(np.floor(tL), n_bins)
        iM = lo.get_rate_index(np.floor(tM), n_bins)
        assert np.array_equal(iL, G[name + "/state_indL"][i])
        assert np.array_equal(iM, G[name + "/state_indM"][i])
        for model in (0, 1, 2, 3):
            with np.errstate(all="ignore"):
                v = lo.calc_likelihood(model, L[iL], M[iM], stats)
            ref = G["%s/lik_model%d" % (name, model)][i]
            assert v == pytest.approx(ref, rel=1e-13), (name, model, i)


@pytest.mark.parametrize("name", DATASETS)
def test_per_lineage_forms_match_reference_binned(G, name):
    """The per-lineage sum (BDIx:124-146 form) equals the reference's binned value at floored
    shift times - the reference authors' own cross-check (BDIx:365-368) - to <= 1e-9 relative."""
    ts, te = G[name + "/ts"], G[name + "/te"]
    t0 = float(int(G[name + "/start_end"][0]))
    br = G[name + "/br"]
    n_bins = len(br)
    pre = lo.lineage_bins(ts, te, t0, n_bins)
    dead = te < G[name + "/start_end"][1]
    pre_dead = lo.lineage_bins(ts[dead], te[dead], t0, n_bins)
    for i, L, tL, M, tM in _states(G, name):
        lam = L[G[name + "/state_indL"][i]]
        mu = M[G[name + "/state_indM"][i]]
        for model in (0, 1, 2):
            ref = G["%s/lik_model%d" % (name, model)][i]
            v = lo.per_lineage_loglik(ts, te, t0, lam, mu, model, br, pre=pre)
            assert v == pytest.approx(ref, rel=1e-9, abs=1e-9), (name, model, i)
            if i < 3 and len(ts) < 40000:
                d = lo.per_lineage_loglik_dense(ts, te, t0, lam, mu, model, br)
                assert d == pytest.approx(ref, rel=1e-9, abs=1e-9)
        # model 3: birth half on everybody, death half on the te<end_time subset (LRF:141-142)
        ref3 = G["%s/lik_model3" % name][i]
        # evaluate the two halves explicitly with their own exposure rates
        logB, logD, R, _ = lo.per_lineage_tables(lam, mu, 2)
        js, fs, je, fe = pre
        cumL = np.concatenate(([0.0], np.cumsum(lam)))
        SA = np.concatenate(([0.0], logB + cumL[:-1], [cumL[-1]]))
        SR = np.concatenate(([0.0], lam, [0.0]))
        EA = np.concatenate(([0.0], -cumL[:-1], [-cumL[-1]]))
        bh = np.sum(SA[js] + fs * SR[js] + EA[je] - fe * SR[je])
        jsd, fsd, jed, fed = pre_dead
        cumM = np.concatenate(([0.0], np.cumsum(mu)))
        SAm = np.concatenate(([0.0], cumM[:-1], [cumM[-1]]))
        SRm = np.concatenate(([0.0], mu, [0.0]))
        EAm = np.concatenate(([0.0], logD - cumM[:-1], [-cumM[-1]]))
        dh = np.sum(SAm[jsd] + fsd * SRm[jsd] + EAm[jed] - fed * SRm[jed])
        assert bh + dh == pytest.approx(ref3, rel=1e-9)


def test_get_BDlik_segments_equal_keiding(G):
    name = "example_TBP"
    ts, te = G[name + "/ts"], G[name + "/te"]
    for i, L, tL, M, tM in _states(G, name):
        fl, fm = np.floor(tL), np.floor(tM)
        fl[-1] = fm[-1] = len(G[name + "/sp"])      # window ends where the last unit bin ends
        v = lo.get_BDlik(ts, te, fl, L, "l") + lo.get_BDlik(ts, te, fm, M, "m")
        assert v == pytest.approx(G[name + "/lik_model2"][i], rel=1e-12)


def test_survey_known_answers(G):
    # example TBP state
    i = 0
    assert G["example_TBP/lik_model0"][i] == pytest.approx(15.824528451812753, rel=1e-14)
    assert G["example_TBP/lik_model2"][i] == pytest.approx(-352.6785362157869, rel=1e-14)


def test_notebook_worked_example_kat():
    """2_introduction_to_literate_final.ipynb:173-216: N=100 lineages, window (10,40),
    lambda=.12, mu=.04 -> B=42, D=9, S=462.68, log-lik -192.04975094421764.  The notebook's
    data are generated from a seed we do not have; the KAT here is the closed form on its
    printed sufficient statistics."""
    ll = 42 * np.log(.12) + 9 * np.log(.04) - (.12 + .04) * 462.68
    assert ll == pytest.approx(-192.04975094421764, abs=2e-3)   # S is printed to 2 decimals


def test_shipped_metal_bands_log_row_kat(golden_dir):
    """Shipped run, row 2 of metal_bands_1_mcmc.log: L=3.20027104905, M=3.86301204216 ->
    likelihood -621495.769068 with the Keiding form on the shipped _div.log statistics."""
    d = np.load(os.path.join(golden_dir, "shipped_metal_bands.npz"))
    v = lo.BD_lik_Keiding(np.full(len(d["sp"]), 3.20027104905), np.full(len(d["sp"]), 3.86301204216),
                          d["sp"], d["ex"], d["br"])
    assert v == pytest.approx(-621495.769068, rel=1e-11)


def _shipped_ddrate_rows(golden_dir):
    """(args as the sampler holds them, logged columns) of the shipped DDRate log rows; ORIGIN of that run = 1968,
    its DT is embedded as niche_i * nicheFrac_i, its N_SPEC / N_EXTI are the shipped _div.log's."""
    d = np.load(os.path.join(golden_dir, "shipped_ddrate_log.npz"))
    s = np.load(os.path.join(golden_dir, "shipped_metal_bands.npz"))
    n = len(s["sp"])
    out = []
    for r in d["rows"][1:]:
        head = r[:14]
        cols = [r[14 + k * n:14 + (k + 1) * n] for k in range(4)]
        args = np.array([head[6], head[7], head[8] - 1968.0, head[9], head[10] - head[9], head[11], head[12], head[13]])
        out.append((args, head, cols, cols[2] * cols[3]))
    return out, s, str(d["header"])


def test_shipped_ddrate_log_kat(golden_dir):
    """Shipped DDRate run (all_bands_1_8898_LDDN_MDDN.log): from each row's logged parameters the restated
    likelihood_function gives the row's own l_i, m_i, niche_i, nicheFrac_i (print precision) and, on the shipped
    _div.log statistics, its likelihood_birth / likelihood_death to all printed digits."""
    rows, s, header = _shipped_ddrate_rows(golden_dir)
    assert len(rows) == 11
    for args, head, cols, DT in rows:
        assert np.allclose(DT, s["br"], rtol=1e-6)
        lik, b, d, ni, nf = lo.dd_likelihood_function(args, s["sp"], s["ex"], s["br"], np.arange(len(DT)).astype(float), 2, 2)
        assert np.allclose(b, cols[0], rtol=2e-7) and np.allclose(d, cols[1], rtol=2e-7)
        assert np.allclose(ni, cols[2], rtol=2e-7) and np.allclose(nf, cols[3], rtol=2e-7)
        # halves evaluated on the PRINTED rates reproduce the printed halves to all digits
        lb = np.sum(np.log(cols[0]) * s["sp"] - cols[0] * s["br"])
        ld = np.sum(np.log(cols[1]) * s["ex"] - cols[1] * s["br"])
        assert lb == pytest.approx(head[3], rel=2e-12) and ld == pytest.approx(head[4], rel=2e-12)
        assert lik[0] == pytest.approx(head[3], rel=1e-7) and lik[1] == pytest.approx(head[4], rel=1e-7)
        assert head[2] == pytest.approx(head[3] + head[4], rel=1e-11) and head[1] == pytest.approx(head[2] + head[5], rel=1e-11)


def test_add_remove_multiplier_scorers(P):
    pr = P["proposals"]
    for r in pr["add"]:
        ra, ti, sc = lo.add_shift_RJ_weighted_mean(r["rates"], r["times"], r["ind"], r["delta"], r["u"])
        assert np.allclose(ra, r["out_rates"], rtol=1e-14, atol=0)
        assert np.allclose(ti, r["out_times"], rtol=1e-15, atol=0)
        assert sc == pytest.approx(r["score"], rel=1e-12, abs=1e-12)
    for r in pr["remove"]:
        ra, ti, sc = lo.remove_shift_RJ_weighted_mean(r["rates"], r["times"], r["idx"])
        assert np.allclose(ra, r["out_rates"], rtol=1e-14, atol=0)
        assert np.array_equal(ti, r["out_times"])
        assert sc == pytest.approx(r["score"], rel=1e-12, abs=1e-12)
    for r in pr["mult"]:
        q, h = lo.update_multiplier_freq(np.array(r["q"]), r["ff"], r["u"])
        assert np.array_equal(q, r["out"]) and h == r["hastings"]
    for r in pr["mult_scalar"]:
        q, h = lo.update_multiplier_proposal(r["q"], r["u"])
        assert q == r["out"] and h == r["hastings"]
    # SURVEY.md section 8c seed-5 values
    assert pr["add"][0]["score"] == pytest.approx(1.6394932653414631, rel=1e-14)
    assert pr["remove"][0]["score"] == pytest.approx(-5.487354605997915, rel=1e-14)


def test_priors_closed_forms(P):
    pri = P["priors"]
    for r in pri["gamma"]:
        assert lo.prior_gamma(r["x"], r["a"], r["b"]) == pytest.approx(r["out"], rel=1e-13, abs=1e-13)
    for r in pri["poisson"]:
        assert lo.Poisson_prior(r["k"], r["rate"]) == pytest.approx(r["out"], rel=1e-14, abs=1e-14)
    for r in pri["sym_beta"]:
        assert lo.log_beta_sym_pdf(r["x"], r["a"]) == pytest.approx(r["out"], rel=1e-12, abs=1e-12)
    for r in P["adequacy"]:
        v = lo.calculate_r_squared(np.array(r["eb"]), np.array(r["ed"]), np.array(r["sb"]), np.array(r["sd"]))
        assert np.allclose(v, r["out"], rtol=1e-10)
    for r in P["hpd"]:
        assert np.array_equal(lo.calcHPD(r["d"], 0.95), r["out"])


def test_ddrate_likelihood_and_prior(golden_dir):
    D = np.load(os.path.join(golden_dir, "ddrate.npz"))
    origin, present, k0 = D["meta"]
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, 0), (1, 2)):
        key = "mb%d_md%d" % (mb, md)
        for j, args in enumerate(D[key + "/args"]):
            with np.errstate(all="ignore"):
                lik, b, d, ni, nf = lo.dd_likelihood_function(args, D["N_SPEC"], D["N_EXTI"], D["DT"],
                                                              D["TIME_RANGE"], mb, md)
            assert np.allclose(b, D[key + "/birth"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(d, D[key + "/death"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(ni, D[key + "/niche"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(nf, D[key + "/niche_frac"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(lik, D[key + "/lik"][j], rtol=1e-12, equal_nan=True)
            assert lo.dd_calc_prior(args, k0, origin, present) == pytest.approx(D[key + "/prior"][j], rel=1e-12)


def test_ddratev2_and_trend_rate_maps(golden_dir):
    """DDRatev2.py:73-111 and trend_rate.py:58-100 restatements against the reference's outputs."""
    D = np.load(os.path.join(golden_dir, "ratemaps.npz"))
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, -1), (1, 2)):
        key = "ddv2_mb%d_md%d" % (mb, md)
        for j, args in enumerate(D[key + "/args"]):
            with np.errstate(all="ignore"):
                lik, b, d, ni, nf = lo.ddv2_likelihood_function(args, D["N_SPEC"], D["N_EXTI"], D["DT"],
                                                                D["TIME_RANGE"], mb, md)
            for got, name in ((b, "birth"), (d, "death"), (ni, "niche"), (nf, "niche_frac")):
                assert np.allclose(got, D[key + "/" + name][j], rtol=1e-13, equal_nan=True), (key, j, name)
            assert np.allclose(lik, D[key + "/lik"][j], rtol=1e-12, equal_nan=True)
    trend = lo.normalise_trend(D["trend_raw"])
    assert np.array_equal(trend, D["TREND"])
    for cb, cd in ((0, 0), (1, 0), (0, 1)):
        key = "trend_cb%d_cd%d" % (cb, cd)
        for j, args in enumerate(D[key + "/args"]):
            with np.errstate(all="ignore"):
                lik, b, d = lo.trend_likelihood_function(args, D["N_SPEC"], D["N_EXTI"], D["DT"], trend, cb, cd)
            assert np.allclose(b, D[key + "/birth"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(d, D[key + "/death"][j], rtol=1e-13, equal_nan=True)
            assert np.allclose(lik, D[key + "/lik"][j], rtol=1e-12, equal_nan=True)
            assert lo.trend_calc_prior(args) == pytest.approx(D[key + "/prior"][j], rel=1e-12)


TRAJ = ["example_TBP_m0_s42", "example_TBP_m2_s7", "example_TBP_m1_s3", "example_TBP_m3_s9",
        "example_TBP_m0_s11_const_rates1", "example_TBP_m0_s12_const_death_rate1",
        "example_TBP_m0_s13_use_rate_HP0_Poisson_prior2.5", "metal_bands_m2_s5"]


@pytest.mark.parametrize("key", TRAJ)
def test_mcmc_loop_reproduces_reference_trajectory(G, golden_dir, key):
    """Fed numpy's legacy stream with the reference's seed, the restated loop must walk the
    reference's own trajectory: every sampled row of the three logs, to print precision."""
    T = np.load(os.path.join(golden_dir, "trajectories.npz"))
    model, seed, n, s = [int(v) for v in T[key + "/meta"]]
    name = "metal_bands" if key.startswith("metal") else "example_TBP"
    stats = dict(sp=G[name + "/sp"], ex=G[name + "/ex"], br=G[name + "/br"],
                 ex_dead=G[name + "/ex_dead"], br_dead=G[name + "/br_dead"])
    start, end = G[name + "/start_end"]
    st = mo.Settings(model_BDI=model)
    if "const_rates1" in key:
        st.const_rates = 1
    if "const_death_rate1" in key:
        st.const_death_rate = 1
    if "use_rate_HP0" in key:
        st.use_rate_HP, st.Poisson_HP = 0, 2.5
    np.random.seed(seed)
    with np.errstate(all="ignore"):
        out = mo.run_mcmc(stats, start, end, st, mo.NumpyLegacyDraws(), n, s,
                          emp=(G[name + "/B_EMP"], G[name + "/D_EMP"]))
    mc = np.array(out["mcmc"])
    ref = T[key + "/mcmc"]
    assert mc.shape == ref.shape
    # integer columns (it, K_l, K_m) exactly; floats to the log's print precision
    assert np.array_equal(mc[:, [0, 6, 7]], ref[:, [0, 6, 7]])
    assert np.allclose(mc[:, :13], ref[:, :13], rtol=1e-9, atol=1e-9)
    assert np.allclose(mc[:, 13:], ref[:, 13:], rtol=1e-6, atol=1e-8, equal_nan=True)
    for kind in ("sp", "ex"):
        R = T["%s/%s" % (key, kind)]
        for i, row in enumerate(out[kind]):
            assert np.allclose(row, R[i, :len(row)], rtol=1e-10)
            assert np.all(np.isnan(R[i, len(row):]))


DD_TRAJ = ["mb2_md2_s4", "mb1_md1_s5", "mb0_md0_s6", "mb2_md-1_s7", "mb2_md0_s8"]


@pytest.mark.parametrize("key", DD_TRAJ)
def test_dd_mcmc_loop_reproduces_reference_trajectory(golden_dir, key):
    """DDRate.py's sampler (DD:124-241) restated in oracle/dd_mcmc_oracle.py: fed numpy's legacy stream with the
    reference's seed it must write the reference's own log rows (full reference runs on metal_bands)."""
    from oracle import dd_mcmc_oracle as ddo
    T = np.load(os.path.join(golden_dir, "dd_trajectories.npz"))
    D = np.load(os.path.join(golden_dir, "ddrate.npz"))
    mb, md, seed, n, s = [int(v) for v in T[key + "/meta"]]
    origin, present, _ = D["meta"]
    with np.errstate(all="ignore"):
        emp = (D["N_SPEC"] / D["DT"], D["N_EXTI"] / D["DT"])
    np.random.seed(seed)
    rows = ddo.run_dd_mcmc(D["N_SPEC"], D["N_EXTI"], D["DT"], D["TIME_RANGE"], origin, present, mb, md,
                           ddo.NumpyLegacyDraws(), n, s, emp=emp)
    rows = np.array(rows)
    assert rows.shape[0] == T[key + "/head"].shape[0]
    assert np.allclose(rows[:, :14], T[key + "/head"], rtol=1e-9, atol=1e-9)
    assert np.allclose(rows[:, -3:], T[key + "/adequacy"], rtol=1e-7, atol=1e-9, equal_nan=True)
    assert np.allclose(rows[:25], T[key + "/full25"], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert len(set(np.round(rows[:, 2], 6))) > 20          # the chain moved


def test_cfg1_fixed_two_shifts_reproduces_reference_runMCMC(G, golden_dir):
    """BASELINE.json configs[0] (example_dataTBP, 1 chain, fixed 2 rate shifts, CPU): the reference's own runMCMC
    called with a 3-rate initial state and -const_rates 1 (tests/golden/make_golden.py::make_cfg1); the restated loop
    started from the same state on the same MT19937 stream writes the same three logs.  K stays 3 and, update_times
    being a no-op (A9), so do the shift times."""
    T = np.load(os.path.join(golden_dir, "cfg1_fixed_shifts.npz"))
    seed, n, s = [int(v) for v in T["meta"]]
    name = "example_TBP"
    stats = dict(sp=G[name + "/sp"], ex=G[name + "/ex"], br=G[name + "/br"])
    start, end = G[name + "/start_end"]
    np.random.seed(seed)
    with np.errstate(all="ignore"):
        out = mo.run_mcmc(stats, start, end, mo.Settings(model_BDI=0, const_rates=1), mo.NumpyLegacyDraws(), n, s,
                          init=(T["L0"], T["M0"], T["times0"], T["times0"]), emp=(G[name + "/B_EMP"], G[name + "/D_EMP"]))
    mc, ref = np.array(out["mcmc"]), T["mcmc"]
    assert mc.shape == ref.shape == (n // s, 16)
    assert np.all(ref[:, 6] == 3) and np.all(ref[:, 7] == 3)
    assert np.array_equal(mc[:, [0, 6, 7]], ref[:, [0, 6, 7]])
    assert np.allclose(mc[:, :13], ref[:, :13], rtol=1e-9, atol=1e-9)
    assert np.allclose(mc[:, 13:], ref[:, 13:], rtol=1e-6, atol=1e-8, equal_nan=True)
    for kind in ("sp", "ex"):
        for i, row in enumerate(out[kind]):
            assert np.allclose(row, T[kind][i, :len(row)], rtol=1e-10)
            assert np.array_equal(row[3:], T["times0"][1:3])       # the two shift times never move
    assert len(set(np.round(mc[:, 2], 6))) > 100


def test_simulator_oracle_properties():
    """oracle/sim_oracle.py (the reference simulators' per-step Bernoulli scheme): vectorised Philox equals the scalar
    stream; conservation (living count = births - deaths so far); zero rates leave the start population extant;
    a death rate of 1 kills everyone in the first step; the DD generators cap growth near the carrying capacity."""
    from oracle import philox as px
    from oracle import sim_oracle as so
    s = px.Stream(99, 1234)
    u = px.uniform_a_np(np.arange(6) + (1 << 33), px.P_SIM, 0, 99, 1234)
    assert np.array_equal(u, [s.pair(int(i) + (1 << 33), px.P_SIM, 0)[0] for i in range(6)])
    T = 60
    ts, te, trace = so.simulate_bd(300, T, 4, np.full(T, .06), np.full(T, .05))
    assert trace[0] == 300
    for t in range(1, T):      # living at the start of step t = born before t - dead before t
        assert trace[t] == np.sum(ts < t) - np.sum(te < t)
    assert np.sum(te == T) == trace[-1] + np.sum(ts == T - 1) - np.sum(te == T - 1)
    assert np.all(te >= ts) and np.all(np.diff(ts) >= 0)
    ts0, te0, tr0 = so.simulate_bd(50, 10, 1, np.zeros(10), np.zeros(10))
    assert len(ts0) == 50 and np.all(te0 == 10) and np.all(tr0 == 50)
    ts1, te1, tr1 = so.simulate_bd(50, 5, 1, np.zeros(5), np.ones(5))
    assert len(ts1) == 50 and np.all(te1 == 0) and tr1[1] == 0
    for mode in (1, 2):
        tsd, ted, trd = so.simulate_bd(200, 400, 3, mode=mode, l0=.4, m0=.1, K=2000.0, scale=4.0)
        eq = 2000.0 * (.4 - .1) / (.4 + .1) if mode == 1 else 2000.0 / 2.0      # where lambda(D) = mu(D)
        assert abs(trd[-50:].mean() - eq) < 0.12 * eq


@pytest.mark.parametrize("key", ["cb0_cd0_s4", "cb1_cd0_s5", "cb0_cd1_s6"])
def test_trend_mcmc_loop_reproduces_reference_trajectory(golden_dir, key):
    """trend_rate.py's sampler (:102-196) restated in oracle/trend_mcmc_oracle.py: on numpy's legacy stream with the
    reference's seed it writes the reference's own log rows (full reference runs, metal_bands + a synthetic trend)."""
    from oracle import trend_mcmc_oracle as tro
    T = np.load(os.path.join(golden_dir, "trend_trajectories.npz"))
    R = np.load(os.path.join(golden_dir, "ratemaps.npz"))
    cb, cd, seed, n, s = [int(v) for v in T[key + "/meta"]]
    with np.errstate(all="ignore"):
        emp = (R["N_SPEC"] / R["DT"], R["N_EXTI"] / R["DT"])
    np.random.seed(seed)
    rows = np.array(tro.run_trend_mcmc(R["N_SPEC"], R["N_EXTI"], R["DT"], R["TREND"], tro.NumpyLegacyDraws(), n, s,
                                       bool(cb), bool(cd), emp=emp))
    assert rows.shape[0] == T[key + "/head"].shape[0]
    assert np.allclose(rows[:, :12], T[key + "/head"], rtol=1e-9, atol=1e-9)
    assert np.allclose(rows[:, -3:], T[key + "/adequacy"], rtol=1e-7, atol=1e-9, equal_nan=True)
    assert np.allclose(rows[:25], T[key + "/full25"], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert len(set(np.round(rows[:, 2], 6))) > 20
