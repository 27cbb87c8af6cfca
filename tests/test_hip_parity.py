"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference-generated
golden vectors.  Bit-exact for counts / indices, <= 1e-9 relative for fp64 log-likelihoods
(the tolerance BASELINE.json's north_star states)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-9
DATASETS = ["example_TBP", "example_TAD", "metal_bands", "simulated"]


@pytest.fixture(scope="module")
def ops():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X: no ROCm device visible (the HIP path has no fallback)")
    from literate_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "binning_lik.npz"))


@pytest.fixture(scope="module")
def P(golden_dir):
    with open(os.path.join(golden_dir, "proposals_priors.json")) as f:
        return json.load(f)


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("name", DATASETS)
def test_bin_events_golden(ops, G, name):
    ts, te = G[name + "/ts"], G[name + "/te"]
    t0, n_bins = int(G[name + "/start_end"][0]), len(G[name + "/sp"])
    lo = np.arange(t0, t0 + n_bins, dtype=float)
    sp, ex, br = ops.bin_events(ts, te, lo, lo + 1)
    assert np.array_equal(_np(sp), G[name + "/sp"])
    assert np.array_equal(_np(ex), G[name + "/ex"])
    assert np.array_equal(_np(br), G[name + "/br"])          # half-integer data: exact in any order
    # dead-only statistics of model 3 (LRF:529-546)
    dead = te < G[name + "/start_end"][1]
    sp, ex, br = ops.bin_events(ts[dead], te[dead], lo, lo + 1)
    assert np.array_equal(_np(ex), G[name + "/ex_dead"])
    assert np.array_equal(_np(br), G[name + "/br_dead"])
    # arbitrary (non-integer) windows through the library path
    w = G[name + "/lib_windows"]
    sp, ex, br = ops.bin_events(G[name + "/lib_ts"], G[name + "/lib_te"], w[:, 0], w[:, 1])
    assert np.array_equal(_np(sp), w[:, 2].astype(np.int64))
    assert np.array_equal(_np(ex), w[:, 3].astype(np.int64))
    assert np.allclose(_np(br), w[:, 4], rtol=1e-12, atol=0)


def test_bin_events_edge_cases(ops):
    from oracle import literate_oracle as lo
    rng = np.random.default_rng(0)
    for n in (1, 2, 63, 64, 65, 1000, 4097):
        ts = rng.uniform(0, 30, n)
        te = ts + rng.exponential(4, n)
        te[rng.random(n) < 0.1] = 40.5                     # extant
        wl = np.array([0.0, 3.3, 29.0, 100.0, -5.0, 7.0, 7.0, 12.5, 2.0])
        wh = np.array([1.0, 9.9, 31.0, 101.0, -1.0, 7.0, 6.0, 13.5, 40.0])   # includes empty / inverted windows
        sp, ex, br = ops.bin_events(ts, te, wl, wh)
        for k in range(len(wl)):
            a, b, c = lo.precompute_events(ts, te, wl[k], wh[k])
            assert (int(sp[k]), int(ex[k])) == (a, b)
            assert float(br[k]) == pytest.approx(c, rel=1e-12, abs=1e-12)


def _states(G, name):
    KL, KM = G[name + "/state_KL"], G[name + "/state_KM"]
    return (G[name + "/state_L"], G[name + "/state_tL"], KL, G[name + "/state_M"], G[name + "/state_tM"], KM)


@pytest.mark.parametrize("name", DATASETS)
def test_expand_rates_golden(ops, G, name):
    L, tL, KL, M, tM, KM = _states(G, name)
    n_bins = len(G[name + "/sp"])
    lam = _np(ops.expand_rates(L, tL, KL, n_bins, mode=0))
    mu = _np(ops.expand_rates(M, tM, KM, n_bins, mode=0))
    for i in range(len(KL)):
        assert np.array_equal(lam[i], L[i][G[name + "/state_indL"][i]])
        assert np.array_equal(mu[i], M[i][G[name + "/state_indM"][i]])
    # round mode = get_rate_index on raw times (LRF:129)
    from oracle import literate_oracle as lo
    lam_r = _np(ops.expand_rates(L, tL, KL, n_bins, mode=1))
    for i in range(len(KL)):
        ind = lo.get_rate_index(tL[i, :KL[i] + 1], n_bins)
        if len(ind) == n_bins:
            assert np.array_equal(lam_r[i], L[i][ind])


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_loglik_batch_matches_reference(ops, G, name, model):
    """Summed per-lineage log-likelihood == the reference's calc_likelihood on identical states."""
    ts, te = G[name + "/ts"], G[name + "/te"]
    start, end = G[name + "/start_end"]
    L, tL, KL, M, tM, KM = _states(G, name)
    n_bins = len(G[name + "/sp"])
    lam = ops.expand_rates(L, tL, KL, n_bins)
    mu = ops.expand_rates(M, tM, KM, n_bins)
    out = _np(ops.bd_loglik_batch(ts, te, float(int(start)), lam, mu, model, G[name + "/br"], end))
    ref = G["%s/lik_model%d" % (name, model)]
    ok = np.isfinite(ref)
    assert ok.sum() >= len(ref) - 2
    assert np.allclose(out[ok], ref[ok], rtol=REL, atol=1e-9), np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok]))


def test_loglik_random_float_data_vs_oracle(ops):
    """Non-integer times, ragged sizes, odd chain counts, unaligned pointers, wide windows."""
    import torch
    from oracle import literate_oracle as lo
    rng = np.random.default_rng(1)
    for n, C, n_bins in [(1, 1, 3), (2, 3, 5), (511, 5, 24), (513, 8, 24), (4099, 9, 130), (20001, 17, 257),
                         (3000, 2, 1025), (1000, 1, 4000)]:
        t0 = 3.0
        ts = rng.uniform(t0 - 2, t0 + n_bins + 1, n)          # some lineages start before / after the window
        te = ts + rng.exponential(n_bins / 6.0, n)
        te[rng.random(n) < 0.2] = t0 + n_bins + 0.5            # extant
        k = rng.integers(0, n, max(1, n // 20))
        ts[k] = np.floor(ts[k])                                 # exact bin edges
        te[k] = np.ceil(te[k])
        lam = np.exp(rng.uniform(np.log(.02), np.log(1.5), (C, n_bins)))
        mu = np.exp(rng.uniform(np.log(.02), np.log(1.5), (C, n_bins)))
        br = _np(ops.bin_events(ts, te, t0 + np.arange(n_bins), t0 + np.arange(n_bins) + 1.0)[2])
        for model in (0, 1, 2):
            got = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, model, br))
            for c in range(C):
                ref = lo.per_lineage_loglik(ts, te, t0, lam[c], mu[c], model, br)
                assert got[c] == pytest.approx(ref, rel=REL, abs=1e-9), (n, C, n_bins, model, c)
            if n <= 4099:
                dense = lo.per_lineage_loglik_dense(ts, te, t0, lam[0], mu[0], model, br)
                assert got[0] == pytest.approx(dense, rel=REL, abs=1e-9)
        # unaligned device pointers take the scalar-load path: identical bits
        if n > 8:
            tsd = torch.as_tensor(np.concatenate([[0.0], ts])).cuda()[1:]
            ted = torch.as_tensor(np.concatenate([[0.0], te])).cuda()[1:]
            a = _np(ops.bd_loglik_batch(tsd, ted, t0, lam, mu, 2))
            b = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2))
            assert np.array_equal(a, b)


def test_loglik_bad_arguments(ops):
    ts, te = np.array([0.0, 1.0]), np.array([2.5, 3.5])
    lam = np.full((1, 3), .1)
    with pytest.raises(ValueError, match="LR_ERR_T0"):
        ops.bd_loglik_batch(ts, te, 0.5, lam, lam, 2)
    with pytest.raises(ValueError, match="LR_ERR_MODEL"):
        ops.bd_loglik_batch(ts, te, 0.0, lam, lam, 0, None)
    with pytest.raises(ValueError, match="LR_ERR_MODEL"):
        ops.bd_loglik_batch(ts, te, 0.0, lam, lam, 7, None)
    with pytest.raises(ValueError, match="LR_ERR_SIZE"):
        ops.bd_loglik_batch(ts, te, 0.0, np.full((1, 5000), .1), np.full((1, 5000), .1), 2)


def test_loglik_properties_full_size(ops):
    """cfg4-sized input (1e5 lineages): size-independent properties instead of an O(N*bins) oracle."""
    import torch
    from literate_amd import synth
    ts, te, _ = synth.make_lineages(100_000, n_bins=128, seed=0)
    n_bins, t0 = 128, 0.0
    rng = np.random.default_rng(2)
    C = 64
    lam = np.exp(rng.uniform(np.log(.05), np.log(.6), (C, n_bins)))
    mu = np.exp(rng.uniform(np.log(.02), np.log(.3), (C, n_bins)))
    full = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2))
    # (1) additivity over a split of the lineages
    cut = 33_333
    a = _np(ops.bd_loglik_batch(ts[:cut], te[:cut], t0, lam, mu, 2))
    b = _np(ops.bd_loglik_batch(ts[cut:], te[cut:], t0, lam, mu, 2))
    assert np.allclose(a + b, full, rtol=1e-12)
    # (2) permutation invariance
    p = rng.permutation(len(ts))
    assert np.allclose(_np(ops.bd_loglik_batch(ts[p], te[p], t0, lam, mu, 2)), full, rtol=1e-12)
    # (3) a chain's value does not depend on what else is in the batch
    solo = _np(ops.bd_loglik_batch(ts, te, t0, lam[5:6], mu[5:6], 2))
    assert solo[0] == pytest.approx(full[5], rel=1e-12)
    # (4) equals the binned Keiding form on statistics binned by the HIP kernel (BDIx:365-368 check)
    lo_ = np.arange(n_bins, dtype=float)
    sp, ex, br = [_np(x) for x in ops.bin_events(ts, te, lo_, lo_ + 1)]
    keiding = (np.log(lam) * sp - lam * br).sum(1) + (np.log(mu) * ex - mu * br).sum(1)
    assert np.allclose(full, keiding, rtol=REL)
    # (5) bitwise reproducible
    again = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2))
    assert np.array_equal(full, again)
    # (6) scaling exposure: constant rates -> closed form
    one = _np(ops.bd_loglik_batch(ts, te, t0, np.full((1, n_bins), .3), np.full((1, n_bins), .2), 2))
    assert one[0] == pytest.approx(np.log(.3) * sp.sum() + np.log(.2) * ex.sum() - .5 * br.sum(), rel=1e-12)
    assert torch.cuda.is_available()


def test_proposal_scorers_golden(ops, P):
    pr = P["proposals"]
    kmax = 16

    def pack(recs, move, idx_key, draw_fn):
        C = len(recs)
        rates = np.zeros((C, kmax)); times = np.zeros((C, kmax + 1)); K = np.zeros(C, np.int32)
        index = np.zeros(C, np.int32); draws = np.zeros((C, 2 * kmax))
        for i, r in enumerate(recs):
            k = len(r.get("rates", r.get("q")))
            K[i] = k
            rates[i, :k] = r.get("rates", r.get("q"))
            if "times" in r:
                times[i, :k + 1] = r["times"]
            index[i] = r.get(idx_key, 0)
            draw_fn(r, draws[i], k)
        return rates, times, K, np.full(C, move, np.int32), index, draws

    def d_add(r, d, k):
        d[0], d[1] = r["delta"], r["u"]

    def d_mult(r, d, k):
        d[:k], d[kmax:kmax + k] = r["ff"], r["u"]

    o_r, o_t, o_k, o_s = [_np(x) for x in ops.rj_propose_score(*pack(pr["add"], 1, "ind", d_add))]
    for i, r in enumerate(pr["add"]):
        k = len(r["out_rates"])
        assert o_k[i] == k
        assert np.allclose(o_r[i, :k], r["out_rates"], rtol=1e-13)
        assert np.allclose(o_t[i, :k + 1], r["out_times"], rtol=1e-15)
        assert o_s[i] == pytest.approx(r["score"], rel=1e-11, abs=1e-11)
    o_r, o_t, o_k, o_s = [_np(x) for x in ops.rj_propose_score(*pack(pr["remove"], 2, "idx", lambda r, d, k: None))]
    for i, r in enumerate(pr["remove"]):
        k = len(r["out_rates"])
        assert o_k[i] == k
        assert np.allclose(o_r[i, :k], r["out_rates"], rtol=1e-13)
        assert np.array_equal(o_t[i, :k + 1], r["out_times"])
        assert o_s[i] == pytest.approx(r["score"], rel=1e-11, abs=1e-11)
    o_r, o_t, o_k, o_s = [_np(x) for x in ops.rj_propose_score(*pack(pr["mult"], 0, "none", d_mult))]
    for i, r in enumerate(pr["mult"]):
        k = len(r["out"])
        assert np.allclose(o_r[i, :k], r["out"], rtol=1e-14)
        assert o_s[i] == pytest.approx(r["hastings"], rel=1e-12, abs=1e-14)


def test_priors_golden(ops, P):
    pri = P["priors"]
    kmax = 16
    recs = pri["gamma"]
    rates = np.ones((len(recs), kmax)); K = np.zeros(len(recs), np.int32); b = np.zeros(len(recs))
    for i, r in enumerate(recs):
        K[i] = len(r["x"]); rates[i, :K[i]] = r["x"]; b[i] = r["b"]
    out = _np(ops.log_priors(rates, K, 2.0, b))
    assert np.allclose(out, [r["out"] for r in recs], rtol=1e-12, atol=1e-12)
    recs = pri["poisson"]
    K = np.array([r["k"] for r in recs], np.int32)
    rate = np.array([r["rate"] for r in recs], float)
    rates = np.ones((len(recs), 40))
    g = np.ones(len(recs))
    base = _np(ops.log_priors(rates, K, 2.0, g))
    out = _np(ops.log_priors(rates, K, 2.0, g, rate)) - base
    assert np.allclose(out, [r["out"] for r in recs], rtol=1e-11, atol=1e-11)


def test_dd_rates_and_likelihood_golden(ops, G, golden_dir):
    D = np.load(os.path.join(golden_dir, "ddrate.npz"))
    ts, te = G["metal_bands/lib_ts"], G["metal_bands/lib_te"]
    origin = D["meta"][0]
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, 0), (1, 2)):
        key = "mb%d_md%d" % (mb, md)
        b, d, ni, nf = ops.dd_rates(D[key + "/args"], D["DT"], mb, md)
        assert np.allclose(_np(b), D[key + "/birth"], rtol=1e-12, equal_nan=True)
        assert np.allclose(_np(d), D[key + "/death"], rtol=1e-12, equal_nan=True)
        assert np.allclose(_np(ni), D[key + "/niche"], rtol=1e-12, equal_nan=True)
        assert np.allclose(_np(nf), D[key + "/niche_frac"], rtol=1e-12, equal_nan=True)
        # likelihood half (DD:86,101) = per-lineage Keiding scan on those rates
        ref = D[key + "/lik"].sum(1)
        ok = np.isfinite(ref)
        lik = _np(ops.bd_loglik_batch(ts, te, origin, b, d, 2))
        assert np.allclose(lik[ok], ref[ok], rtol=REL)


def test_ddv2_and_trend_rate_maps_golden(ops, G, golden_dir):
    """lr_ddv2_rates / lr_trend_rates against the reference's own outputs (DDRatev2.py, trend_rate.py
    run in the build container), and their likelihood halves through the per-lineage Keiding scan."""
    D = np.load(os.path.join(golden_dir, "ratemaps.npz"))
    ts, te = G["metal_bands/lib_ts"], G["metal_bands/lib_te"]
    origin = np.load(os.path.join(golden_dir, "ddrate.npz"))["meta"][0]
    for mb, md in ((2, 2), (1, 1), (0, 0), (2, -1), (1, 2)):
        key = "ddv2_mb%d_md%d" % (mb, md)
        b, d, ni, nf = ops.ddv2_rates(D[key + "/args"], D["DT"], mb, md)
        for got, name in ((b, "birth"), (d, "death"), (ni, "niche"), (nf, "niche_frac")):
            assert np.allclose(_np(got), D[key + "/" + name], rtol=1e-12, equal_nan=True), (key, name)
        ref = D[key + "/lik"].sum(1)
        ok = np.isfinite(ref)
        lik = _np(ops.bd_loglik_batch(ts, te, origin, b, d, 2))
        assert ok.sum() >= 5 and np.allclose(lik[ok], ref[ok], rtol=REL)
    for cb, cd in ((0, 0), (1, 0), (0, 1)):
        key = "trend_cb%d_cd%d" % (cb, cd)
        b, d = ops.trend_rates(D[key + "/args"], D["TREND"], cb, cd)
        assert np.allclose(_np(b), D[key + "/birth"], rtol=1e-12, equal_nan=True)
        assert np.allclose(_np(d), D[key + "/death"], rtol=1e-12, equal_nan=True)
        ref = D[key + "/lik"].sum(1)
        ok = np.isfinite(ref)
        lik = _np(ops.bd_loglik_batch(ts, te, origin, b, d, 2))
        assert ok.sum() >= 5 and np.allclose(lik[ok], ref[ok], rtol=REL)
    with pytest.raises(ValueError):
        ops.trend_rates(np.zeros((2, 5)), D["TREND"])


def test_device_rng_matches_oracle_stream(ops):
    from oracle import philox as px
    rng = np.random.default_rng(3)
    n = 400
    it = rng.integers(0, 2**40, n)
    purpose = rng.integers(0, 11, n).astype(np.int32)
    idx = rng.integers(0, 200, n).astype(np.int32)
    kind = rng.integers(0, 4, n).astype(np.int32)
    shape = rng.uniform(1.0, 30.0, n)
    seed, chain = 123456789, 77
    got = _np(ops.debug_draws(seed, chain, it, purpose, idx, kind, shape))
    s = px.Stream(seed, chain)
    # Philox known-answer (Random123 kat_vectors: zero counter, zero key)
    assert px.philox4x32_10(0, 0, 0, 0, 0, 0) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    for i in range(n):
        if kind[i] == 0:
            ref = s.pair(int(it[i]), int(purpose[i]), int(idx[i]))[0]
            assert got[i] == ref
        elif kind[i] == 1:
            ref = s.pair(int(it[i]), int(purpose[i]), int(idx[i]))[1]
            assert got[i] == ref
        elif kind[i] == 2:
            ref = s.normal(int(it[i]), int(purpose[i]), int(idx[i]))
            assert got[i] == pytest.approx(ref, rel=1e-13, abs=1e-15)
        else:
            ref = s.gamma(int(it[i]), int(purpose[i]), int(idx[i]), float(shape[i]))
            assert got[i] == pytest.approx(ref, rel=1e-12)


@pytest.mark.parametrize("case", ["constant", "key_innovation", "mass_extinction", "dd_nb4", "dd_abc", "overflow"])
def test_device_simulator_matches_oracle(ops, case):
    """lr_simulate_bd against oracle/sim_oracle.py: the multiset of (birth step, death step) pairs and the
    living count per step, bit for bit (integers), for given-rate and diversity-dependent runs."""
    from literate_amd import synth
    from oracle import sim_oracle as so
    scale, T, seed = 4, 40, 77
    n_steps = T * scale
    t = np.arange(n_steps) // scale
    kw = dict(mode=0)
    if case == "constant":
        la, mu = synth.rates_constant(T, .12, .08)
    elif case == "key_innovation":
        la, mu = synth.rates_key_innovation(T, .05, .05, .6, .3)
    elif case in ("mass_extinction", "overflow"):
        la, mu = synth.rates_mass_extinction(T, .15, .1, .4, 1.0, .08)
    if case.startswith("dd"):
        kw = dict(mode=1 if case == "dd_nb4" else 2, l0=.4, m0=.1, K=5000.0, scale=float(scale))
        lam_s = mu_s = None
    else:
        lam_s, mu_s = la[t] / scale, mu[t] / scale
    n_start = 800
    if case == "overflow":
        with pytest.raises(OverflowError):
            ops.simulate_bd(n_start, n_steps, seed, lam_s, mu_s, capacity=1000)
        return
    ts, te, trace = ops.simulate_bd(n_start, n_steps, seed, lam_s, mu_s, **kw)
    rts, rte, rtrace = so.simulate_bd(n_start, n_steps, seed, lam_s, mu_s, **kw)
    ts, te = _np(ts), _np(te)
    order = np.lexsort((te, ts))
    assert len(ts) == len(rts) > n_start
    assert np.array_equal(ts[order], rts) and np.array_equal(te[order], rte)
    assert np.array_equal(_np(trace), rtrace)
    # LiteRate input made from it: integer time units, jitter on te, sorted, extant at T + jitter
    if case == "constant":
        lts, lte, _ = synth.simulate(n_start, T, scale, seed, rates=(la, mu))
        assert len(lts) == len(ts) and np.all(np.diff(lts) >= 0) and lte.max() == T + 0.5 and np.all(lte - 0.5 >= lts)


def test_new_entry_points_reject_bad_arguments(ops):
    """Error behaviour of the later ABI additions: status codes, no launch (0 ok, < 0 invalid argument)."""
    import ctypes as C
    import torch
    from literate_amd import _hip
    from literate_amd.engine import ChainEngine
    lib = _hip.load()
    dt = torch.ones(5, dtype=torch.float64, device="cuda")
    out = [torch.empty((2, 5), dtype=torch.float64, device="cuda") for _ in range(4)]
    a9 = torch.ones((2, 9), dtype=torch.float64, device="cuda")
    P = _hip.ptr
    assert lib.lr_ddv2_rates(P(a9), P(dt), 5, 2, 3, 2, P(out[0]), P(out[1]), P(out[2]), P(out[3]), None) == _hip.LR_ERR_MODEL
    assert lib.lr_ddv2_rates(None, P(dt), 5, 2, 2, 2, P(out[0]), P(out[1]), P(out[2]), P(out[3]), None) == _hip.LR_ERR_NULL
    assert lib.lr_trend_rates(P(a9), P(dt), 0, 2, 0, 0, P(out[0]), P(out[1]), None) == _hip.LR_ERR_SIZE
    i64 = torch.ones(5, dtype=torch.int64, device="cuda")
    assert lib.lr_binned_keiding(P(out[0]), P(out[1]), P(i64), None, P(dt), 5, 2, P(out[2]), P(out[3]), None) == _hip.LR_ERR_NULL
    ws = torch.zeros(64, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    ts = torch.empty(100, dtype=torch.float64, device="cuda")
    assert lib.lr_simulate_bd(None, None, 4, 0, 0., 0., 1., 1., 10, 100, 1, P(ts), P(ts), P(cnt), None, P(ws), 64, None) == _hip.LR_ERR_NULL
    assert lib.lr_simulate_bd(P(dt), P(dt), 4, 5, 0., 0., 1., 1., 10, 100, 1, P(ts), P(ts), P(cnt), None, P(ws), 64, None) == _hip.LR_ERR_MODEL
    assert lib.lr_simulate_bd(P(dt), P(dt), 4, 0, 0., 0., 1., 1., 200, 100, 1, P(ts), P(ts), P(cnt), None, P(ws), 64, None) == _hip.LR_ERR_SIZE
    assert lib.lr_simulate_bd(P(dt), P(dt), 4, 0, 0., 0., 1., 1., 10, 100, 1, P(ts), P(ts), P(cnt), None, P(ws), 8, None) == _hip.LR_ERR_WORKSPACE
    assert lib.lr_mcmc_restore(None, None) == _hip.LR_ERR_NULL
    # a parametric sampler needs the Keiding model, at most 256 bins and its per-bin array
    with pytest.raises(ValueError, match="LR_ERR_MODEL"):
        ChainEngine(np.array([0., 1.]), np.array([2.5, 3.5]), 2, model=0, stats=(0.0, 3, np.ones(3)),
                    dd=dict(m_birth=2, m_death=2, present=3.5))
    with pytest.raises(ValueError, match="LR_ERR_MODEL"):
        ChainEngine(np.array([0., 1.]), np.array([2.5, 3.5]), 2, model=2, stats=(0.0, 3, np.ones(3)),
                    dd=dict(m_birth=5, m_death=2, present=3.5))
