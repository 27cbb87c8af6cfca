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


def _general_bin_events(ops, ts, te, lo, hi):
    """lr_bin_events proper (arbitrary windows): device tensors are never routed to the unit-window kernel"""
    import torch
    return ops.bin_events(ts, te, torch.as_tensor(np.asarray(lo, float)).cuda(), torch.as_tensor(np.asarray(hi, float)).cuda())


def _fsum_br(ts, te, lo, hi):
    """the reference's per-lineage overlap terms of one window (get_br, lib:74-79), added EXACTLY (math.fsum)"""
    import math
    with np.errstate(invalid="ignore"):
        d = np.minimum(te, hi) - np.maximum(ts, lo)
        d = d[d > 0]
    return math.fsum(d.tolist())


def test_bin_unit_events_against_oracle_and_general_kernel(ops):
    """lr_bin_unit_events (one pass, integer accumulation) on float times: counts exact against the oracle's
    precompute_events, br_length = the EXACT sum of the reference's per-lineage terms rounded once (bit for bit against
    math.fsum when every time is >= 1, where each term is a multiple of 2^-52), and against lr_bin_events on the same
    windows.  Ragged sizes, lineages outside the window, exact bin edges, te <= ts, NaN / inf, unaligned pointers, bin
    counts on both sides of the replication steps of the LDS histograms."""
    import torch
    from oracle import literate_oracle as lo
    rng = np.random.default_rng(11)
    for n, W, t0 in [(1, 1, 5.0), (2, 3, 2.0), (63, 24, 1.0), (1000, 24, 10.0), (4097, 128, 3.0), (20001, 300, 7.0),
                     (30000, 1300, 100.0), (5000, 4000, 1.0), (70000, 33, -20.0), (9000, 2001, 4.0), (9000, 3003, 2.0),
                     (6000, 4094, 1.0)]:
        ts = rng.uniform(t0 - 3, t0 + W + 2, n)
        te = ts + rng.exponential(max(W / 6.0, 0.7), n)
        te[rng.random(n) < 0.15] = t0 + W + 0.5                    # extant, past the last window
        k = rng.integers(0, n, max(1, n // 10))
        ts[k], te[k] = np.floor(ts[k]), np.ceil(te[k])              # exact bin edges
        k = rng.integers(0, n, max(1, n // 25))
        te[k] = ts[k] - rng.integers(0, 3, len(k)) * rng.random(len(k))     # te <= ts: events, no lineage-time
        if n > 100:
            ts[5], te[7], ts[9], te[11], te[13] = np.nan, np.nan, -np.inf, np.inf, -np.inf
            te[15] = ts[15] = t0 + 4.0                              # born and dead on one edge
        sp, ex, br = [_np(x) for x in ops.bin_unit_events(ts, te, t0, W)]
        lo_e = t0 + np.arange(W, dtype=float)
        gsp, gex, gbr = [_np(x) for x in _general_bin_events(ops, ts, te, lo_e, lo_e + 1.0)]
        assert np.array_equal(sp, gsp) and np.array_equal(ex, gex), (n, W)
        assert np.allclose(br, gbr, rtol=1e-13, atol=1e-13), (n, W)
        exact_terms = t0 >= 1.0
        for w in (range(W) if W <= 130 else rng.integers(0, W, 40)):
            with np.errstate(invalid="ignore"):
                a, b, c = lo.precompute_events(ts, te, lo_e[w], lo_e[w] + 1.0)
            assert (int(sp[w]), int(ex[w])) == (a, b), (n, W, w)
            ref = _fsum_br(ts, te, lo_e[w], lo_e[w] + 1.0)
            if exact_terms:
                assert br[w] == ref, (n, W, w, br[w], ref)
            else:
                assert br[w] == pytest.approx(ref, rel=1e-15, abs=n * 2.0 ** -53)
        # the routing of ops.bin_events: host unit windows take this kernel (identical bits)
        r = [_np(x) for x in ops.bin_events(ts, te, lo_e, lo_e + 1.0)]
        assert np.array_equal(r[0], sp) and np.array_equal(r[1], ex) and np.array_equal(r[2], br, equal_nan=True)
        if n > 8:
            tsd = torch.as_tensor(np.concatenate([[0.0], ts])).cuda()[1:]
            ted = torch.as_tensor(np.concatenate([[0.0], te])).cuda()[1:]
            u = [_np(x) for x in ops.bin_unit_events(tsd, ted, t0, W)]
            assert np.array_equal(u[0], sp) and np.array_equal(u[1], ex) and np.array_equal(u[2], br)
    with pytest.raises(ValueError, match="LR_ERR_T0"):
        ops.bin_unit_events(np.array([1.0, 2.0]), np.array([2.5, 3.5]), 0.5, 4)
    with pytest.raises(ValueError, match="LR_ERR_SIZE"):
        ops.bin_unit_events(np.array([1.0, 2.0]), np.array([2.5, 3.5]), 0.0, 5000)


@pytest.mark.parametrize("blocks_per_cu", [1, 2, 4])
def test_bin_unit_publication_under_out_of_step_blocks(blocks_per_cu):
    """lr_bin_unit_kernel publishes its column sums without a release / acquire pair (csrc/lr_stats.hip: returning
    agent-scope atomics, s_waitcnt vmcnt(0), barrier, ticket).  Stress of that hand-off: LR_UB_BLOCKS_PER_CU = 1 / 2 / 4
    (256 / 512 / 1024 blocks on all eight XCDs, several per CU with 24 windows, four rounds of blocks with 128), the last
    block 1/64 the size of the others so that tickets are taken far out of step, 200 launches per case - a lost or
    late add shows as a run that differs from the first, which itself equals lr_bin_events and torch.bincount."""
    import json
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "bin_unit_stress.py")
    env = {k: v for k, v in os.environ.items() if k != "LR_UB_BLOCKS_PER_CU"}
    r = subprocess.run([sys.executable, child, str(blocks_per_cu), "200"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("BIN_UNIT_STRESS ")]
    assert len(line) == 1
    out = json.loads(line[0][len("BIN_UNIT_STRESS "):])
    assert out["blocks"] == min(1024, 256 * blocks_per_cu) and len(out["cases"]) == 2
    for c in out["cases"]:
        assert c["first_run_matches_lr_bin_events"] and c["reps"] == 200 and c["differing_runs"] == 0, c


def _big_lineages(n, general, seed):
    """n synthetic lineages on the device: cfg4's generator tiled (bench.py's abi workload), unsorted"""
    import torch
    from literate_amd import synth
    ts0, te0, _ = synth.make_lineages(100_000, n_bins=128, n_shifts=20, seed=0)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    reps = -(-n // len(ts0))
    ts = torch.as_tensor(ts0, device="cuda").repeat(reps)[:n].contiguous()
    te = torch.as_tensor(te0, device="cuda").repeat(reps)[:n].contiguous()
    if general:
        ts = ts + torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 0.999
        te = torch.maximum(torch.ceil(te) - 1.0 + 1e-3 + 0.998 * torch.rand(n, generator=g, device="cuda", dtype=torch.float64),
                           ts + 1e-3)
    return ts, te


@pytest.mark.parametrize("general", [False, True])
def test_streaming_kernels_at_1e7_lineages(ops, general):
    """The two entry points that stream ts / te from HBM, at the size bench.py's `abi` section times them (1e7 lineages,
    160 MB per pass), through size-independent properties: additivity over a split, permutation invariance (BIT-exact
    for the binning: integer accumulation), agreement of the two binning kernels, conservation of lineage-time,
    likelihood == binned Keiding on the binned statistics, closed form at constant rates, bitwise reproducibility."""
    import torch
    n, W, t0 = 10_000_000, 128, 0.0
    ts, te = _big_lineages(n, general, 99)
    sp, ex, br = ops.bin_unit_events(ts, te, t0, W)
    lo_e = torch.arange(W, dtype=torch.float64, device="cuda")
    gsp, gex, gbr = ops.bin_events(ts, te, lo_e, lo_e + 1.0)
    assert torch.equal(sp, gsp) and torch.equal(ex, gex)
    assert torch.allclose(br, gbr, rtol=1e-12, atol=0)
    # counts against torch's own histogram of the bins
    assert torch.equal(sp, torch.bincount(torch.floor(ts).long(), minlength=W)[:W])
    be = (torch.ceil(te) - 1).long()
    assert torch.equal(ex, torch.bincount(be[be < W], minlength=W)[:W])
    # lineage-time is conserved: sum over the windows = sum of the clipped durations
    total = (torch.clamp(te, max=float(W)) - torch.clamp(ts, min=0.0)).clamp(min=0.0).sum()
    assert float(br.sum()) == pytest.approx(float(total), rel=1e-12)
    # ... and window by window against torch's own arithmetic (no kernel of this library: get_br's definition, lib:74-85,
    # overlap of [ts, te] with [w, w + 1] summed over the lineages, in chunks of windows)
    br_torch = torch.empty(W, dtype=torch.float64, device="cuda")
    for w0 in range(0, W, 8):
        lo_w = torch.arange(w0, min(w0 + 8, W), dtype=torch.float64, device="cuda")[:, None]
        ov = (torch.minimum(te[None, :], lo_w + 1.0) - torch.maximum(ts[None, :], lo_w)).clamp_(min=0.0)
        ov[:, te <= ts] = 0.0                                # (a lineage with te <= ts carries no time: lib:74-79)
        br_torch[w0:w0 + ov.shape[0]] = ov.sum(1)
        del ov
    assert torch.allclose(br, br_torch, rtol=1e-11, atol=0)
    # a split, and a permutation: bit-identical (the accumulation is integer)
    cut = 3_333_333
    a = ops.bin_unit_events(ts[:cut], te[:cut], t0, W)
    b = ops.bin_unit_events(ts[cut:].clone(), te[cut:].clone(), t0, W)
    assert torch.equal(a[0] + b[0], sp) and torch.equal(a[1] + b[1], ex)
    assert torch.allclose(a[2] + b[2], br, rtol=1e-15, atol=0)
    perm = torch.randperm(n, device="cuda")
    tsp, tep = ts[perm].contiguous(), te[perm].contiguous()
    p = ops.bin_unit_events(tsp, tep, t0, W)
    assert torch.equal(p[0], sp) and torch.equal(p[1], ex) and torch.equal(p[2], br)
    # ---- lr_bd_loglik_batch on the same lineages ----
    rng = np.random.default_rng(2)
    C = 12                                                  # two passes: a full group of 8 and a ragged one
    lam = np.exp(rng.uniform(np.log(.05), np.log(.6), (C, W)))
    mu = np.exp(rng.uniform(np.log(.02), np.log(.3), (C, W)))
    full = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2))
    # (the statistics of the binned form from torch - bincount and the overlap sums above -, not from this library's binning)
    sp_, ex_, br_ = _np(torch.bincount(torch.floor(ts).long(), minlength=W)[:W]), _np(torch.bincount(be[be < W], minlength=W)[:W]), _np(br_torch)
    keiding = (np.log(lam) * sp_ - lam * br_).sum(1) + (np.log(mu) * ex_ - mu * br_).sum(1)
    assert np.allclose(full, keiding, rtol=REL)
    a = _np(ops.bd_loglik_batch(ts[:cut], te[:cut], t0, lam, mu, 2))
    b = _np(ops.bd_loglik_batch(ts[cut:].clone(), te[cut:].clone(), t0, lam, mu, 2))
    assert np.allclose(a + b, full, rtol=1e-11)
    assert np.allclose(_np(ops.bd_loglik_batch(tsp, tep, t0, lam, mu, 2)), full, rtol=1e-11)
    assert np.array_equal(_np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2)), full)
    solo = _np(ops.bd_loglik_batch(ts, te, t0, lam[5:6], mu[5:6], 2))
    assert solo[0] == pytest.approx(full[5], rel=1e-11)
    one = _np(ops.bd_loglik_batch(ts, te, t0, np.full((1, W), .3), np.full((1, W), .2), 2))
    assert one[0] == pytest.approx(np.log(.3) * sp_.sum() + np.log(.2) * ex_.sum() - .5 * br_.sum(), rel=1e-11)
    # model 0 carries the data constant sum (U + D) log k (LRF:150-162)
    m0 = _np(ops.bd_loglik_batch(ts, te, t0, lam[:2], mu[:2], 0, br_))
    ok = br_ > 0                                            # bins without lineage-time are dropped (LRF:156-160)
    const = ((sp_ + ex_)[ok] * np.log(br_[ok])).sum()
    assert np.allclose(m0, keiding[:2] + const, rtol=REL)


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


def test_loglik_tiled_path_ragged_sizes_vs_oracle(ops):
    """Beyond the one-launch kernel's window (more than 2^18 lineages) lr_bd_loglik_batch runs tiled: table kernel +
    lr_scan_fast_kernel<1 / 2 / 4 / 8, H> - or, for more than 8 chains whose 16 tables fit half a CU's LDS (H <= 136),
    lr_scan_wide_kernel<H>: 16 chains per pass, 512 threads, two pairs of lineages in flight - + reduction of the tile
    partials.  Ragged sizes, ragged chain groups (9, 17, 33 chains: one, two and three groups of sixteen with 7, 15 and 15
    empty slots), non-integer times, models 0 - 2, against the oracle's per-lineage evaluation; unaligned pointers give
    identical bits; a chain scores the same alone, in a group of two, four, eight or sixteen; LR_SCAN_WIDE=0 (the
    eight-chain kernel in a process of its own) gives the same values."""
    import torch
    from literate_amd import _hip
    from oracle import literate_oracle as lo
    lib = _hip.load()
    rng = np.random.default_rng(7)
    NC = 33
    for n, n_bins in ((262_145, 24), (300_001, 120), (280_001, 130), (270_003, 257)):   # H = 40, 136, 264, generic
        t0 = 3.0
        ts = rng.uniform(t0 - 2, t0 + n_bins + 1, n)
        te = ts + rng.exponential(n_bins / 6.0, n)
        te[rng.random(n) < 0.2] = t0 + n_bins + 0.5
        k = rng.integers(0, n, n // 20)
        ts[k] = np.floor(ts[k])
        te[k] = np.ceil(te[k])
        lam = np.exp(rng.uniform(np.log(.02), np.log(1.5), (NC, n_bins)))
        mu = np.exp(rng.uniform(np.log(.02), np.log(1.5), (NC, n_bins)))
        br = _np(ops.bin_events(ts, te, t0 + np.arange(n_bins), t0 + np.arange(n_bins) + 1.0)[2])
        tsd, ted = torch.as_tensor(ts).cuda(), torch.as_tensor(te).cuda()
        plan = (_hip.c_i32 * 4)()
        for C, want in ((8, 8), (9, 16), (NC, 16)):          # sixteen per pass where their tables fit half a CU's LDS (H <= 136)
            assert lib.lr_bd_loglik_plan(n, n_bins, C, 2, plan) == 0
            assert plan[0] == (want if n_bins <= 120 else 4), (n_bins, C, plan[0])
            assert plan[2] == {24: 40, 120: 136, 130: 264}.get(n_bins, n_bins + 2)
        for model in (0, 1, 2):
            ref = np.array([lo.per_lineage_loglik(ts, te, t0, lam[c], mu[c], model, br) for c in range(NC)])
            got = {C: _np(ops.bd_loglik_batch(tsd, ted, t0, lam[:C], mu[:C], model, br)) for C in (1, 2, 3, 4, 8, 9, 16, 17, NC)}
            for C, g in got.items():
                assert np.allclose(g, ref[:C], rtol=REL, atol=1e-9), (n, n_bins, model, C)
                assert np.allclose(g, got[NC][:C], rtol=1e-11, atol=0)
            # chains 16.. of the 33 are scored by the second and third group of sixteen: the same values as on their own
            # (to rounding: the number of lineage tiles - the summation order - depends on the number of chain groups)
            tail = _np(ops.bd_loglik_batch(tsd, ted, t0, lam[16:], mu[16:], model, br))
            assert np.allclose(tail, got[NC][16:], rtol=1e-11, atol=0)
            # the same lineages SORTED BY BIRTH (how the reference's input files come): a wave's 128 lineages then share their
            # birth bin, and the kernels keep that bin's entries in scalar registers instead of gathering them per lane
            # (lr_scan.h: lr_birth_cache) - same operations on the same values
            order = np.argsort(ts, kind="stable")
            tss, tes = torch.as_tensor(ts[order]).cuda(), torch.as_tensor(te[order]).cuda()
            for C in (4, 8, 16, NC):
                g = _np(ops.bd_loglik_batch(tss, tes, t0, lam[:C], mu[:C], model, br))
                assert np.allclose(g, ref[:C], rtol=REL, atol=1e-9), (n, n_bins, model, C, "sorted")
        tsu = torch.as_tensor(np.concatenate([[0.0], ts])).cuda()[1:]
        teu = torch.as_tensor(np.concatenate([[0.0], te])).cuda()[1:]
        for C in (1, 3, 9, 17):
            assert np.array_equal(_np(ops.bd_loglik_batch(tsu, teu, t0, lam[:C], mu[:C], 2)),
                                  _np(ops.bd_loglik_batch(tsd, ted, t0, lam[:C], mu[:C], 2)))


def test_loglik_session_zero_copy_and_small_kernel(ops, G, monkeypatch):
    """The calc_likelihood seam's prepared session (ops.LoglikSession): with few states on few lineages lr_bd_loglik_batch is
    the one-launch kernel and the session runs it on pinned host buffers (rates read, result polled: no copy, no
    synchronisation); beyond 16 states per call, or with LR_LOGLIK_SMALL=0, the tiled three-launch path with copies.  All
    of them against the reference's calc_likelihood values and against each other, models 0-3, repeated calls."""
    name = "metal_bands"
    ts, te = G[name + "/ts"], G[name + "/te"]
    start, end = G[name + "/start_end"]
    L, tL, KL, M, tM, KM = _states(G, name)
    n_bins = len(G[name + "/sp"])
    lam = _np(ops.expand_rates(L, tL, KL, n_bins))
    mu = _np(ops.expand_rates(M, tM, KM, n_bins))
    for model in (0, 1, 2, 3):
        ref = G["%s/lik_model%d" % (name, model)]
        ok = np.isfinite(ref)
        one = ops.LoglikSession(ts, te, float(int(start)), n_bins, 1, model, G[name + "/br"], end)
        four = ops.LoglikSession(ts, te, float(int(start)), n_bins, 4, model, G[name + "/br"], end)
        many = ops.LoglikSession(ts, te, float(int(start)), n_bins, len(ref), model, G[name + "/br"], end)
        assert one.zero_copy and four.zero_copy and not many.zero_copy
        got_many = many(lam, mu).copy()
        assert np.allclose(got_many[ok], ref[ok], rtol=REL, atol=1e-9)
        for rep in range(2):
            for i in range(len(ref)):
                v = one(lam[i], mu[i])[0]
                assert (np.isfinite(v) and v == pytest.approx(ref[i], rel=REL, abs=1e-9)) or not ok[i], (model, i, v, ref[i])
        got4 = four(lam[:4], mu[:4]).copy()
        assert np.allclose(got4[ok[:4]], ref[:4][ok[:4]], rtol=REL, atol=1e-9)
    monkeypatch.setenv("LR_LOGLIK_SMALL", "0")
    plain = ops.LoglikSession(ts, te, float(int(start)), n_bins, 1, 2, G[name + "/br"], end)
    assert not plain.zero_copy
    assert plain(lam[0], mu[0])[0] == pytest.approx(G[name + "/lik_model2"][0], rel=REL)


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


def test_streaming_kernels_at_1e8_lineages(ops):
    """The size bench.py's `abi` rows now reach - 1e8 lineages, 1.6 GB per pass, six times the Infinity Cache - through the
    same size-independent properties: lr_bin_unit_events against torch.bincount and conservation of lineage-time, additive
    over a split bit for bit; lr_bd_loglik_batch with 17 chains (lr_scan_wide_kernel: sixteen per pass, and a second pass of
    one) == binned Keiding on those statistics, additive over the split, bitwise reproducible, a chain the same alone."""
    import torch
    n, W, t0 = 100_000_000, 128, 0.0
    ts, te = _big_lineages(n, True, 5)
    sp, ex, br = ops.bin_unit_events(ts, te, t0, W)
    assert torch.equal(sp, torch.bincount(torch.floor(ts).long(), minlength=W)[:W])
    be = (torch.ceil(te) - 1).long()
    assert torch.equal(ex, torch.bincount(be[be < W], minlength=W)[:W])
    del be
    total = (torch.clamp(te, max=float(W)) - torch.clamp(ts, min=0.0)).clamp(min=0.0).sum()
    assert float(br.sum()) == pytest.approx(float(total), rel=1e-11)
    cut = 41_234_567                                        # (odd: the second part starts on an unaligned pointer)
    a = ops.bin_unit_events(ts[:cut], te[:cut], t0, W)
    b = ops.bin_unit_events(ts[cut:], te[cut:], t0, W)
    assert torch.equal(a[0] + b[0], sp) and torch.equal(a[1] + b[1], ex) and torch.allclose(a[2] + b[2], br, rtol=1e-15, atol=0)
    rng = np.random.default_rng(12)
    C = 17
    lam = np.exp(rng.uniform(np.log(.05), np.log(.6), (C, W)))
    mu = np.exp(rng.uniform(np.log(.02), np.log(.3), (C, W)))
    full = _np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2))
    sp_, ex_, br_ = _np(sp), _np(ex), _np(br)
    keiding = (np.log(lam) * sp_ - lam * br_).sum(1) + (np.log(mu) * ex_ - mu * br_).sum(1)
    assert np.allclose(full, keiding, rtol=REL)
    pa = _np(ops.bd_loglik_batch(ts[:cut], te[:cut], t0, lam, mu, 2))
    pb = _np(ops.bd_loglik_batch(ts[cut:], te[cut:], t0, lam, mu, 2))
    assert np.allclose(pa + pb, full, rtol=1e-11)
    assert np.array_equal(_np(ops.bd_loglik_batch(ts, te, t0, lam, mu, 2)), full)
    assert _np(ops.bd_loglik_batch(ts, te, t0, lam[3:4], mu[3:4], 2))[0] == pytest.approx(full[3], rel=1e-11)


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


def test_device_draws_follow_their_distributions(ops):
    """The engines' own variate functions (Philox -> uniform pair, Box-Muller normal, Marsaglia-Tsang gamma; the reference
    draws from numpy's MT19937 generators: np.random.random / normal / gamma / beta, LRF:29-108) as DISTRIBUTIONS:
    Kolmogorov-Smirnov against scipy on 400k draws per case and the mean within 4.5 standard errors - the posterior
    agreement with the reference rests on these being exact, whatever stream they come from."""
    from scipy import stats
    n = 400_000
    it = np.arange(n, dtype=np.int64)
    for kind, shape, dist in [(0, 1.0, stats.uniform()), (1, 1.0, stats.uniform()), (2, 1.0, stats.norm()), (3, 2.0, stats.gamma(2.0)),
                              (3, 1.2, stats.gamma(1.2)), (3, 3.2, stats.gamma(3.2)), (3, 10.0, stats.gamma(10.0)),
                              (3, 25.2, stats.gamma(25.2)), (3, 0.7, stats.gamma(0.7))]:
        x = _np(ops.debug_draws(97531, 3, it, np.full(n, 5, np.int32), np.full(n, 1, np.int32), np.full(n, kind, np.int32),
                                np.full(n, shape)))
        ks = stats.kstest(x, dist.cdf)
        m, v = dist.stats("mv")
        assert ks.pvalue > 1e-4, (kind, shape, ks)
        assert abs(x.mean() - m) < 4.5 * np.sqrt(v / n), (kind, shape, x.mean(), m)
        assert abs(x.var() / v - 1.0) < 0.02
    # Beta(10, 10) of the add-shift move (LRF:41) is formed from two such gammas
    ga = _np(ops.debug_draws(97531, 3, it, np.full(n, 6, np.int32), np.zeros(n, np.int32), np.full(n, 3, np.int32), np.full(n, 10.0)))
    gb = _np(ops.debug_draws(97531, 3, it, np.full(n, 7, np.int32), np.zeros(n, np.int32), np.full(n, 3, np.int32), np.full(n, 10.0)))
    assert stats.kstest(ga / (ga + gb), stats.beta(10, 10).cdf).pvalue > 1e-4


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
