"""GPU: the drop-in literate_library surface and the CLI end to end, against reference-generated vectors."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(golden_dir):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    return np.load(os.path.join(golden_dir, "binning_lik.npz"))


def test_library_binning_functions(G):
    import literate_library as ll
    for name in ("example_TBP", "metal_bands"):
        ts, te = G[name + "/lib_ts"], G[name + "/lib_te"]
        for a, b, s_, e_, br_ in G[name + "/lib_windows"]:
            assert ll.precompute_events(ts, te, a, b) == (int(s_), int(e_), pytest.approx(br_, rel=1e-12))
            assert ll.get_br(ts, te, a, b) == pytest.approx(br_, rel=1e-12)
        for rm in (0, 1):
            o, p, nb = G["%s/lib_bins_rm%d_meta" % (name, rm)]
            o2, p2, sp, ex, dt, nb2, tr = ll.create_bins(o - rm, p, ts, te, rm)
            assert (o2, p2, nb2) == (o, p, nb) and np.array_equal(tr, np.arange(nb))
            assert np.array_equal(sp, G["%s/lib_bins_rm%d_sp" % (name, rm)])
            assert np.array_equal(ex, G["%s/lib_bins_rm%d_ex" % (name, rm)])
            assert np.array_equal(dt, G["%s/lib_bins_rm%d_dt" % (name, rm)])


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_library_likelihood_operators(G, model):
    import literate_library as ll
    name = "example_TBP"
    sp, ex, br = ll.bind_lineages(G[name + "/ts"], G[name + "/te"], model)
    assert np.array_equal(sp, G[name + "/sp"]) and np.array_equal(br, G[name + "/br"])
    assert ll.n_bins == 24 and ll.start_time == 0.0 and ll.end_time == 24.5
    calc = ll.BDI_partial_lik if model <= 1 else ll.BD_lik_Keiding            # LRF:430-431
    KL, KM = G[name + "/state_KL"], G[name + "/state_KM"]
    for i in range(len(KL)):
        L, M = G[name + "/state_L"][i, :KL[i]], G[name + "/state_M"][i, :KM[i]]
        tL, tM = G[name + "/state_tL"][i, :KL[i] + 1], G[name + "/state_tM"][i, :KM[i] + 1]
        iL, iM = ll.get_rate_index(np.floor(tL)), ll.get_rate_index(np.floor(tM))
        assert np.array_equal(iL, G[name + "/state_indL"][i])
        v = calc(L[iL], M[iM])
        assert isinstance(v, np.float64)
        assert v == pytest.approx(G["%s/lik_model%d" % (name, model)][i], rel=1e-9, abs=1e-9)
        if model == 2:
            fl, fm = np.floor(tL), np.floor(tM)
            fl[-1] = fm[-1] = ll.n_bins
            assert ll.get_BDlik(fl, L, "l") + ll.get_BDlik(fm, M, "m") == pytest.approx(v, rel=1e-9)
    with pytest.raises(SystemExit):
        ll.BD_lik_Keiding(np.ones(5), np.ones(5))           # wrong length: the reference prints and exits


def test_likelihood_operator_reads_br_length_bin_on_every_call(G):
    """The reference's BDI_partial_lik reads the module global br_length_bin when it is CALLED (LRF:150-162): an in-place
    edit and a rebind of literate_library.br_length_bin must both show in the next value (the prepared session used to
    upload it once and was keyed by id() - a stale session, or a recycled id, gave a wrong likelihood without an error).
    The device evaluates the per-lineage form (BDIx:124-146): br_length_bin enters through the constants log k_b and the
    k_b > 0 mask, the exposure comes from the bound lineages themselves - the expected values are the oracle's
    per-lineage evaluator given the same array.  One state per call (the zero-copy one-launch path) and 32 states per
    call (the copy path)."""
    import literate_library as ll
    from oracle import literate_oracle as lo
    name = "metal_bands"
    ts, te = G[name + "/ts"], G[name + "/te"]
    ll.bind_lineages(ts, te, 0)
    nb = ll.n_bins
    rng = np.random.default_rng(8)
    L, M = np.exp(rng.uniform(-3, -1, (32, nb))), np.exp(rng.uniform(-3, -1, (32, nb)))

    def ref():
        br = np.array(ll.br_length_bin)
        return np.array([lo.per_lineage_loglik(ts, te, float(int(ts.min())), L[i], M[i], 0, br) for i in range(32)])
    r0 = ref()
    st = dict(sp=ll.sp_events_bin, ex=ll.ex_events_bin, br=np.array(ll.br_length_bin))
    assert lo.calc_likelihood(0, L[0], M[0], st) == pytest.approx(r0[0], rel=1e-9)      # consistent statistics: the binned form
    assert ll.BDI_partial_lik(L[0], M[0]) == pytest.approx(r0[0], rel=1e-9) and np.allclose(ll.BDI_partial_lik(L, M), r0, rtol=1e-9)
    ll.br_length_bin[3:9] *= 3.0                                 # edited in place
    ll.br_length_bin[nb - 2] += 11.0
    r1 = ref()
    assert abs(r1[0] - r0[0]) > 1e-6 * abs(r0[0])                # (a stale br_length would miss the 1e-9 below by far)
    assert ll.BDI_partial_lik(L[0], M[0]) == pytest.approx(r1[0], rel=1e-9) and np.allclose(ll.BDI_partial_lik(L, M), r1, rtol=1e-9)
    for k in range(6):                                           # rebound (fresh arrays: ids get recycled)
        ll.br_length_bin = ll.br_length_bin * (1.0 + 0.1 * k)
        r2 = ref()
        assert ll.BDI_partial_lik(L[1], M[1]) == pytest.approx(r2[1], rel=1e-9)
    assert np.allclose(ll.BDI_partial_lik(L, M), r2, rtol=1e-9)
    ll.bind_lineages(ts, te, 0)
    assert ll.BDI_partial_lik(L[0], M[0]) == pytest.approx(r0[0], rel=1e-9)


def test_library_proposals_consume_numpy_stream_like_reference(G, golden_dir):
    import literate_library as ll
    with open(os.path.join(golden_dir, "proposals_priors.json")) as f:
        P = json.load(f)
    pr = P["proposals"]
    for r in pr["add"][:12]:
        np.random.seed(r["seed"])
        ra, ti, sc = ll.add_shift_RJ_weighted_mean(np.array(r["rates"]), np.array(r["times"]))
        assert np.allclose(ra, r["out_rates"], rtol=1e-13) and np.allclose(ti, r["out_times"], rtol=1e-15)
        assert sc == pytest.approx(r["score"], rel=1e-11, abs=1e-11)
    for r in pr["remove"][:12]:
        np.random.seed(r["seed"])
        ra, ti, sc = ll.remove_shift_RJ_weighted_mean(np.array(r["rates"]), np.array(r["times"]))
        assert np.allclose(ra, r["out_rates"], rtol=1e-13) and np.array_equal(ti, r["out_times"])
        assert sc == pytest.approx(r["score"], rel=1e-11, abs=1e-11)
    for r in pr["mult"][:12]:
        np.random.seed(r["seed"])
        q, h = ll.update_multiplier_proposal_vec(np.array(r["q"]), 1.1, r["f"])
        assert np.allclose(q, r["out"], rtol=1e-14) and h == pytest.approx(r["hastings"], rel=1e-12, abs=1e-14)
    for r in pr["mult_scalar"][:12]:
        np.random.seed(r["seed"])
        q, h = ll.update_multiplier_proposal(r["q"], 1.1)
        assert q == pytest.approx(r["out"], rel=1e-14) and h == pytest.approx(r["hastings"], rel=1e-12, abs=1e-15)
    for r in P["priors"]["gamma"][:8]:
        assert ll.prior_gamma_LRF(np.array(r["x"]), r["a"], r["b"]) == pytest.approx(r["out"], rel=1e-12, abs=1e-12)
    for r in P["priors"]["poisson"][:20]:
        assert ll.Poisson_prior(r["k"], r["rate"]) == pytest.approx(r["out"], rel=1e-11, abs=1e-11)


def test_cli_end_to_end(G, tmp_path):
    """python LiteRateForward.py -d <file> -TBP ... writes the reference's four logs; the div log is
    byte-identical to the reference's, the chain rows equal the engine/oracle trajectory."""
    from oracle import mcmc_oracle as mo
    # rebuild the example file from the golden arrays (TBP: ts = max - ts_years), 3 columns
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    root = ts.max() if False else 24.0
    data = tmp_path / "example.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, root - a, root - b))
    cmd = [sys.executable, os.path.join(ROOT, "LiteRateForward.py"), "-d", str(data), "-TBP", "-n", "400", "-s", "20",
           "-p", "200", "-seed", "31", "-model_BDI", "2", "--chains", "3", "--combine", "0.25", "--block", "90"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, timeout=300)
    logdir = tmp_path / "literate_mcmc_logs"
    comb = open(logdir / "COMBINED_mcmc.log").read().splitlines()
    assert len(comb) == 1 + 3 * 15 and comb[1].split("\t")[0] == "0" and comb[-1].split("\t")[0] == "44"
    assert len(open(logdir / "COMBINED_sp_rates.log").read().splitlines()) == 45
    assert open(logdir / "COMBINED_div.log").read().splitlines()[0] == "sp_events\tex_events\tbr_length"
    div = open(logdir / "example_BDk_div.log").read().splitlines()
    assert div[0] == "sp_events\tex_events\tbr_length"
    for line, a, b, c in zip(div[1:], G["example_TBP/sp"], G["example_TBP/ex"], G["example_TBP/br"]):
        assert line == "%d\t%d\t%s" % (a, b, c)
    stats = dict(sp=G["example_TBP/sp"], ex=G["example_TBP/ex"], br=G["example_TBP/br"])
    emp = (G["example_TBP/B_EMP"], G["example_TBP/D_EMP"])
    for c in range(3):
        mc = np.loadtxt(logdir / ("example_BDk_c%d_mcmc.log" % c), skiprows=1)
        assert mc.shape == (20, 16)
        with np.errstate(all="ignore"):
            ref = mo.run_mcmc(stats, 0.0, 24.5, mo.Settings(model_BDI=2), mo.PhiloxDraws(31, c), 400, 20, emp=emp, k_max=32)
        assert np.allclose(mc[:, :13], np.array(ref["mcmc"])[:, :13], rtol=1e-9)
        assert np.allclose(mc[:, 13:], np.array(ref["mcmc"])[:, 13:], rtol=1e-8, equal_nan=True)
        sp_rows = [np.array(l.split(), float) for l in open(logdir / ("example_BDk_c%d_sp_rates.log" % c))]
        assert all(np.allclose(a, b, rtol=1e-10) for a, b in zip(sp_rows, ref["sp"]))
    shutil.rmtree(logdir)


def test_cli_logs_are_streamed_and_survive_a_kill(G, tmp_path):
    """The logs are written and flushed window by window while the run goes on (the reference flushes every sample,
    LRF:334-359): the files are read while the CLI is still running, then the process is KILLED mid-run - what it leaves
    are complete, parseable logs whose rows are a prefix of the oracle loop's trajectory, up to the last flushed window."""
    import signal
    import time
    from oracle import mcmc_oracle as mo
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    data = tmp_path / "example.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
    n_total, s, block = 40_000_000, 500, 20_000                 # would take minutes: it is killed long before the end
    cmd = [sys.executable, "-u", os.path.join(ROOT, "LiteRateForward.py"), "-d", str(data), "-TBP", "-n", str(n_total), "-s", str(s),
           "-p", "1000", "-seed", "31", "-model_BDI", "2", "--chains", "2", "--block", str(block)]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    logdir = tmp_path / "literate_mcmc_logs"
    windows = 0
    try:
        t_start = time.time()
        for line in proc.stdout:                                 # the state line of a window follows its flush
            if line.split() and line.split()[0].isdigit() and int(line.split()[0]) % block == 0 and "\t" not in line[:1]:
                windows += 1
                if windows == 3:
                    break
            assert time.time() - t_start < 240
        assert proc.poll() is None                               # still running: the files are read in flight
        in_flight = [open(logdir / ("example_BDk_c%d_mcmc.log" % c)).read() for c in range(2)]
        for text in in_flight:
            assert text.endswith("\n") and len(text.splitlines()) >= 1 + 3 * (block // s)
    finally:
        proc.send_signal(signal.SIGKILL)
        proc.wait()
    stats = dict(sp=G["example_TBP/sp"], ex=G["example_TBP/ex"], br=G["example_TBP/br"])
    emp = (G["example_TBP/B_EMP"], G["example_TBP/D_EMP"])
    for c in range(2):
        text = open(logdir / ("example_BDk_c%d_mcmc.log" % c)).read()
        assert text.endswith("\n") and text.startswith(in_flight[c])      # appended to, never rewritten
        mc = np.loadtxt(logdir / ("example_BDk_c%d_mcmc.log" % c), skiprows=1)
        n_rows = mc.shape[0]
        assert mc.shape[1] == 16 and n_rows >= 3 * (block // s) and n_rows % (block // s) == 0     # whole windows only
        assert np.array_equal(mc[:, 0], np.arange(n_rows) * s)
        sp_rows = [np.array(l.split(), float) for l in open(logdir / ("example_BDk_c%d_sp_rates.log" % c))]
        ex_rows = [np.array(l.split(), float) for l in open(logdir / ("example_BDk_c%d_ex_rates.log" % c))]
        assert abs(len(sp_rows) - n_rows) <= block // s and abs(len(ex_rows) - n_rows) <= block // s   # (killed between files)
        n_chk = 3 * (block // s)
        with np.errstate(all="ignore"):
            ref = mo.run_mcmc(stats, 0.0, 24.5, mo.Settings(model_BDI=2), mo.PhiloxDraws(31, c), n_chk * s, s, emp=emp, k_max=32)
        assert np.allclose(mc[:n_chk, :13], np.array(ref["mcmc"])[:, :13], rtol=1e-9)
        assert all(np.allclose(a, b, rtol=1e-10) for a, b in zip(sp_rows[:n_chk], ref["sp"]))
    shutil.rmtree(logdir)


def test_cli_two_ranks_stream_the_same_logs(G, tmp_path):
    """The sharded CLI (chains split over two ranks, every window's rows gathered to rank 0 on a side stream while the next
    window runs) writes the logs of the one-process run: same rows, same iterations, values equal to 1e-10 (a chain's
    log-likelihood is a sum whose partition depends on how many chains an engine holds: last-digit differences).
    Rehearsed with two ranks on ONE GPU (LR_DIST_BACKEND=gloo: host-staged gather); the real runs use one rank per GPU
    over RCCL."""
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    data = tmp_path / "example.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
    args = [os.path.join(ROOT, "LiteRateForward.py"), "-d", str(data), "-TBP", "-n", "600", "-s", "20", "-p", "200", "-seed", "31",
            "-model_BDI", "2", "--chains", "5", "--block", "130"]
    env = dict(os.environ, LR_SHARED_DEVICE="1")          # (no teams of CUs: the ranks of the rehearsal share one device)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    subprocess.run([sys.executable] + args + ["-out", "_one"], check=True, stdout=subprocess.DEVNULL, timeout=300, env=env)
    env["LR_DIST_BACKEND"] = "gloo"
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29547"] + args + ["-out", "_two"], check=True, stdout=subprocess.DEVNULL, timeout=300, env=env)
    logdir = tmp_path / "literate_mcmc_logs"
    for c in range(5):
        for kind in ("mcmc", "sp_rates", "ex_rates"):
            a = open(logdir / ("example_BDk_one_c%d_%s.log" % (c, kind))).read().splitlines()
            b = open(logdir / ("example_BDk_two_c%d_%s.log" % (c, kind))).read().splitlines()
            assert len(a) == len(b) >= 30
            for la, lb in zip(a[1:] if kind == "mcmc" else a, b[1:] if kind == "mcmc" else b):
                va, vb = np.array(la.split("\t"), float), np.array(lb.split("\t"), float)
                assert va.shape == vb.shape and np.allclose(va, vb, rtol=1e-10, atol=1e-12, equal_nan=True), (c, kind, la, lb)
    shutil.rmtree(logdir)


def test_cli_checkpoint_resume(G, tmp_path):
    """--checkpoint: a run stopped at -n 200 and continued to -n 400 writes the same logs as one run of 400
    (the flags that size the trace, -n and -s, have to match, so the first leg uses the engine API)."""
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    data = tmp_path / "example.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
    base = [sys.executable, os.path.join(ROOT, "LiteRateForward.py"), "-d", str(data), "-TBP", "-n", "400", "-s", "20",
            "-p", "200", "-seed", "31", "--chains", "2"]
    subprocess.run(base + ["-out", "_full"], check=True, stdout=subprocess.DEVNULL, timeout=300)
    ck = str(tmp_path / "ck")
    out = subprocess.run(base + ["-out", "_res", "--checkpoint", ck], check=True, capture_output=True, text=True, timeout=300)
    assert "resumed" not in out.stdout and os.path.exists(ck + ".npz")
    # second invocation finds the finished checkpoint: nothing left to run, logs rewritten from the restored trace
    out = subprocess.run(base + ["-out", "_res", "--checkpoint", ck], check=True, capture_output=True, text=True, timeout=300)
    assert "resumed from" in out.stdout and "at iteration 400" in out.stdout
    logdir = tmp_path / "literate_mcmc_logs"
    for c in range(2):
        for kind in ("mcmc", "sp_rates", "ex_rates"):
            a = open(logdir / ("example_BD_full_c%d_%s.log" % (c, kind))).read()
            b = open(logdir / ("example_BD_res_c%d_%s.log" % (c, kind))).read()
            assert a == b and len(a) > 100
    shutil.rmtree(logdir)


def test_ddrate_cli_end_to_end_and_shipped_log_kat(G, golden_dir, tmp_path):
    """python DDRate.py -d <metal_bands> ...: the log has the reference's header (the shipped DDRate log's own header
    line) and its rows equal the oracle loop's (Philox draws); the HIP rate / likelihood-half kernels reproduce the
    shipped log's per-bin columns and likelihood halves from its logged parameters."""
    from literate_amd import ops
    from oracle import dd_mcmc_oracle as ddo
    from test_oracle_golden import _shipped_ddrate_rows
    rows, s, header = _shipped_ddrate_rows(golden_dir)
    args = np.array([r[0] for r in rows])
    b, d, ni, nf = ops.dd_rates(args, s["br"], 2, 2)
    lb, ld = ops.binned_keiding(b, d, s["sp"], s["ex"], s["br"])
    for i, (_, head, cols, _) in enumerate(rows):
        assert np.allclose(b[i].cpu().numpy(), cols[0], rtol=2e-7) and np.allclose(d[i].cpu().numpy(), cols[1], rtol=2e-7)
        assert np.allclose(ni[i].cpu().numpy(), cols[2], rtol=2e-7) and np.allclose(nf[i].cpu().numpy(), cols[3], rtol=2e-7)
        assert float(lb[i]) == pytest.approx(head[3], rel=1e-7) and float(ld[i]) == pytest.approx(head[4], rel=1e-7)
    pb, pd_ = ops.binned_keiding(np.array([r[2][0] for r in rows]), np.array([r[2][1] for r in rows]), s["sp"], s["ex"], s["br"])
    assert np.allclose(pb.cpu().numpy(), [r[1][3] for r in rows], rtol=2e-12)
    assert np.allclose(pd_.cpu().numpy(), [r[1][4] for r in rows], rtol=2e-12)
    # the CLI on the metal_bands lineages (AD years; te carries the 0.5 jitter in the fixture)
    D = np.load(os.path.join(golden_dir, "ddrate.npz"))
    ts, te = G["metal_bands/lib_ts"], G["metal_bands/lib_te"] - 0.5
    data = tmp_path / "bands.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, c) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, a, c))
    cmd = [sys.executable, os.path.join(ROOT, "DDRate.py"), "-d", str(data), "-n", "300", "-s", "10", "-p", "100",
           "-seed", "21", "--chains", "2", "--block", "70"]          # five windows, the log appended and fsynced per window
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, timeout=300)
    origin, present, _ = D["meta"]
    with np.errstate(all="ignore"):
        emp = (D["N_SPEC"] / D["DT"], D["N_EXTI"] / D["DT"])
    for c in range(2):
        path = tmp_path / ("bands_21_LDDN_MDDN_c%d.log" % c)
        lines = open(path).read().splitlines()
        assert lines[0] == header
        got = np.array([[float(v) for v in l.split("\t")] for l in lines[1:]])
        ref = np.array(ddo.run_dd_mcmc(D["N_SPEC"], D["N_EXTI"], D["DT"], D["TIME_RANGE"], origin, present, 2, 2,
                                       ddo.PhiloxDraws(21, c), 300, 10, emp=emp))
        assert got.shape == ref.shape == (30, 145)
        assert np.allclose(got[:, :-3], ref[:, :-3], rtol=1e-9, atol=1e-9)
        assert np.allclose(got[:, -3:], ref[:, -3:], rtol=1e-7, atol=1e-9, equal_nan=True)


def test_trend_rate_cli_end_to_end(G, golden_dir, tmp_path):
    """python trend_rate.py -d <metal_bands> -trend_data <file> ...: the log has the reference's own header line (from a
    reference run) and its rows equal the oracle loop's (Philox draws)."""
    from oracle import trend_mcmc_oracle as tro
    R = np.load(os.path.join(golden_dir, "ratemaps.npz"))
    T = np.load(os.path.join(golden_dir, "trend_trajectories.npz"))
    ts, te = G["metal_bands/lib_ts"], G["metal_bands/lib_te"] - 0.5
    data = tmp_path / "bands.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, c) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, a, c))
    trend_file = tmp_path / "trend.tsv"
    with open(trend_file, "w") as f:
        f.write("year\ttrend\n")
        for i, v in enumerate(R["trend_raw"]):
            f.write("%d\t%r\n" % (i, float(v)))
    cmd = [sys.executable, os.path.join(ROOT, "trend_rate.py"), "-d", str(data), "-n", "300", "-s", "10", "-p", "100",
           "-seed", "23", "-trend_data", str(trend_file), "-trend_index", "1", "--chains", "2"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, timeout=300)
    with np.errstate(all="ignore"):
        emp = (R["N_SPEC"] / R["DT"], R["N_EXTI"] / R["DT"])
    for c in range(2):
        lines = open(tmp_path / ("bands_23_EXPB_EXPD_c%d_1.trendrate.log" % c)).read().splitlines()
        assert lines[0] == str(T["cb0_cd0_s4/header"])
        got = np.array([[float(v) for v in l.split("\t")] for l in lines[1:]])
        ref = np.array(tro.run_trend_mcmc(R["N_SPEC"], R["N_EXTI"], R["DT"], R["TREND"], tro.PhiloxDraws(23, c), 300, 10, emp=emp))
        assert got.shape == ref.shape == (30, 12 + 2 * len(R["DT"]) + 3)
        assert np.allclose(got[:, :-3], ref[:, :-3], rtol=1e-9, atol=1e-9)
        assert np.allclose(got[:, -3:], ref[:, -3:], rtol=1e-7, atol=1e-9, equal_nan=True)


def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py's multi-rank path (chain offsets per rank, barrier + max-over-ranks timing, gather of the sampled
    rows to rank 0, one JSON line from rank 0) rehearsed with two ranks on ONE GPU: LR_DIST_BACKEND=gloo maps all
    ranks to device 0 and stages the gather through the host.  The real runs use one rank per GPU over RCCL."""
    import json
    env = dict(os.environ, LR_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    # plain `python bench.py --gpus 2`, as the driver invokes it: bench.py starts its own two ranks (before touching the GPU)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "300", "--warmup", "50",
           "--chains", "64", "--workload", "cfg3", "--no-cpu-baseline"]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300, env=env).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    # the line the driver parses is the LAST one and compact (the detail goes to `#detail` lines before it)
    assert out.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["chains_total"] == 128 and b["config"]["chains_per_gpu"] == 64
    assert b["scaling"] == "weak" and b["value"] > 0 and b["cpu_baseline"] is None
    assert b["value"] == pytest.approx(300 * b["config"]["lineages"] * 128 / (b["ms_per_step"] * 1e-3 * 300))
    # the second leg of a multi-rank run: the same 64 chains IN TOTAL, sharded over the two ranks (strong scaling, the way
    # BASELINE.json words cfg4: "1024 chains sharded across 8 GPUs")
    s = b["strong_scaling"]
    assert s["chains_total"] == 64 and s["chains_per_gpu"] == 32 and s["value"] > 0
    assert s["value"] == pytest.approx(300 * b["config"]["lineages"] * 64 / (s["ms_per_step"] * 1e-3 * 300))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_rccl_on_the_one_gpu():
    """RCCL on hardware (every other multi-rank test here runs under gloo): a world-size-1 "nccl" process group on the one
    leased GPU.  (i) a child started by torch.distributed.run creates the communicator on cuda:0 and runs the engine's
    sharded path - TraceStreamer(gather=True) over two windows: status all-reduce + dist.gather on the side stream - with
    librccl mapped into the process; (ii) `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`, the driver's
    multi-GPU command at N = 1: bench.py then keeps its process group, barriers and the device-side gather of the sampled
    rows, and says so in config.process_group."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LR_DIST_BACKEND", "LR_SHARED_DEVICE"):
        env.pop(k, None)
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    r = subprocess.run(launcher + ["--master-port", _free_port(), os.path.join(ROOT, "tests", "helpers", "rccl_child.py")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_CHILD ")]
    assert len(line) == 1, r.stdout[-2000:]
    c = json.loads(line[0][len("RCCL_CHILD "):])
    assert c["rccl_loaded"] and c["backend"] == "nccl" and c["rows"] == 40 and c["all_reduce"] == [0.0, 1.0, 2.0, 3.0]
    r = subprocess.run(launcher + ["--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "300",
                                   "--warmup", "50", "--workload", "cfg3", "--chains", "64", "--sample-every", "10",
                                   "--no-cpu-baseline", "--no-configs", "--no-pmc", "--no-abi"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192
    b = json.loads(lines[0])
    assert b["n_gpus"] == 1 and b["config"]["process_group"] == "nccl" and b["config"]["trace_rows_gathered_in_region"] == 30
    assert b["value"] == pytest.approx(300 * b["config"]["lineages"] * 64 / (b["ms_per_step"] * 1e-3 * 300))


def test_cli_cfg1_fixed_two_shifts(G, tmp_path):
    """BASELINE.json configs[0]: example_dataTBP, 1 chain, fixed 2 rate shifts (-const_rates 1 with a
    3-rate initial state, SURVEY 8c 'config-1 note').  K stays (3, 3), shift times never move (A9), the
    rows equal the oracle loop started from the same state."""
    from oracle import mcmc_oracle as mo
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    data = tmp_path / "example.tsv"
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
    cmd = [sys.executable, os.path.join(ROOT, "LiteRateForward.py"), "-d", str(data), "-TBP", "-n", "3000", "-s", "50",
           "-p", "1000", "-seed", "17", "-const_rates", "1", "--init_shifts", "2", "--chains", "1", "-calc_adequacy", "0"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, timeout=300)
    logdir = tmp_path / "literate_mcmc_logs"
    mc = np.loadtxt(logdir / "example_BD_mcmc.log", skiprows=1)          # reference file name for 1 chain
    assert mc.shape == (60, 13) and np.all(mc[:, 6] == 3) and np.all(mc[:, 7] == 3)
    sp_rows = [np.array(l.split(), float) for l in open(logdir / "example_BD_sp_rates.log")]
    t = np.linspace(0.0, 24.5, 4)
    assert all(np.array_equal(r[3:], t[1:3]) for r in sp_rows)           # shift times fixed
    rng = np.random.default_rng(17)
    L0, M0 = rng.gamma(2, 2, 3), rng.gamma(2, 2, 3)
    stats = dict(sp=G["example_TBP/sp"], ex=G["example_TBP/ex"], br=G["example_TBP/br"])
    with np.errstate(all="ignore"):
        ref = mo.run_mcmc(stats, 0.0, 24.5, mo.Settings(model_BDI=0, const_rates=1), mo.PhiloxDraws(17, 0), 3000, 50,
                          init=(L0, M0, t, t), k_max=32)
    assert np.allclose(mc, np.array(ref["mcmc"])[:, :13], rtol=1e-9)
    assert all(np.allclose(a, b, rtol=1e-10) for a, b in zip(sp_rows, ref["sp"]))
