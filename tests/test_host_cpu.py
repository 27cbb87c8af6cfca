"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the product has no
CPU compute path, host logic (sharding, log writers, marginal rates), world_size-2 gloo gather."""
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from literate_amd import _hip
    from literate_amd.build import build_hip
    lib_path = build_hip()
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "literate_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(lr_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _hip.SIGNATURES, "ctypes binding missing for " + name
    assert _hip.load().lr_version() >= 100
    # layout query is pure host code: check it without a GPU
    cfg = _hip.McmcConfig(n_lineages=100000, n_bins=128, n_chains=1024, model=0, s_freq=100, n_trace_slots=10,
                          t0=0.0, start_time=0.0, end_time=128.5, seed=1)
    lay = _hip.McmcLayout()
    # general lineage times: the four-chain persistent kernel on pair-general tables (32-bit fixed-point fractions) ...
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0
    assert lay.persistent == 2 and lay.table_mode == 2 and lay.chains_per_block == 4 and lay.table_stride == 2 * 136
    assert lay.lineage_frac > lay.lineage_idx > 0 and lay.pack_tmp > lay.lineage_frac
    # ... or, forced, the launch-based engine on chain-major tables with fp64 fractions
    cfg.engine_mode = 1
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0
    assert lay.persistent == 0 and lay.table_mode == 0
    assert lay.chains_per_block == 8 and lay.table_stride == 2 * 136 and lay.trace_width == _hip.LR_TRACE_W
    assert lay.total_bytes > lay.trace > lay.partials > lay.tables > lay.bin_consts > lay.state_i32 > 0
    assert lay.n_parts == 2 and lay.pipelined == 1
    cfg.engine_mode = 0
    cfg.unit_resolution, cfg.frac_death = 1, 0.5                  # unit-resolution tables: 8-byte entries, 16 chains/block
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0
    assert lay.chains_per_block == 16 and lay.table_stride == 136 and lay.pipelined == 1
    assert lay.persistent == 2 and lay.lineage_idx > 0 and lay.table_mode == 1   # auto: four-chain kernel for 1024 chains x 100k
    cfg.n_chains = 128                                            # a 128-chain shard: speculative kernel, a team of 2 CUs per CHAIN
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0
    assert lay.persistent == 3 and lay.team_blocks == 2 and lay.spec_chains_per_team == 1 and lay.xchg > 0 and lay.reserved1 == 768
    cfg.team_request = 4 | (2 << 8)                               # ... or, asked for, a team of 4 CUs per chain PAIR
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0
    assert lay.persistent == 3 and lay.team_blocks == 4 and lay.spec_chains_per_team == 2 and lay.reserved1 == 768
    cfg.team_request = 0
    cfg.n_chains = 1024
    cfg.engine_mode = 1
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.persistent == 0
    # few chains, huge input at unit resolution: tiled launches whose scan reads the PACKED lineages once for all chains
    # (two rounds of one 1024-thread block per CU) ...
    cfg.engine_mode, cfg.n_chains, cfg.n_lineages = 0, 16, 10_000_000
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.persistent == 0 and lay.streaming == 0 and lay.packed_scan == 1
    assert lay.tiles <= 512 and lay.n_parts == 1 and lay.pipelined == 0 and lay.chains_per_block == 8 and lay.lineage_idx > 0
    # (long scans - 16 chains x 1e8 lineages - run in two partitions of eight chains on their own streams: one's step kernel
    # under the other's scan)
    cfg.n_lineages = 100_000_000
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.packed_scan == 1 and lay.n_parts == 2 and lay.pipelined == 0
    cfg.n_lineages = 10_000_000
    # ... engine_mode 1: the scan of ts / te (one round of four 256-thread blocks per CU)
    cfg.engine_mode = 1
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.persistent == 0 and lay.streaming == 0 and lay.packed_scan == 0
    tiles_launch, xchg_launch = lay.tiles, lay.total_bytes - lay.xchg
    assert 900 <= tiles_launch <= 1024
    # ... or, asked for (engine="stream"), the same plan's iterations inside the resident streaming kernel: tiles sized to
    # the device's block slots less the stepper blocks, a second table buffer + the launch's counters in xchg
    cfg.engine_mode = 6
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.persistent == 0 and lay.streaming == 1 and lay.packed_scan == 0 and lay.n_parts == 1 and lay.pipelined == 0
    assert lay.tiles <= 1024 - 4 and lay.tiles <= tiles_launch and lay.total_bytes - lay.xchg > xchg_launch + 16 * 136 * 16
    cfg.n_chains = 64                                                 # (chains for two halves: the pipelined launches stay)
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == 0 and lay.streaming == 0 and lay.packed_scan == 0 and lay.pipelined == 1
    cfg.engine_mode = 0
    cfg.n_chains, cfg.n_lineages = 1024, 100000
    cfg.t0 = 0.5
    assert _hip.load().lr_mcmc_query_layout(cfg, lay) == -5        # LR_ERR_T0
    assert _hip.load().lr_bd_loglik_workspace_bytes(1000, 5000, 4, 2) == -2   # LR_ERR_SIZE


def test_header_constants_match_python_binding():
    from literate_amd import _hip
    header = open(os.path.join(ROOT, "include", "literate_hip.h")).read()
    defs = dict(re.findall(r"#define\s+(LR_\w+)\s+(-?\d+)\b", header))
    assert int(defs["LR_KMAX"]) == _hip.LR_KMAX and int(defs["LR_ROW"]) == _hip.LR_ROW
    assert int(defs["LR_STATE_ROWS"]) == _hip.LR_STATE_ROWS and int(defs["LR_ISTATE_ROWS"]) == _hip.LR_ISTATE_ROWS
    for name in ("S_LIKA", "S_PRIORA", "S_POI", "S_LIK_P", "I_KL", "I_PKM", "I_IT_HI", "I_MOVE"):
        assert int(defs["LR_" + name]) == getattr(_hip, name)
    for name in ("ROW_L", "ROW_PTM", "ROW_SCALARS", "IROW_EL", "IROW_SCALARS"):
        assert int(defs["LR_" + name]) == getattr(_hip, name)


def test_no_cpu_compute_path():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from literate_amd import _hip, ops
    import literate_library as ll
    with pytest.raises(_hip.HipLibraryError):
        ops.bin_events(np.zeros(3), np.ones(3), [0.0], [1.0])
    with pytest.raises(_hip.HipLibraryError):
        ll.precompute_events(np.zeros(3), np.ones(3), 0, 1)
    with pytest.raises(_hip.HipLibraryError):
        ll.bind_lineages(np.zeros(3), np.ones(3))
    with pytest.raises(NameError):
        ll.BDI_partial_lik(np.ones(3), np.ones(3))     # data not bound (the reference raises NameError too)
    # the product never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "literate_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src, f
    for f in ("LiteRateForward.py", "DDRate.py", "trend_rate.py", "literate_library.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read(), f
    with pytest.raises(_hip.HipLibraryError):
        ops.simulate_bd(10, 4, 1, np.zeros(4), np.zeros(4))
    with pytest.raises(_hip.HipLibraryError):
        ops.dd_rates(np.ones(8), np.ones(5))


def test_drop_in_surface_names():
    import literate_library as ll
    for name in ["BD_partial_lik", "BDI_partial_lik", "BD_lik_Keiding", "precompute_events", "get_br", "create_bins",
                 "get_rate_index", "update_multiplier_proposal", "update_multiplier_proposal_vec", "parse_ts_te",
                 "core_arguments", "set_seed", "calculate_r_squared", "calcHPD"]:
        assert callable(getattr(ll, name)), name
    a = ll.core_arguments().parse_args(["-d", "x.tsv"])
    assert (a.n, a.p, a.s, a.seed, a.death_jitter, a.rm_first_bin, a.TBP) == (10000000, 1000, 1000, -1, .5, 0, False)
    import LiteRateForward as cli
    b = cli.build_parser().parse_args(["-d", "x.tsv"])
    assert (b.n, b.s, b.model_BDI, b.use_rate_HP, b.Poisson_prior, b.update_fraction, b.calc_adequacy,
            b.const_rates, b.const_death_rate, b.death_jitter, b.chains) == (10000000, 1000, 0, 1, 0, 0.75, 1, 0, 0, .5, 1)


def test_shard_chains():
    from literate_amd.dist import shard_chains
    for total, world in ((1024, 8), (10, 3), (1, 4), (7, 7)):
        spans = [shard_chains(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (o1, n1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + n1 == o2


def _trace_row(it, L, tL, M, tM, start, end):
    from literate_amd import _hip
    K = _hip.LR_KMAX
    row = np.full(_hip.LR_TRACE_W, np.nan)
    row[:13] = [it, -10.5, -8.25, -2.25, np.mean(L), np.mean(M), len(L), len(M), start, end, 1.0, 1.5, 2.0]
    rl = row[13:13 + 2 * K - 1]; rm = row[13 + 2 * K - 1:]
    rl[:len(L)] = L; rl[K:K + len(L) - 1] = tL[1:-1]
    rm[:len(M)] = M; rm[K:K + len(M) - 1] = tM[1:-1]
    return row


def test_native_number_format_is_python_str_of_float():
    """lr_format_rows (csrc/lr_format.hip, host only) writes every double exactly as Python's str(float) does - what the
    reference's csv writer emits (LRF:334-359): random bit patterns, log-uniform magnitudes, the notation switches at 1e-4 and
    1e16, subnormals, the largest double, whole numbers, zeros, nan and inf; integer columns; ragged and empty rows."""
    from literate_amd import logs
    if logs._native() is None:
        pytest.fail("libliterate_hip.so does not load or lacks lr_format_rows")
    rng = np.random.default_rng(0)
    y = rng.integers(0, 2 ** 63, 300_000, dtype=np.int64).view(np.float64)
    y = y[np.isfinite(y)]
    y = np.concatenate([y, -y, np.exp(rng.uniform(np.log(1e-12), np.log(1e20), 300_000)), rng.normal(-300, 20, 100_000),
                        rng.integers(-10 ** 6, 10 ** 6, 50_000).astype(float), 10.0 ** np.arange(-30, 31), 9.999999999999999 * 10.0 ** np.arange(-8, 20),
                        [0.0, -0.0, 1e16, 1e15, 9999999999999998.0, 1e-4, 9.999e-5, 1e-5, 123456789012345680.0, 5e-324, 2.2250738585072014e-308,
                         1.7976931348623157e308, 24.0, 0.5, 1 / 3, np.nan, np.inf, -np.inf, 1e22, 1e23, 2.5e-5, -2.5e-5, 12345678.9]])
    got = logs._native_format(y, np.arange(len(y) + 1))
    assert got == "".join(str(v) + "\n" for v in y.tolist()).encode()
    assert logs._native_format([3000.0, -1.5, 2.0, 7.0, 1e-5, -12.0], [0, 3, 3, 6], 0b101) == b"3000\t-1.5\t2\n\n7\t1e-05\t-12\n"
    assert logs._native_format([], [0]) == b"" and logs._native_format([], [0, 0]) == b"\n"


def test_table_logs_native_and_csv_writer_write_the_same_bytes(tmp_path, monkeypatch):
    """logs.append_table_logs (the DDRate.py / trend_rate.py logs: fixed-width rows, column 0 an integer) through
    lr_format_rows and through csv.writer(delimiter='\\t') as the reference uses it (DD:236-238): the same bytes,
    '\\r\\n' line ends included; several chains, an empty window."""
    import csv
    from literate_amd import logs
    rng = np.random.default_rng(11)
    tables = rng.normal(0, 1, (6, 37, 23)) * 10.0 ** rng.integers(-7, 17, (6, 37, 23))
    tables[..., 0] = np.arange(37) * 1000
    tables[2, 5, 7], tables[3, 1, 2], tables[0, 0, 3] = np.nan, np.inf, -0.0
    out = {}
    for fmt in ("native", "python"):
        monkeypatch.setenv("LR_LOG_FORMAT", fmt)
        monkeypatch.setattr(logs, "_NATIVE", [])
        paths = [str(tmp_path / ("%s_%d.log" % (fmt, c))) for c in range(6)]
        logs.append_table_logs(paths, tables[:, :20]), logs.append_table_logs(paths, tables[:, 20:]), logs.append_table_logs(paths, tables[:, :0])
        out[fmt] = [open(p, "rb").read() for p in paths]
    assert out["native"] == out["python"]
    with open(tmp_path / "ref.log", "w", newline="") as f:
        w = csv.writer(f, delimiter="\t")
        for row in tables[4]:
            w.writerow([int(row[0])] + [float(v) for v in row[1:]])
    assert open(tmp_path / "ref.log", "rb").read() == out["native"][4] and out["native"][4].count(b"\r\n") == 37


@pytest.mark.parametrize("fmt", ["native", "python"])
def test_window_log_writer_is_byte_identical_to_a_row_by_row_formatter(tmp_path, monkeypatch, fmt):
    """ChainLogWriter.append formats a whole window at once (the adequacy columns of all rows in one set of array operations;
    the numbers by lr_format_rows, or - LR_LOG_FORMAT=python, or without the library - by str on one tolist()): its files
    are byte for byte what a plain per-row formatter in the reference's style writes - csv-style `str(float)` numbers,
    calculate_r_squared (lib:268-279) per row -, with and without -pyrate_output, with NaN / inf in the empirical rates, and
    when a row's shift times do not ascend (the per-row fallback)."""
    from literate_amd import logs, _hip
    monkeypatch.setenv("LR_LOG_FORMAT", fmt)
    monkeypatch.setattr(logs, "_NATIVE", [])
    assert (logs._native() is not None) == (fmt == "native")
    rng = np.random.default_rng(5)
    start, end, n_bins, S, C = 0.0, 24.5, 24, 40, 5
    rows = np.empty((S, C, _hip.LR_TRACE_W))
    for s_ in range(S):
        for c in range(C):
            kl, km = int(rng.integers(1, 7)), int(rng.integers(1, 4))
            tL = np.concatenate([[start], np.sort(rng.uniform(0.2, 24.2, kl - 1)), [end]])
            tM = np.concatenate([[start], np.sort(rng.uniform(0.2, 24.2, km - 1)), [end]])
            rows[s_, c] = _trace_row(s_ * 1000, rng.gamma(2, .1, kl) * 10.0 ** rng.integers(-6, 3), tL, rng.gamma(2, .1, km), tM, start, end)
            rows[s_, c, 1:4] = rng.normal(0, 1, 3) * 10.0 ** rng.integers(-8, 18)      # both notations of str(float)
    emp = (rng.gamma(2, .1, n_bins), rng.gamma(2, .1, n_bins))
    emp[0][3], emp[1][7] = np.nan, np.inf

    def reference_lines(chain_rows, emp_, pyrate, root):
        lm, ls, le = [], [], []
        for row in chain_rows:
            head, sp, ex = logs.split_row(row)
            kl, km = int(head[6]), int(head[7])
            v = [str(int(head[0]))] + [str(float(x)) for x in head[1:6]] + [str(kl), str(km)]
            v += [str(float(root)), str(float(root - head[9]))] if pyrate else [str(float(head[8])), str(float(head[9]))]
            v += [str(float(x)) for x in head[10:13]]
            if emp_ is not None:
                with np.errstate(all="ignore"):
                    v += [str(float(x)) for x in logs.adequacy(emp_[0], emp_[1], logs.rates_per_bin(sp[:kl], sp[kl:], head[8], n_bins),
                                                              logs.rates_per_bin(ex[:km], ex[km:], head[8], n_bins))]
            lm.append("\t".join(v) + "\n")
            if pyrate:
                sp, ex = np.concatenate([sp[:kl], root - sp[kl:]]), np.concatenate([ex[:km], root - ex[km:]])
            ls.append("\t".join(str(float(x)) for x in sp) + "\n")
            le.append("\t".join(str(float(x)) for x in ex) + "\n")
        return {"mcmc": lm, "sp_rates": ls, "ex_rates": le}

    os.mkdir(tmp_path / "literate_mcmc_logs")
    unsorted = rows.copy()
    unsorted[7, 2, 13 + _hip.LR_KMAX:13 + _hip.LR_KMAX + 2] = [20.0, 3.0]        # a row whose shift times descend
    unsorted[7, 2, 6] = 3
    unsorted[7, 2, 13:16] = [.1, .2, .3]
    for tag, R, emp_, pyrate in (("a", rows, emp, False), ("b", rows, emp, True), ("c", rows, None, False), ("d", unsorted, emp, False)):
        w = logs.ChainLogWriter(str(tmp_path / "toy.tsv"), 0, tag, C, emp_, n_bins, pyrate, 30.25)
        w.append(R[:25]), w.append(R[25:]), w.append(R[:0])                  # two windows and an empty one
        for c in range(C):
            want = reference_lines(R[:, c], emp_, pyrate, 30.25)
            for key in ("mcmc", "sp_rates", "ex_rates"):
                got = open(w.paths[c][key]).read().splitlines(keepends=True)
                assert got[1 if key == "mcmc" else 0:] == want[key], (tag, c, key)
        # one chain's rows through write_chain_logs: the same lines
        p1 = logs.log_paths(str(tmp_path / "toy.tsv"), 0, tag + "_one")[1]
        logs.write_chain_logs(p1, R[:, 1], emp_, n_bins, pyrate, 30.25)
        assert open(p1["mcmc"]).read().splitlines(keepends=True)[1:] == reference_lines(R[:, 1], emp_, pyrate, 30.25)["mcmc"]


def test_log_writers_and_marginal_rates(tmp_path):
    from literate_amd import logs
    from oracle import literate_oracle as lo
    rng = np.random.default_rng(0)
    start, end, n_bins = 0.0, 24.5, 24
    rows, sp_rows = [], []
    for i in range(60):
        k = int(rng.integers(1, 6))
        tL = np.concatenate([[start], np.sort(rng.uniform(1, 23, k - 1)), [end]])
        L = rng.uniform(.05, .6, k)
        rows.append(_trace_row(i * 10, L, tL, np.array([.2]), np.array([start, end]), start, end))
        sp_rows.append(np.concatenate([L, tL[1:-1]]))
    rows = np.array(rows)
    head, sp, ex = logs.split_row(rows[3])
    assert np.array_equal(sp, sp_rows[3]) and np.array_equal(ex, [.2])
    data = tmp_path / "toy.tsv"
    data.write_text("id\tts\tte\n")
    out_dir, paths = logs.log_paths(str(data), 0, "")
    assert paths["mcmc"].endswith("literate_mcmc_logs/toy_BD_mcmc.log")
    assert logs.log_paths(str(data), 2, "_x", 3)[1]["sp_rates"].endswith("toy_BDk_x_c3_sp_rates.log")
    os.mkdir(out_dir)
    emp = (rng.uniform(.1, 1, n_bins), rng.uniform(.1, 1, n_bins))
    logs.write_chain_logs(paths, rows, emp, n_bins)
    lines = open(paths["mcmc"]).read().splitlines()
    assert lines[0].split("\t") == logs.MCMC_HEAD + logs.ADEQUACY_HEAD and len(lines) == 61
    assert lines[1].split("\t")[0] == "0" and lines[1].split("\t")[6] in "12345"
    back = [np.array(l.split(), float) for l in open(paths["sp_rates"])]
    assert all(np.array_equal(a, b) for a, b in zip(back, sp_rows))
    # adequacy columns = oracle's statistic on the reconstructed per-bin rates
    cols = np.array(lines[4].split("\t"), float)
    k = int(cols[6])
    lam = sp_rows[3][:k][lo.get_rate_index(np.concatenate([[start], np.floor(sp_rows[3][k:]), [np.floor(end)]]), n_bins)] \
        if k > 1 else np.full(n_bins, sp_rows[3][0])
    ref = lo.calculate_r_squared(emp[0], emp[1], lam, np.full(n_bins, .2))
    assert np.allclose(cols[13:], ref, rtol=1e-12)
    # marginal rates: product restatement == oracle restatement of plotRJforward.v3.py:92-139
    a = logs.marginal_rates(sp_rows, end, start)
    b = lo.marginal_rates_from_rows(sp_rows, end, start)
    assert np.array_equal(a[1], b[0]) and np.array_equal(a[2], b[1]) and np.array_equal(a[4], b[3])
    logs.write_div_log(paths["div"], np.arange(3), np.arange(3), np.array([1.5, 2.0, 3.0]))
    assert open(paths["div"]).read().splitlines()[1] == "0\t0\t1.5"
    # combine_logs (plotRJforward.v3.py:307-350): two chains pooled, 25 % burn-in per file, `it` renumbered
    files = []
    for c in range(2):
        _, pc = logs.log_paths(str(data), 0, "", c)
        logs.write_chain_logs(pc, rows[c * 20:c * 20 + 20], emp, n_bins)
        files.append(pc["mcmc"])
    logs.combine_logs(files, out_dir, 0.25)
    comb = open(os.path.join(out_dir, "COMBINED_mcmc.log")).read().splitlines()
    assert comb[0] == lines[0] and len(comb) == 1 + 2 * 15
    assert [l.split("\t")[0] for l in comb[1:]] == [str(i) for i in range(30)]
    assert comb[1].split("\t")[1:] == open(files[0]).read().splitlines()[6].split("\t")[1:]
    sp_comb = open(os.path.join(out_dir, "COMBINED_sp_rates.log")).read().splitlines()
    assert len(sp_comb) == 30 and np.array_equal(np.array(sp_comb[15].split(), float), sp_rows[25])
    assert open(os.path.join(out_dir, "COMBINED_div.log")).read().splitlines()[3] == "2.0\t2.0\t3.0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, total, ret):
    import torch
    import torch.distributed as dist
    from literate_amd.dist import gather_traces, shard_chains
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    off, n = shard_chains(total, world, rank)
    local = torch.zeros(4, n, 5, dtype=torch.float64)
    for c in range(n):
        local[:, c, :] = float(off + c) + torch.arange(4, dtype=torch.float64)[:, None] * 100
    out = gather_traces(local, total)
    # the engine status word agreed over the ranks before a window's gather (TraceStreamer.collect): a rank whose team
    # exchange timed out (status 1) is seen by every rank, so that all raise alike and none is left waiting in the gather
    from literate_amd.dist import agree_status
    agreed = [agree_status(0, "cpu"), agree_status(1 if rank == 1 else 0, "cpu"), agree_status(0, "cpu")]
    if rank == 0:
        ok = out.shape == (4, total, 5) and all(float(out[s, c, 0]) == c + 100 * s for s in range(4) for c in range(total))
        ret.put(bool(ok) and agreed == [0, 1, 0])
    else:
        assert out is None and agreed == [0, 1, 0]
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_trace_gather_world2_gloo(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, total, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(timeout=5) is True


def test_planner_refuses_teams_beyond_the_device():
    """The speculative kernel's teams spin on each other's partial sums, so all blocks of a launch must be resident:
    the planner never plans more blocks (chain pairs x team size) than the device has compute units, drops the
    speculative kernel when the pairs alone exceed them, and refuses a team size that is not 1, 2, 4 or 8."""
    import ctypes as C
    from literate_amd import _hip
    lib = _hip.load()

    def layout(n_chains, team=0, cus=None, engine=0):
        cfg = _hip.McmcConfig(n_lineages=100_000, n_bins=128, n_chains=n_chains, model=0, use_rate_HP=1, s_freq=100,
                              n_trace_slots=4, update_fraction=0.75, t0=0.0, start_time=0.0, end_time=128.5, seed=1,
                              unit_resolution=1, frac_birth=0.0, frac_death=0.5, team_request=team, engine_mode=engine)
        lay = _hip.McmcLayout()
        if cus is None:
            os.environ.pop("LR_DEVICE_CUS", None)
        else:
            os.environ["LR_DEVICE_CUS"] = str(cus)
        try:
            return lib.lr_mcmc_query_layout(C.byref(cfg), C.byref(lay)), lay
        finally:
            os.environ.pop("LR_DEVICE_CUS", None)

    rc, lay = layout(128)
    assert rc == 0 and lay.persistent == 3 and lay.team_blocks == 2 and lay.spec_chains_per_team == 1   # 128 chains x 2 = 256 blocks on 256 CUs
    for cus in (128, 100, 64):
        rc, lay = layout(128, cus=cus)
        teams = 128 if lay.spec_chains_per_team == 1 else 64
        assert rc == 0 and lay.persistent == 3 and teams * lay.team_blocks <= cus, (cus, lay.team_blocks, lay.spec_chains_per_team)
    rc, lay = layout(128, cus=32)                                               # fewer CUs than chain pairs: no teams at all
    assert rc == 0 and lay.persistent != 3
    rc, lay = layout(128, team=8)                                               # 64 x 8 > 256: the request cannot be met
    assert rc == 0 and not (lay.persistent == 3 and lay.team_blocks == 8)
    rc, lay = layout(128, team=3)
    assert rc == _hip.LR_ERR_SIZE
    rc, lay = layout(16, team=8, engine=5)
    assert rc == 0 and lay.persistent == 3 and lay.team_blocks == 8


def test_bench_gpus_n_starts_its_own_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher's environment (how the driver invokes it) must start N ranks itself
    - torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 - from a parent that has not imported torch,
    and exit with the children's code."""
    import importlib
    import sys
    monkeypatch.syspath_prepend(ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch_loaded_before:
        assert "torch" not in sys.modules          # the parent never got as far as importing torch


def test_parse_ts_te_against_the_reference_outcomes(tmp_path, golden_dir):
    """literate_library.parse_ts_te (the DDRate / trend_rate reader, lib:196-229) on every filter path against what the
    REFERENCE's function returned - or died with - on the same files (tests/golden/parse_paths.npz, generated by
    make_golden.py --only parse): arrays bit for bit, PRESENT / ORIGIN, and the exception type where the reference
    raises (the TBP -last_year mask quirk: IndexError; an empty selection: ValueError)."""
    import warnings
    import pandas as pd
    from literate_amd import literate_library as ll
    P = np.load(os.path.join(golden_dir, "parse_paths.npz"))
    files = {}
    for name in ("example_TBP", "example_TAD", "metal_bands", "nan_TBP", "nan_TAD"):     # nan_*: every 7th death missing
        t = pd.DataFrame(P[name + "/table"], columns=[str(c) for c in P[name + "/header"]])
        for c in t.columns:
            if not t[c].isna().any():
                t[c] = t[c].astype(np.int64)             # the shipped files hold integers
        path = tmp_path / (name + ".tsv")
        t.to_csv(path, sep="\t", index=False, na_rep="")
        files[name] = str(path)
    n_ok = n_err = 0
    for name, tbp, fy, ly, jitter, tag in P["cases"]:
        want_err = str(P[tag + "/error"])
        args = (files[name], bool(int(tbp)), int(fy), int(ly), float(jitter))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if want_err:
                with pytest.raises(Exception) as ei:
                    ll.parse_ts_te(*args)
                assert type(ei.value).__name__ == want_err, (tag, type(ei.value).__name__, want_err)
                n_err += 1
                continue
            ts, te, present, origin = ll.parse_ts_te(*args)
        assert np.array_equal(np.asarray(ts, float), P[tag + "/ts"], equal_nan=True), tag
        assert np.array_equal(np.asarray(te, float), P[tag + "/te"], equal_nan=True), tag
        assert np.array_equal([float(present), float(origin)], P[tag + "/present_origin"], equal_nan=True), tag
        if name.startswith("nan_"):
            assert np.isnan(np.asarray(te, float)).sum() >= 3, tag         # a missing death stays missing (strict comparison)
        n_ok += 1
    assert n_ok >= 16 and n_err >= 8


def test_unit_window_routing_is_decided_on_the_host():
    """ops.bin_events sends the windows the reference always bins into - [t0 + w, t0 + w + 1] on an integer origin (LRF:519-523,
    lib create_bins:231-245) - to the one-pass kernel and everything else to lr_bin_events; the decision reads host arrays
    only (no device read-back, no GPU needed to make it)."""
    from literate_amd import ops
    lo = 7.0 + np.arange(24)
    assert ops._unit_windows(lo, lo + 1) == (7.0, 24)
    assert ops._unit_windows([3.0], [4.0]) == (3.0, 1)
    assert ops._unit_windows(list(lo), list(lo + 1)) == (7.0, 24)
    assert ops._unit_windows(-5.0 + np.arange(4), -4.0 + np.arange(4)) == (-5.0, 4)
    for bad_lo, bad_hi in ((lo + 0.5, lo + 1.5),                 # not on an integer origin
                           (lo, lo + 2),                         # two units wide
                           (lo[::-1], lo[::-1] + 1),             # descending
                           (np.r_[lo[:5], lo[6:]], np.r_[lo[:5], lo[6:]] + 1),   # a gap
                           ([0.0, 3.3], [1.0, 9.9]),             # arbitrary windows (precompute_events / get_BDlik segments)
                           ([np.nan], [1.0]), ([1e12], [1e12 + 1]), ([], [])):
        assert ops._unit_windows(bad_lo, bad_hi) is None
    assert ops._unit_windows(np.arange(5000.0), np.arange(5000.0) + 1) is None      # more windows than LR_MAX_BINS

    class FakeTensor:                                            # anything with a data_ptr is left alone
        def data_ptr(self):
            return 0
    assert ops._unit_windows(FakeTensor(), FakeTensor()) is None
