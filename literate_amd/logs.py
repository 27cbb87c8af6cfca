"""Reference-format log writers and the posterior marginal-rate summary (host side).

File names, headers and column order follow LiteRateForward.py:485-512, 321-359, 558-564 so
that plotRJforward.v3.py-style consumers read them unchanged; `marginal_rates` restates
plotRJforward.v3.py:92-139, the definition of "posterior rate marginals"."""
import csv
import re
import os

import numpy as np

from ._hip import LR_KMAX, LR_TRACE_HEAD

MCMC_HEAD = ["it", "posterior", "likelihood", "prior", "lambda_avg", "mu_avg", "K_l", "K_m", "root_age", "death_age",
             "gamma_rate_hp_BI", "gamma_rate_hp_D", "poisson_rate_hp"]
ADEQUACY_HEAD = ["corr_coeff", "rsquared", "gelman_r2"]
SUFFIX = {0: "_BD", 1: "_ID", 2: "_BDk", 3: "_BDd"}       # LRF:422-428


def split_row(row):
    """trace row -> (head[13], sp_rates row, ex_rates row): rates then interior shift times."""
    head = row[:LR_TRACE_HEAD]
    kl, km = int(head[6]), int(head[7])
    rl = row[LR_TRACE_HEAD:LR_TRACE_HEAD + 2 * LR_KMAX - 1]
    rm = row[LR_TRACE_HEAD + 2 * LR_KMAX - 1:]
    return head, np.concatenate([rl[:kl], rl[LR_KMAX:LR_KMAX + kl - 1]]), np.concatenate([rm[:km], rm[LR_KMAX:LR_KMAX + km - 1]])


def rates_per_bin(rates, shifts, start_time, n_bins):
    """L_acc[indLA] from a logged row: floor-to-bin index of get_rate_index (LRF:125-135, 262)."""
    edges = np.floor(np.asarray(shifts, dtype=float)) - np.floor(start_time)
    seg = np.searchsorted(edges, np.arange(n_bins), side="right")
    return np.asarray(rates, dtype=float)[seg]


def adequacy(emp_birth, emp_death, est_birth, est_death):
    """calculate_r_squared (lib:268-279)."""
    x = np.concatenate([emp_birth, emp_death])
    y = np.concatenate([est_birth, est_death])
    coeff = np.sum(x * y) / np.sum(x * x)
    fitted = coeff * x
    resid = y - fitted
    r2 = 1 - np.sum(resid ** 2) / np.sum(y ** 2)
    vf = np.var(fitted, ddof=1)
    return coeff, r2, vf / (vf + np.var(resid, ddof=1))


def log_paths(data_file, model, out="", chain=None):
    out_dir = os.path.dirname(data_file) or os.getcwd()
    out_dir = "%s/literate_mcmc_logs" % out_dir
    stem = os.path.splitext(os.path.basename(data_file))[0] + SUFFIX[model] + out
    if chain is not None:
        stem += "_c%d" % chain
    return out_dir, {k: "%s/%s_%s.log" % (out_dir, stem, k) for k in ("mcmc", "sp_rates", "ex_rates", "div")}


def write_div_log(path, sp_events, ex_events, br_length):
    with open(path, "w") as f:
        f.write('sp_events\tex_events\tbr_length\n')
        w = csv.writer(f, delimiter='\t')
        for row in zip(sp_events.tolist(), ex_events.tolist(), br_length.tolist()):
            w.writerow(row)


def _compact_rows(rows):
    """[..., LR_TRACE_W] trace rows -> (array holding only the columns in use, kl_max, km_max):
    head[13] | birth rates[kl_max] | birth shift times[kl_max - 1] | death rates[km_max] | death shift times[km_max - 1].
    (A row reserves 2 x 63 slots; the runs hold a handful of rates: converting the padding would cost more than the data.)"""
    R = np.asarray(rows, dtype=np.float64)
    H, K = LR_TRACE_HEAD, LR_KMAX
    M0 = H + 2 * K - 1                       # first death-process slot of a row
    if R.size == 0:
        return R[..., :H], 1, 1
    kl, km = int(R[..., 6].max()), int(R[..., 7].max())
    parts = [R[..., :H], R[..., H:H + kl], R[..., H + K:H + K + kl - 1], R[..., M0:M0 + km], R[..., M0 + K:M0 + K + km - 1]]
    return np.concatenate(parts, axis=-1), kl, km


def _rates_per_bin_rows(rates, shifts, K, start, n_bins):
    """rates_per_bin for many rows at once: rates [R, kmax], shifts [R, kmax - 1] (row r holds K[r] rates), start [R]
    -> [R, n_bins], or None when a row's shift times do not ascend (searchsorted's answer is then its own)."""
    k = np.arange(shifts.shape[1])
    valid = k[None, :] < (K[:, None] - 1)
    edges = np.where(valid, np.floor(shifts) - np.floor(start)[:, None], np.inf)
    if edges.shape[1] > 1 and not np.all(edges[:, 1:] >= edges[:, :-1]):
        return None
    seg = (edges[:, :, None] <= np.arange(n_bins, dtype=np.float64)[None, None, :]).sum(axis=1)     # = searchsorted(side="right")
    return np.take_along_axis(rates, seg, axis=1)


def _r_squared_block(x, y):
    """calculate_r_squared (lib:268-279) of every row of y [rows, 2 n_bins] against x [2 n_bins] -> (coeff, r2, gelman)."""
    with np.errstate(all="ignore"):
        coeff = np.sum(x * y, axis=1) / np.sum(x * x)
        fitted = coeff[:, None] * x
        resid = y - fitted
        r2 = 1 - np.sum(resid ** 2, axis=1) / np.sum(y ** 2, axis=1)
        vf = np.var(fitted, axis=1, ddof=1)
        return coeff, r2, vf / (vf + np.var(resid, axis=1, ddof=1))


def _in_row_blocks(n_rows, n_cols, block):
    """block(a, b) -> bool over the row ranges [a, b) of a table of n_rows x n_cols doubles, ~4 MB of it at a time (the
    temporaries stay in cache), on a few threads when there are several blocks: they are independent and numpy's loops
    release the interpreter lock (a thousand chains x 128 bins cost 11 us a row on one core - more than the device needs
    for a sample's 1000 iterations).  -> all(block results)."""
    step = max(1, (1 << 19) // max(1, n_cols))
    spans = [(a, min(a + step, n_rows)) for a in range(0, n_rows, step)]
    if len(spans) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, len(spans), (os.cpu_count() or 2) // 2))) as pool:
            return all(list(pool.map(lambda ab: block(*ab), spans)))
    return all([block(a, b) for a, b in spans])


def adequacy_rows(emp_birth, emp_death, est_birth, est_death):
    """`adequacy` for many rows at once: est_birth, est_death [rows, n_bins] -> [rows, 3], the very numbers the per-row
    function gives (reductions run along the contiguous axis: the same summation per row)."""
    x = np.concatenate([emp_birth, emp_death])
    est_birth, est_death = np.asarray(est_birth, dtype=np.float64), np.asarray(est_death, dtype=np.float64)
    out = np.empty((est_birth.shape[0], 3))

    def block(a, b):
        y = np.ascontiguousarray(np.concatenate([est_birth[a:b], est_death[a:b]], axis=1))
        out[a:b, 0], out[a:b, 1], out[a:b, 2] = _r_squared_block(x, y)
        return True

    _in_row_blocks(out.shape[0], x.size, block)
    return out


def _adequacy_rows(compact, kl_max, km_max, emp, n_bins):
    """The three adequacy columns (calculate_r_squared, lib:268-279) of many compact rows at once: [R, 3], the very
    numbers `adequacy` gives row by row - a call per row costs 26 us of small-array numpy, more than the device needs for
    the 1000 iterations between two samples.  None when a row's shift times do not ascend (rates_per_bin then decides)."""
    A = compact.reshape(-1, compact.shape[-1])
    H = LR_TRACE_HEAD
    o_st, o_er = H + kl_max, H + 2 * kl_max - 1
    o_et = o_er + km_max
    x = np.concatenate([emp[0], emp[1]])
    out = np.empty((A.shape[0], 3))

    def block(a, b):
        B = A[a:b]
        KL, KM, start = B[:, 6].astype(np.int64), B[:, 7].astype(np.int64), B[:, 8]
        lam = _rates_per_bin_rows(B[:, H:H + kl_max], B[:, o_st:o_st + kl_max - 1], KL, start, n_bins)
        mu = _rates_per_bin_rows(B[:, o_er:o_er + km_max], B[:, o_et:o_et + km_max - 1], KM, start, n_bins)
        if lam is None or mu is None:
            return False
        out[a:b, 0], out[a:b, 1], out[a:b, 2] = _r_squared_block(x, np.ascontiguousarray(np.concatenate([lam, mu], axis=1)))
        return True

    ok = _in_row_blocks(A.shape[0], 2 * n_bins, block)
    return out.reshape(compact.shape[:-1] + (3,)) if ok else None


def _lines(compact, kl_max, km_max, emp, n_bins, pyrate_output, true_root_age, adeq=None):
    """compact rows of ONE chain (_compact_rows(...).tolist()) -> three lists of text lines (mcmc, sp_rates, ex_rates) in
    the reference's format (LRF:321-359).  Numbers are Python floats formatted with `str`: the shortest round-trip form
    the reference's `csv` writer produces.  adeq: the rows' adequacy columns (_adequacy_rows(...).tolist()); without
    them they are computed row by row."""
    lm, ls, le = [], [], []
    H = LR_TRACE_HEAD
    o_st, o_er = H + kl_max, H + 2 * kl_max - 1
    o_et = o_er + km_max
    root = float(true_root_age)
    for i, r in enumerate(compact):
        kl, km = int(r[6]), int(r[7])
        start, end = r[8], r[9]
        vals = [str(int(r[0])), str(r[1]), str(r[2]), str(r[3]), str(r[4]), str(r[5]), str(kl), str(km)]
        if pyrate_output:
            vals += [str(root), str(root - end)]
        else:
            vals += [str(start), str(end)]
        vals += [str(r[10]), str(r[11]), str(r[12])]
        sp_r, sp_t = r[H:H + kl], r[o_st:o_st + kl - 1]
        ex_r, ex_t = r[o_er:o_er + km], r[o_et:o_et + km - 1]
        if emp is not None:
            if adeq is not None:
                a = adeq[i]
                vals += [str(a[0]), str(a[1]), str(a[2])]
            else:
                lam = rates_per_bin(sp_r, sp_t, start, n_bins)
                mu = rates_per_bin(ex_r, ex_t, start, n_bins)
                with np.errstate(all="ignore"):
                    vals += [str(float(v)) for v in adequacy(emp[0], emp[1], lam, mu)]
        lm.append('\t'.join(vals) + '\n')
        if pyrate_output:
            sp_t = [root - t for t in sp_t]
            ex_t = [root - t for t in ex_t]
        ls.append('\t'.join(map(str, sp_r + sp_t)) + '\n')
        le.append('\t'.join(map(str, ex_r + ex_t)) + '\n')
    return lm, ls, le


def _chain_lines(rows, emp, n_bins, pyrate_output, true_root_age):
    """rows [samples, LR_TRACE_W] of ONE chain -> the three lists of text lines."""
    compact, kl, km = _compact_rows(rows)
    adeq = _adequacy_rows(compact, kl, km, emp, n_bins) if (emp is not None and compact.size) else None
    return _lines(compact.tolist(), kl, km, emp, n_bins, pyrate_output, true_root_age,
                  None if adeq is None else adeq.tolist())


class ChainLogWriter:
    """The three per-chain logs of one run, written as the run goes: the reference opens them once (LRF:485-512) and
    writes + flushes a row per sample (LRF:334-359); here every window of samples (TraceStreamer) is appended and
    flushed as soon as it has left the device, so a killed run leaves complete, parseable logs up to its last window.
    Files are opened per append (a thousand chains x three files would not fit a process's descriptor limit)."""

    def __init__(self, data_file, model, out, n_chains, emp=None, n_bins=None, pyrate_output=False, true_root_age=0.0):
        self.emp, self.n_bins, self.pyrate, self.root = emp, n_bins, pyrate_output, true_root_age
        self.paths = [log_paths(data_file, model, out, None if n_chains == 1 else c)[1] for c in range(n_chains)]
        head = '\t'.join(MCMC_HEAD + (ADEQUACY_HEAD if emp is not None else [])) + '\n'
        for p in self.paths:
            with open(p["mcmc"], "w") as f:
                f.write(head)
            open(p["sp_rates"], "w").close()
            open(p["ex_rates"], "w").close()

    def append(self, rows):
        """rows: [samples, chains, LR_TRACE_W] of one window."""
        if rows is None or len(rows) == 0:
            return
        # the whole window at once (only the columns in use), chain-major
        compact, kl, km = _compact_rows(np.asarray(rows).transpose(1, 0, 2))
        adeq = _adequacy_rows(compact, kl, km, self.emp, self.n_bins) if self.emp is not None else None
        if _native() is not None and (self.emp is None or adeq is not None):
            return self._append_native(compact, kl, km, adeq)
        compact, adeq = compact.tolist(), (None if adeq is None else adeq.tolist())
        for c, p in enumerate(self.paths):
            lm, ls, le = _lines(compact[c], kl, km, self.emp, self.n_bins, self.pyrate, self.root,
                                None if adeq is None else adeq[c])
            for key, lines in (("mcmc", lm), ("sp_rates", ls), ("ex_rates", le)):
                with open(p[key], "a") as f:
                    f.writelines(lines)
                    f.flush()

    def _append_native(self, A, kl_max, km_max, adeq):
        """The same bytes through lr_format_rows (csrc/lr_format.hip: Python's str(float) in C++, ~60 ns a number where
        CPython needs ~400): A = compact rows [chains, samples, columns]."""
        H = LR_TRACE_HEAD
        o_st, o_er = H + kl_max, H + 2 * kl_max - 1
        o_et = o_er + km_max
        root = float(self.root)
        S = A.shape[1]
        KL, KM = A[..., 6].astype(np.int64), A[..., 7].astype(np.int64)
        head = A[..., :H]
        if self.pyrate:
            head = head.copy()
            head[..., 8], head[..., 9] = root, root - A[..., 9]
        if adeq is not None:
            head = np.concatenate([head, adeq], axis=-1)
        head = np.ascontiguousarray(head)
        rs_head = np.arange(S + 1, dtype=np.int64) * head.shape[-1]

        C_ = A.shape[0]

        def ragged(o_r, o_t, kmax, K):
            """-> (the rates and shift times in use of ALL rows, flat in (chain, sample) order; where row i starts)"""
            j = np.arange(kmax)
            T = A[..., o_t:o_t + kmax - 1]
            vals = np.concatenate([A[..., o_r:o_r + kmax], root - T if self.pyrate else T], axis=-1)
            mask = np.concatenate([j < K[..., None], j[:kmax - 1] < (K[..., None] - 1)], axis=-1)
            starts = np.zeros(C_ * S + 1, dtype=np.int64)
            np.cumsum((2 * K - 1).reshape(-1), out=starts[1:])
            return np.ascontiguousarray(vals[mask]), starts

        (sp_v, sp_rs), (ex_v, ex_rs) = ragged(H, o_st, kl_max, KL), ragged(o_er, o_et, km_max, KM)
        head_v = head.reshape(-1)
        head_rs = np.arange(C_ * S + 1, dtype=np.int64) * head.shape[-1]

        def one(c):
            # (the window's numbers are formatted in place: a chain's rows are a range of the flat arrays)
            p = self.paths[c]
            for key, vals, starts, int_cols in (("mcmc", head_v, head_rs, _MCMC_INT_COLS), ("sp_rates", sp_v, sp_rs, 0),
                                                ("ex_rates", ex_v, ex_rs, 0)):
                with open(p[key], "ab") as f:
                    f.write(_native_format(vals, starts, int_cols, first_row=c * S, n_rows=S))
                    f.flush()

        # (one after the other: with ~100 us of formatting per call, threads spend more time handing the interpreter lock
        # around than they save - measured 7.0 against 2.0 us per row at 1024 chains)
        for c in range(len(self.paths)):
            one(c)


def append_table_logs(paths, tables):
    """paths[c] <- the rows of tables[c] ([rows, columns] float64; column 0 an integer), appended as csv.writer(delimiter
    '\\t') writes them (DD:236-238, trend_rate.py:183-195: '\\r\\n' line ends), flushed and fsynced per file; the
    chains on a few threads (formatting and writing both release the interpreter lock)."""
    def one(c):
        M = np.ascontiguousarray(tables[c], dtype=np.float64)
        if M.shape[0] == 0:
            return
        if _native() is not None:
            with open(paths[c], "ab") as f:
                f.write(_native_format(M.reshape(-1), np.arange(M.shape[0] + 1, dtype=np.int64) * M.shape[1], 1, crlf=True))
                f.flush()
                os.fsync(f.fileno())
        else:
            with open(paths[c], "a", newline="") as f:
                csv.writer(f, delimiter='\t').writerows([[int(r[0])] + r[1:] for r in M.tolist()])
                f.flush()
                os.fsync(f.fileno())

    if len(paths) > 1 and _native() is not None:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, len(paths), (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(one, range(len(paths))))
    else:
        for c in range(len(paths)):
            one(c)


_MCMC_INT_COLS = (1 << 0) | (1 << 6) | (1 << 7)        # it, K_l, K_m are written as integers (LRF:321-323)
_NATIVE = []


def _native():
    """libliterate_hip.so's lr_format_rows, or None (LR_LOG_FORMAT=python, or no library: the Python formatter writes the
    same bytes, tests/test_host_cpu.py)."""
    if not _NATIVE:
        lib = None
        if os.environ.get("LR_LOG_FORMAT", "native") != "python":
            try:
                from . import _hip
                lib = _hip.load()
                lib.lr_format_rows
            except Exception:
                lib = None
        _NATIVE.append(lib)
    return _NATIVE[0]


def _native_format(vals, row_start, int_cols=0, crlf=False, first_row=0, n_rows=None):
    """Tab-separated lines of `vals` (row i = vals[row_start[i]:row_start[i + 1]]) in the reference's csv form -> bytes
    (crlf: csv.writer's default line end; first_row, n_rows: a range of the rows).  The call releases the interpreter lock:
    chains can be formatted on threads."""
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    row_start = np.ascontiguousarray(row_start, dtype=np.int64)
    if n_rows is None:
        n_rows = int(row_start.size) - 1 - first_row
    n_vals = int(row_start[first_row + n_rows] - row_start[first_row]) if n_rows > 0 else 0
    cap = 26 * n_vals + 2 * n_rows + 2
    out = np.empty(cap, dtype=np.uint8)
    n = _native().lr_format_rows(vals.ctypes.data, row_start.ctypes.data + 8 * first_row, n_rows, int(int_cols), 1 if crlf else 0,
                                 out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("lr_format_rows: %d" % n)
    return out[:n].tobytes()


def write_chain_logs(paths, rows, emp=None, n_bins=None, pyrate_output=False, true_root_age=0.0):
    """rows: [samples, LR_TRACE_W] of ONE chain.  emp=(B_EMP, D_EMP) adds the adequacy columns
    (-calc_adequacy 1, LRF:327-329); pyrate_output flips times to root_age - t (LRF:324-341)."""
    lm, ls, le = _chain_lines(rows, emp, n_bins, pyrate_output, true_root_age)
    with open(paths["mcmc"], "w") as fm, open(paths["sp_rates"], "w") as fs, open(paths["ex_rates"], "w") as fe:
        fm.write('\t'.join(MCMC_HEAD + (ADEQUACY_HEAD if emp is not None else [])) + '\n')
        fm.writelines(lm), fs.writelines(ls), fe.writelines(le)


def combine_logs(mcmc_files, wd, burnin_pct):
    """plotRJforward.v3.py:307-350: pool the per-chain logs of one analysis into COMBINED_{mcmc,sp_rates,ex_rates,
    div}.log (burn-in dropped per file, the `it` column renumbered, the div statistics averaged over the files)."""
    mcmc_files = list(mcmc_files)
    total, header = [], ""
    for name in mcmc_files:
        with open(name) as f:
            lines = f.readlines()
        header = lines[0]
        total += lines[int(burnin_pct * len(lines[1:])) + 1:]
    with open(wd + '/COMBINED_mcmc.log', 'w') as o:
        o.write(header)
        for i, l in enumerate(total):
            l = l.split('\t')
            l[0] = str(i)
            o.write('\t'.join(l))
    for kind in ("sp_rates", "ex_rates"):
        total = []
        for name in mcmc_files:
            with open(name.replace('mcmc.log', kind + '.log')) as f:
                lines = f.readlines()
            total += lines[int(burnin_pct * len(lines)):]
        with open(wd + '/COMBINED_%s.log' % kind, 'w') as o:
            o.writelines(total)
    divs = []
    for name in mcmc_files:
        div_name = name.replace('mcmc.log', 'div.log')
        if not os.path.exists(div_name):              # one div log per data file: all chains of a run share it
            div_name = re.sub(r'_c\d+_div\.log$', '_div.log', div_name)
        divs.append(np.loadtxt(div_name, skiprows=1, ndmin=2))
    mean = np.mean(np.array(divs), axis=0)
    with open(wd + '/COMBINED_div.log', 'w') as o:
        o.write('sp_events\tex_events\tbr_length\n')
        for row in mean:
            o.write('\t'.join(str(float(v)) for v in row) + '\n')


def calcHPD(data, level=0.95):
    d = np.sort(np.asarray(data, dtype=float))
    n_in = int(round(level * len(d)))
    if n_in < 2:
        raise RuntimeError("not enough data")
    i = int(np.argmin(d[n_in - 1:] - d[:len(d) - n_in + 1]))
    return np.array([d[i], d[i + n_in - 1]])


def marginal_rates(rows, start_age, end_age, burnin=0.2):
    """plotRJforward.v3.py:92-139.  rows: sp_rates / ex_rates rows (rates then shift times).
    Returns (time_frames, mean, hpd_lo, hpd_hi, matrix[samples, nbins]); bins most recent first."""
    nbins = abs(int(end_age - start_age))
    edges = np.arange(end_age, start_age)
    if burnin < 1:
        burnin = min(int(burnin * len(rows)), int(0.9 * len(rows)))
    mat = []
    for row in rows[burnin:]:
        row = np.asarray(row, dtype=float)
        if len(row) == 1:
            mat.append(np.zeros(nbins) + row[0])
            continue
        nr = int(np.ceil(len(row) / 2.))
        h = np.histogram(row[nr:], bins=edges)[0]
        mat.append(row[:nr][np.cumsum(h)][::-1])
    mat = np.array(mat)
    hpd = np.array([calcHPD(mat[:, i], 0.95) for i in range(mat.shape[1])])
    frames = (edges - abs(edges[1] - edges[0]) / 2.)[1:]
    return frames, mat.mean(axis=0), hpd[:, 0], hpd[:, 1], mat
