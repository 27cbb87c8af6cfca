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


def _chain_lines(rows, emp, n_bins, pyrate_output, true_root_age):
    """rows of ONE chain -> three lists of text lines (mcmc, sp_rates, ex_rates) in the reference's format (LRF:321-359)."""
    lm, ls, le = [], [], []
    for row in rows:
        head, sp, ex = split_row(row)
        kl, km = int(head[6]), int(head[7])
        start, end = head[8], head[9]
        vals = [str(int(head[0]))] + [str(float(v)) for v in head[1:6]] + [str(kl), str(km)]
        if pyrate_output:
            vals += [str(float(true_root_age)), str(float(true_root_age - end))]
        else:
            vals += [str(float(start)), str(float(end))]
        vals += [str(float(v)) for v in head[10:13]]
        if emp is not None:
            lam = rates_per_bin(sp[:kl], sp[kl:], start, n_bins)
            mu = rates_per_bin(ex[:km], ex[km:], start, n_bins)
            with np.errstate(all="ignore"):
                vals += [str(float(v)) for v in adequacy(emp[0], emp[1], lam, mu)]
        lm.append('\t'.join(vals) + '\n')
        if pyrate_output:
            sp = np.concatenate([sp[:kl], true_root_age - sp[kl:]])
            ex = np.concatenate([ex[:km], true_root_age - ex[km:]])
        ls.append('\t'.join(str(float(v)) for v in sp) + '\n')
        le.append('\t'.join(str(float(v)) for v in ex) + '\n')
    return lm, ls, le


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
        for c, p in enumerate(self.paths):
            lm, ls, le = _chain_lines(rows[:, c], self.emp, self.n_bins, self.pyrate, self.root)
            for key, lines in (("mcmc", lm), ("sp_rates", ls), ("ex_rates", le)):
                with open(p[key], "a") as f:
                    f.writelines(lines)
                    f.flush()


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
