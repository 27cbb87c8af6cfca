"""trend_rate.py's sampler (trend_rate.py:102-196) on the multi-chain engine: host mirror (see ddrate.py for the
scheme: the chains run in `ChainEngine` with lr_mcmc_config.sampler = 2, a trace row holds [it, posterior,
likelihood, prior, args[6]]; per-bin columns and likelihood halves are recomputed from the logged parameters)."""
import csv

import numpy as np

from . import ops
from .engine import ChainEngine
from .literate_library import calculate_r_squared, create_bins

SMALL_NUMBER = 0.000000000000001
LOG_HEAD = ["it", "posterior", "likelihood", "likelihood_birth", "likelihood_death", "prior", "l_min", "m_min", "alpha",
            "beta", "delta", "gamma"]                                                                   # trend_rate.py:113


def parse_trend_data(trend_file_path, index, rm_first_bin):
    """trend_rate.py:58-69: column `index` of a tab-separated file, last bin dropped, min-max scaled, zeros floored."""
    import pandas as pd
    trend = pd.read_csv(trend_file_path, sep='\t').iloc[:, index].to_numpy().astype(float)
    return normalise_trend(trend, rm_first_bin)


def normalise_trend(trend, rm_first_bin=0):
    trend = np.array(trend, dtype=float)[:-1]
    if rm_first_bin:
        trend = trend[1:]
    trend = (trend - np.min(trend)) / (np.max(trend) - np.min(trend))
    trend[trend == 0] = SMALL_NUMBER
    return trend


def model_suffix(const_birth, const_death, no_death=False):
    """trend_rate.py:103-108."""
    return ("_CONB" if const_birth else "_EXPB") + ("_ND" if no_death else ("_COND" if const_death else "_EXPD"))


class TrendRateEngine(ChainEngine):
    def __init__(self, ts, te, origin, present, trend, n_chains, const_birth=False, const_death=False, seed=1,
                 s_freq=1000, n_trace_slots=0, chain_offset=0, rm_first_bin=0, engine="auto", **kw):
        """trend: the normalised covariate, one value per time bin of create_bins (trend_rate.py:71)."""
        (self.origin, self.present, self.n_spec, self.n_exti, self.DT, n_time_bins,
         self.time_range) = create_bins(origin, present, ts, te, rm_first_bin)
        self.trend = np.asarray(trend, dtype=float)
        if len(self.trend) != n_time_bins:
            raise ValueError("trend has %d entries, the data %d time bins" % (len(self.trend), n_time_bins))
        self.const_birth, self.const_death = bool(const_birth), bool(const_death)
        super().__init__(ts, te, n_chains, model=2, seed=seed, s_freq=s_freq, n_trace_slots=n_trace_slots,
                         chain_offset=chain_offset, stats=(self.origin, n_time_bins, self.trend), engine=engine,
                         dd=dict(kind="trend", m_birth=int(self.const_birth), m_death=int(self.const_death)), **kw)

    def log_rows(self, chain, emp=None, n_samples=None):
        """Rows as trend_rate.py writes them (:183-189)."""
        return self.log_rows_from(self.trace_rows(n_samples)[:, chain], emp)

    def log_rows_from(self, tr, emp=None):
        """The same from given trace rows [samples, LR_TRACE_W] of one chain (a window of a streamed run)."""
        return list(self.log_table_from(tr, emp))

    def log_table_from(self, tr, emp=None):
        """Trace rows [..., LR_TRACE_W] (a chain's rows, or a whole window [chains, samples]) -> the log rows [..., columns]
        as one float64 array (one lr_trend_rates / lr_binned_keiding launch per 64k rows, adequacy as array operations)."""
        from . import logs
        tr = np.asarray(tr, dtype=float)
        lead, R = tr.shape[:-1], tr.reshape(-1, tr.shape[-1])
        n = len(self.DT)
        out = np.empty((len(R), 12 + 2 * n + (3 if emp is not None else 0)))
        for a in range(0, len(R), 1 << 16):
            T = R[a:a + (1 << 16)]
            args = T[:, 4:10]
            b, d = [x.cpu().numpy() for x in ops.trend_rates(args, self.trend, self.const_birth, self.const_death)]
            lb, ld = [x.cpu().numpy() for x in ops.binned_keiding(b, d, self.n_spec, self.n_exti, self.DT)]
            O = out[a:a + (1 << 16)]
            O[:, 0], O[:, 1], O[:, 2], O[:, 3], O[:, 4], O[:, 5] = T[:, 0], T[:, 1], T[:, 2], lb, ld, T[:, 3]
            O[:, 6:12] = args
            O[:, 12:12 + n], O[:, 12 + n:12 + 2 * n] = b, d
            if emp is not None:
                O[:, 12 + 2 * n:] = logs.adequacy_rows(emp[0], emp[1], b, d)
        return out.reshape(lead + (out.shape[1],))

    def log_head(self):
        n = len(self.DT)
        return list(LOG_HEAD) + ["l_%s" % i for i in range(n)] + ["m_%s" % i for i in range(n)] + \
            ["corr_coeff", "rsquared", "gelman_r2"]

    def write_log(self, path, chain, emp=None, n_samples=None):
        self.start_log(path)
        self.append_log(path, self.trace_rows(n_samples)[:, chain], emp)

    def start_log(self, path):
        with open(path, "w") as f:
            csv.writer(f, delimiter='\t').writerow(self.log_head())

    def append_log(self, path, tr, emp=None):
        """Append the rows of one window and push them to disk (the reference flushes and fsyncs every sample,
        trend_rate.py:190-195)."""
        self.append_logs([path], np.asarray(tr, dtype=float)[:, None, :], emp)

    def append_logs(self, paths, rows, emp=None):
        """A window of all local chains at once: rows [samples, chains, LR_TRACE_W] -> paths[c]."""
        from . import logs
        rows = np.asarray(rows, dtype=float)
        if rows.shape[0] == 0:
            return
        logs.append_table_logs(paths, self.log_table_from(rows.transpose(1, 0, 2), emp))
