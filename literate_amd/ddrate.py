"""DDRate.py's sampler (DDRate.py:124-241) on the multi-chain engine: host mirror.

The chains run in `ChainEngine` with `dd=...` (lr_mcmc_config.sampler = 1): every proposal is scored by the
per-lineage scan on the DD per-bin rates, a trace row holds [it, posterior, likelihood, prior, args[8]].  The
per-bin log columns (l_i, m_i, niche_i, nicheFrac_i) and the two likelihood halves are functions of the logged
parameter vector alone and are recomputed from it at write time by lr_dd_rates / lr_binned_keiding.
"""
import csv

import numpy as np

from . import ops
from .engine import ChainEngine
from .literate_library import calculate_r_squared, create_bins

LOG_HEAD = ["it", "posterior", "likelihood", "likelihood_birth", "likelihood_death", "prior", "l_max", "steepness_k",
            "midpoint_x0", "initCarryingCap", "maxCarryingCap", "m_max", "nuB", "nuD"]          # DD:138-139


def model_suffix(m_birth, m_death):
    """DD:127-133."""
    out = {0: "_LL", 1: "_LDD", 2: "_LDDN"}[m_birth]
    return out + ("_ML" if m_death <= 0 else ("_MDD" if m_death == 1 else "_MDDN"))


class DDRateEngine(ChainEngine):
    def __init__(self, ts, te, origin, present, n_chains, m_birth=2, m_death=2, init_death=0.1, seed=1, s_freq=1000,
                 n_trace_slots=0, chain_offset=0, rm_first_bin=0, engine="auto", **kw):
        """origin, present: ORIGIN, PRESENT of parse_ts_te (lib:196-229); the statistics are create_bins' (DD:36)."""
        (self.origin, self.present, self.n_spec, self.n_exti, self.DT, n_time_bins,
         self.time_range) = create_bins(origin, present, ts, te, rm_first_bin)
        self.m_birth, self.m_death = int(m_birth), int(m_death)
        super().__init__(ts, te, n_chains, model=2, seed=seed, s_freq=s_freq, n_trace_slots=n_trace_slots,
                         chain_offset=chain_offset, stats=(self.origin, n_time_bins, self.DT), engine=engine,
                         dd=dict(m_birth=m_birth, m_death=m_death, present=self.present, init_death=init_death), **kw)

    def log_rows(self, chain, emp=None, n_samples=None):
        """The rows DDRate.py writes for one chain (DD:219-235): [it, posterior, likelihood, likelihood_birth,
        likelihood_death, prior, args as logged (x0 + ORIGIN, L + div_0), l_i.., m_i.., niche_i.., nicheFrac_i..
        (+ adequacy when emp=(B_EMP, D_EMP))]."""
        return self.log_rows_from(self.trace_rows(n_samples)[:, chain], emp)

    def log_rows_from(self, tr, emp=None):
        """The same from given trace rows [samples, LR_TRACE_W] of one chain (a window of a streamed run)."""
        return list(self.log_table_from(tr, emp))

    def log_table_from(self, tr, emp=None):
        """Trace rows [..., LR_TRACE_W] (any leading shape: a chain's rows, or a whole window [chains, samples]) -> the log
        rows [..., columns] as one float64 array: the per-bin columns of ALL rows from one lr_dd_rates / lr_binned_keiding
        launch per 64k rows, the adequacy columns as array operations (logs.adequacy_rows)."""
        from . import logs
        tr = np.asarray(tr, dtype=float)
        lead, R = tr.shape[:-1], tr.reshape(-1, tr.shape[-1])
        n = len(self.DT)
        out = np.empty((len(R), 14 + 4 * n + (3 if emp is not None else 0)))
        for a in range(0, len(R), 1 << 16):
            T = R[a:a + (1 << 16)]
            args = T[:, 4:12]
            b, d, ni, nf = [x.cpu().numpy() for x in ops.dd_rates(args, self.DT, self.m_birth, self.m_death)]
            lb, ld = [x.cpu().numpy() for x in ops.binned_keiding(b, d, self.n_spec, self.n_exti, self.DT)]
            O = out[a:a + (1 << 16)]
            O[:, 0], O[:, 1], O[:, 2], O[:, 3], O[:, 4], O[:, 5] = T[:, 0], T[:, 1], T[:, 2], lb, ld, T[:, 3]
            O[:, 6:14] = args
            O[:, 8] += self.origin           # DD:224
            O[:, 10] += O[:, 9]              # DD:225
            O[:, 14:14 + n], O[:, 14 + n:14 + 2 * n], O[:, 14 + 2 * n:14 + 3 * n], O[:, 14 + 3 * n:14 + 4 * n] = b, d, ni, nf
            if emp is not None:
                O[:, 14 + 4 * n:] = logs.adequacy_rows(emp[0], emp[1], b, d)
        return out.reshape(lead + (out.shape[1],))

    def log_head(self):
        n = len(self.DT)
        head = list(LOG_HEAD)
        for name in ("l_%s", "m_%s", "niche_%s", "nicheFrac_%s"):
            head += [name % i for i in range(n)]
        return head + ["corr_coeff", "rsquared", "gelman_r2"]

    def write_log(self, path, chain, emp=None, n_samples=None):
        self.start_log(path)
        self.append_log(path, self.trace_rows(n_samples)[:, chain], emp)

    def start_log(self, path):
        with open(path, "w") as f:
            csv.writer(f, delimiter='\t').writerow(self.log_head())

    def append_log(self, path, tr, emp=None):
        """Append the rows of one window and push them to disk (the reference flushes and fsyncs every sample,
        DD:236-238)."""
        self.append_logs([path], np.asarray(tr, dtype=float)[:, None, :], emp)

    def append_logs(self, paths, rows, emp=None):
        """A window of all local chains at once: rows [samples, chains, LR_TRACE_W] -> paths[c] (one device launch for the
        per-bin columns of the whole window, the numbers formatted natively on a few threads: literate_amd.logs)."""
        from . import logs
        rows = np.asarray(rows, dtype=float)
        if rows.shape[0] == 0:
            return
        logs.append_table_logs(paths, self.log_table_from(rows.transpose(1, 0, 2), emp))
