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
        tr = np.asarray(tr, dtype=float)
        if len(tr) == 0:
            return []
        args = tr[:, 4:12]
        b, d, ni, nf = [x.cpu().numpy() for x in ops.dd_rates(args, self.DT, self.m_birth, self.m_death)]
        lb, ld = [x.cpu().numpy() for x in ops.binned_keiding(b, d, self.n_spec, self.n_exti, self.DT)]
        logged = args.copy()
        logged[:, 2] += self.origin          # DD:224
        logged[:, 4] += logged[:, 3]         # DD:225
        rows = []
        for i in range(len(tr)):
            row = [tr[i, 0], tr[i, 1], tr[i, 2], lb[i], ld[i], tr[i, 3]] + list(logged[i]) + list(b[i]) + list(d[i]) \
                + list(ni[i]) + list(nf[i])
            if emp is not None:
                with np.errstate(all="ignore"):
                    row += list(calculate_r_squared(emp[0], emp[1], b[i], d[i]))
            rows.append(np.array(row, dtype=float))
        return rows

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
        import os
        rows = self.log_rows_from(tr, emp)
        if not rows:
            return
        with open(path, "a") as f:
            w = csv.writer(f, delimiter='\t')
            # (one tolist() per window: Python floats, which csv writes in their shortest round-trip form - per-element
            # numpy scalars cost several times the formatting itself)
            w.writerows([[int(r[0])] + r[1:] for r in np.asarray(rows, dtype=np.float64).tolist()])
            f.flush()
            os.fsync(f.fileno())
