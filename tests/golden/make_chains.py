#!/usr/bin/env python3
"""Long runs of the REFERENCE CLI (unmodified, separate processes) -> posterior summaries.

Build container only.  For each (dataset, model) runs several seeds of
/root/reference/LiteRateForward.py, then stores only KB-sized summaries in
tests/golden/posterior_<name>.npz:

  per-chain per-bin marginal birth/death rates (definition: plotRJforward.v3.py:92-139,
  restated in oracle.literate_oracle.marginal_rates_from_rows), K_l / K_m histograms,
  mean log-likelihood, and the hyper-parameter means.

Usage: python tests/golden/make_chains.py example_TBP 0 2000000 200 4
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import literate_oracle as lo  # noqa: E402
from make_golden import DATASETS, REF, parse_logs  # noqa: E402


def main():
    name, model, n, s, n_chains = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rel, flags = DATASETS[name]
    suffix = {0: "_BD", 1: "_ID", 2: "_BDk", 3: "_BDd"}[model]
    work = tempfile.mkdtemp(prefix="lr_chains_")
    procs = []
    for c in range(n_chains):
        d = os.path.join(work, "c%d" % c)
        os.mkdir(d)
        dst = os.path.join(d, os.path.basename(rel))
        shutil.copy(os.path.join(REF, rel), dst)
        cmd = [sys.executable, "-B", os.path.join(REF, "LiteRateForward.py"), "-d", dst, "-n", str(n), "-s", str(s),
               "-p", str(10**9), "-seed", str(1000 + c), "-model_BDI", str(model), "-calc_adequacy", "0"] + flags
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                      env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1")))
    for p in procs:
        p.wait()
    stem = os.path.splitext(os.path.basename(rel))[0] + suffix
    out = {}
    for c in range(n_chains):
        mc, sp, ex = parse_logs(os.path.join(work, "c%d" % c, "literate_mcmc_logs"), stem)
        start, end = mc[0, 8], mc[0, 9]
        burn = int(0.2 * len(mc))
        m_sp = lo.marginal_rates_from_rows(sp, end, start)[3]
        m_ex = lo.marginal_rates_from_rows(ex, end, start)[3]
        out["c%d/sp_mean" % c] = m_sp.mean(axis=0)
        out["c%d/ex_mean" % c] = m_ex.mean(axis=0)
        out["c%d/sp_var" % c] = m_sp.var(axis=0)
        out["c%d/ex_var" % c] = m_ex.var(axis=0)
        out["c%d/K_l_hist" % c] = np.bincount(mc[burn:, 6].astype(int), minlength=40)[:40]
        out["c%d/K_m_hist" % c] = np.bincount(mc[burn:, 7].astype(int), minlength=40)[:40]
        out["c%d/scalars" % c] = np.array([mc[burn:, 2].mean(), mc[burn:, 3].mean(), mc[burn:, 4].mean(),
                                           mc[burn:, 5].mean(), mc[burn:, 10].mean(), mc[burn:, 11].mean(),
                                           mc[burn:, 12].mean(), len(mc) - burn])
    out["meta"] = np.array([model, n, s, n_chains, start, end], dtype=float)
    np.savez_compressed(os.path.join(HERE, "posterior_%s_m%d.npz" % (name, model)), **out)
    shutil.rmtree(work, ignore_errors=True)
    print("posterior_%s_m%d.npz written" % (name, model))


if __name__ == "__main__":
    main()
