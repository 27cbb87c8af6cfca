"""Device posterior against the 16 reference chains (tests/golden/posterior_*.npz): z and relative difference per bin for a
given run length.  python scratch/posterior_check.py example_TBP 0 256 2000000 400"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from literate_amd.engine import ChainEngine, split_trace_row
from oracle import literate_oracle as lo
name, model, C, n_it, s = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "binning_lik.npz"))
R = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "posterior_%s_m%d.npz" % (name, model)))
n_ref = int(R["meta"][3])
eng = ChainEngine(G[name + "/ts"], G[name + "/te"], C, model=model, seed=int(os.environ.get("SEED", "77")), s_freq=s, n_trace_slots=n_it // s)
eng.init(); eng.steps(n_it)
tr = eng.trace_rows(); eng.close()
start, end = G[name + "/start_end"]
for burn_frac in (0.2, 0.5):
    sp, ex = [], []
    for c in range(C):
        rows = [split_trace_row(tr[i, c]) for i in range(tr.shape[0])]
        sp.append(lo.marginal_rates_from_rows([r[1] for r in rows], end, start, burnin=burn_frac)[0])
        ex.append(lo.marginal_rates_from_rows([r[2] for r in rows], end, start, burnin=burn_frac)[0])
    sp, ex = np.array(sp), np.array(ex)
    ref_sp = np.array([R["c%d/sp_mean" % c] for c in range(n_ref)]); ref_ex = np.array([R["c%d/ex_mean" % c] for c in range(n_ref)])
    for tag, mine, ref in (("sp", sp, ref_sp), ("ex", ex, ref_ex)):
        se = np.sqrt(mine.var(0, ddof=1) / len(mine) + ref.var(0, ddof=1) / len(ref))
        z = (mine.mean(0) - ref.mean(0)) / se
        rel = mine.mean(0) / ref.mean(0) - 1
        print("burn %.1f %s: max|z| %.2f  max|rel| %.4f   z[:8] %s rel[:8] %s  se_ref/se_mine %.2f" % (
            burn_frac, tag, np.max(np.abs(z)), np.max(np.abs(rel)), np.round(z[:8], 2), np.round(rel[:8], 4),
            np.sqrt((ref.var(0, ddof=1) / len(ref)).mean() / (mine.var(0, ddof=1) / len(mine)).mean())))
burn = tr.shape[0] // 5
names = ["likelihood", "prior", "lambda_avg", "mu_avg", "K_l", "K_m", "root", "death", "hp_BI", "hp_D", "poisson"]
ref_sc = np.array([R["c%d/scalars" % c] for c in range(n_ref)])     # lik, prior, lambda_avg, mu_avg, hpBI, hpD, poi, n
dev = tr[burn:, :, :13].mean(0)                                      # per chain means [C, 13]
for j, (col, rj) in enumerate([(2, 0), (3, 1), (4, 2), (5, 3), (10, 4), (11, 5), (12, 6)]):
    m, r = dev[:, col], ref_sc[:, rj]
    se = np.sqrt(m.var(ddof=1) / len(m) + r.var(ddof=1) / len(r))
    print("col %2d: device %.6f  reference %.6f  rel %.5f  z %.2f" % (col, m.mean(), r.mean(), m.mean() / r.mean() - 1, (m.mean() - r.mean()) / se))
kl = np.array([np.dot(R["c%d/K_l_hist" % c], np.arange(40)) / R["c%d/K_l_hist" % c].sum() for c in range(n_ref)])
km = np.array([np.dot(R["c%d/K_m_hist" % c], np.arange(40)) / R["c%d/K_m_hist" % c].sum() for c in range(n_ref)])
for nm, col, r in (("K_l", 6, kl), ("K_m", 7, km)):
    m = tr[burn:, :, col].mean(0)
    se = np.sqrt(m.var(ddof=1) / len(m) + r.var(ddof=1) / len(r))
    print("%s: device %.5f reference %.5f z %.2f" % (nm, m.mean(), r.mean(), (m.mean() - r.mean()) / se))
