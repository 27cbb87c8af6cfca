"""Distribution check of the device draws (lr_debug_draws: the engines' own Philox -> uniform / normal / gamma functions):
Kolmogorov-Smirnov against scipy's distributions, 2e6 draws per case, and the first two moments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import stats
from literate_amd import ops
n = 2_000_000
rng = np.random.default_rng(1)
it = np.arange(n, dtype=np.int64)
for kind, shape, name, dist in [(0, 1.0, "uniform a", stats.uniform()), (1, 1.0, "uniform b", stats.uniform()), (2, 1.0, "normal", stats.norm()),
                                (3, 2.0, "gamma 2", stats.gamma(2.0)), (3, 1.2, "gamma 1.2", stats.gamma(1.2)), (3, 3.2, "gamma 3.2", stats.gamma(3.2)),
                                (3, 10.0, "gamma 10", stats.gamma(10.0)), (3, 21.2, "gamma 21.2", stats.gamma(21.2)), (3, 0.7, "gamma 0.7", stats.gamma(0.7))]:
    x = ops.debug_draws(12345, 7, it, np.full(n, 3, np.int32), np.zeros(n, np.int32), np.full(n, kind, np.int32), np.full(n, shape)).cpu().numpy()
    ks = stats.kstest(x, dist.cdf)
    m, v = dist.stats("mv")
    print("%-10s KS D %.5f p %.3f   mean %.6f (%.6f, z %.2f)  var %.6f (%.6f)  min %.3g max %.3g" % (
        name, ks.statistic, ks.pvalue, x.mean(), m, (x.mean() - m) / np.sqrt(v / n), x.var(), v, x.min(), x.max()))
