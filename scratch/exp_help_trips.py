"""Four-chain kernel with helper waves: us per iteration against the helper waves' scan share (LR_P4_HELP_TRIPS) by input
size - the data behind lr_set_shares' rule.  Usage: python scratch/exp_help_trips.py N trips [trips ...] (one process per
setting: the share is read once per process)."""
import os, subprocess, sys
if len(sys.argv) > 2 and sys.argv[1] != "--one":
    N = int(sys.argv[1])
    for t in sys.argv[2:]:
        out = subprocess.run([sys.executable, __file__, "--one", str(N)], env=dict(os.environ, LR_P4_HELP_TRIPS=t),
                             capture_output=True, text=True).stdout.strip().splitlines()
        print("N=%8d help trips %3s: %s" % (N, t, out[-1] if out else "failed"), flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
N = int(sys.argv[2])
ts, te, _ = synth.make_lineages(N, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine="persistent4")
eng.init(); eng.steps(1500 if N <= 100000 else 300); torch.cuda.synchronize()
n = 1000 if N <= 100000 else 200
print("%.2f us/iter" % (eng.timed_steps(n) / n * 1e3))
