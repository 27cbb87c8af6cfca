"""Greedy search over the scanner-wave shares of the four-chain kernel (LR_P4_SHARES): moves one trip from one wave pair
to another while the measured time per iteration improves.  Each evaluation is a fresh process (the shares are read
when the engine is created)."""
import itertools, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=100, n_trace_slots=60, engine="persistent4")
eng.init(); eng.steps(1500); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.perf_counter(); eng.steps(1500); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 1500 * 1e6)
print("US %%.3f" %% best)
''' % ROOT

def measure(sh):
    env = dict(os.environ, LR_P4_SHARES=",".join(str(x) for x in sh))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120).stdout
    return float([l for l in out.splitlines() if l.startswith("US")][0].split()[1])

cur = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6,6,2,0,-2,-6,-6").split(",")]
best = measure(cur)
print("start", cur, best, flush=True)
for sweep in range(int(os.environ.get("SWEEPS", "2"))):
    improved = False
    for a, b in itertools.permutations(range(7), 2):
        cand = list(cur); cand[a] += 1; cand[b] -= 1
        t = measure(cand)
        print("  try", cand, "%.3f" % t, flush=True)
        if t < best - 0.03:
            best, cur, improved = t, cand, True
            print("  -> accept", cur, best, flush=True)
    if not improved:
        break
print("RESULT", ",".join(str(x) for x in cur), best)
