"""Where the CLI's wall clock goes: the window loop of LiteRateForward.py (steps -> mark -> collect -> append) with a timer
on every phase (example_TBP, 128 chains, windows of 1 M iterations, a sample every 1000)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import logs
from literate_amd.engine import ChainEngine, TraceStreamer
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "binning_lik.npz"))
ts, te = G["example_TBP/ts"], G["example_TBP/te"]
C, n_win, block, s = 128, 6, 1_000_000, 1000
eng = ChainEngine(ts, te, C, model=0, seed=1, s_freq=s, n_trace_slots=n_win * block // s)
eng.init()
tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "literate_mcmc_logs"))
writer = logs.ChainLogWriter(os.path.join(tmp, "x.tsv"), 0, "", C)
st = TraceStreamer(eng, C, C)
T = {}
def tick(k, t0): T[k] = T.get(k, 0.0) + time.perf_counter() - t0
t_all = time.perf_counter()
for w in range(n_win):
    t0 = time.perf_counter(); eng.steps(block); tick("steps() call", t0)
    t0 = time.perf_counter(); st.mark(); tick("mark", t0)
    if len(st.pending) > 1:
        t0 = time.perf_counter(); rows, snap, win = st.collect(); tick("collect", t0)
        t0 = time.perf_counter(); writer.append(rows); tick("append", t0)
while st.pending:
    t0 = time.perf_counter(); rows, snap, win = st.collect(); tick("collect", t0)
    t0 = time.perf_counter(); writer.append(rows); tick("append", t0)
t0 = time.perf_counter(); torch.cuda.synchronize(); tick("final sync", t0)
print("total %.2f s for %d windows" % (time.perf_counter() - t_all, n_win), {k: round(v, 2) for k, v in T.items()})
