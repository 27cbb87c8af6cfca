"""Host overhead of a short timed region (bench.py at --steps 20): wall clock around timed_steps() against the device time
the events report, plus the floor: the same launch with no events at all, and an empty torch kernel.
(Measured round 3, cfg4, K = 20: markers around the launch 169.7 us wall / 150.3 us device = 1.13; no events 163.2 us;
empty kernel + sync 17.8 us.  Events attached to the dispatch itself - hipExtLaunchKernel start / stop events - were
WORSE: 184.4 us wall, the stop event makes the dispatch release at system scope.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts, te, _ = synth.make_lineages(100_000, n_bins=128, n_shifts=20, seed=0)
eng = ChainEngine(ts, te, 1024, model=0, seed=2026, s_freq=100, n_trace_slots=4000)
eng.init()
t = time.perf_counter()
while time.perf_counter() - t < 0.4:
    eng.steps(256); torch.cuda.synchronize()
wall, dev, plain, empty = [], [], [], []
x = torch.zeros(64, device="cuda")
for rep in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ms = eng.timed_steps(K); torch.cuda.synchronize(); t1 = time.perf_counter()
    wall.append((t1 - t0) * 1e6); dev.append(ms * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.steps(K); torch.cuda.synchronize(); t1 = time.perf_counter()
    plain.append((t1 - t0) * 1e6)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); x.add_(1.0); torch.cuda.synchronize(); t1 = time.perf_counter()
    empty.append((t1 - t0) * 1e6)
med = lambda v: float(np.median(v))
print("K=%d: wall %.1f us (min %.1f)  device %.1f us  ratio %.3f | no events: wall %.1f us (min %.1f) | empty torch kernel + sync %.1f us (min %.1f)" % (
    K, med(wall), min(wall), med(dev), med(wall) / med(dev), med(plain), min(plain), med(empty), min(empty)))
