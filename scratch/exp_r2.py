"""Round-2 scoreboard: us per iteration of every BASELINE config + few-chain shards + general-times cfg4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
from literate_amd.ddrate import DDRateEngine

G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "binning_lik.npz"))
only = sys.argv[1:]


WARM = int(os.environ.get("LR_EXP_WARM", "300"))     # iterations before the timed ones (the chains start at K = 1)


def run(name, mk, n=2000, warm=None):
    if only and not any(o in name for o in only):
        return
    eng = mk()
    eng.init(); eng.steps(WARM if warm is None else warm); torch.cuda.synchronize()
    ms = eng.timed_steps(n)
    N, C = eng.ts.numel(), eng.n_chains
    snap = eng.snapshot()
    print('%-34s persistent=%d threads=%4d team=%d x %d: %7.2f us/iter  %.3e evals/s   (mean K_l %.1f K_m %.1f)' % (
        name, eng.layout.persistent, eng.layout.reserved1, eng.layout.team_blocks, eng.layout.spec_chains_per_team,
        ms / n * 1e3, n * N * C / (ms * 1e-3), snap["K_l"].mean(), snap["K_m"].mean()), flush=True)
    eng.close()


ts4, te4, _ = synth.make_lineages(100000, 128, 20, 0)
for C in (32, 64, 128, 256, 512, 1024):
    run("cfg4 100k x %d chains" % C, lambda: ChainEngine(ts4, te4, C, model=0, seed=1, s_freq=100, n_trace_slots=40))
rng = np.random.default_rng(5)
ts4g = ts4 + rng.uniform(0, 1, len(ts4)) * 0.999
te4g = np.maximum(te4 + rng.uniform(-0.49, 0.49, len(te4)), ts4g + 1e-3)
for C in (128, 1024):
    run("cfg4-general 100k x %d" % C, lambda: ChainEngine(ts4g, te4g, C, model=0, seed=1, s_freq=100, n_trace_slots=40))
run("cfg4 model3 100k x 1024", lambda: ChainEngine(ts4, te4, 1024, model=3, seed=1, s_freq=100, n_trace_slots=40))
ts3, te3, _ = synth.make_lineages(10000, 128, 20, 0)
run("cfg3 10k x 256", lambda: ChainEngine(ts3, te3, 256, model=0, seed=1, s_freq=100, n_trace_slots=80), n=4000)
run("cfg2 metal_bands x 128 (model 2)", lambda: ChainEngine(G["metal_bands/ts"], G["metal_bands/te"], 128, model=2, seed=1, s_freq=100, n_trace_slots=80), n=4000)
ts5, te5, _ = synth.make_lineages(50000, 64, 6, 0)
run("cfg5 DDRate 50k x 256", lambda: DDRateEngine(ts5, te5, float(ts5.min()), float(te5.max()), 256, m_birth=2, m_death=2, seed=1, s_freq=100, n_trace_slots=80), n=4000)
for eng_name in ("persistent4", "launch"):
    run("p4general cfg4-general 100k x 1024 %s" % eng_name, lambda: ChainEngine(ts4g, te4g, 1024, model=0, seed=1, s_freq=100, n_trace_slots=40, engine=eng_name))
    run("p4general cfg4-general 100k x 2048 %s" % eng_name, lambda: ChainEngine(ts4g, te4g, 2048, model=0, seed=1, s_freq=100, n_trace_slots=40, engine=eng_name))
