import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
def run(P, total=1024, n=640):
    engs = [ChainEngine(ts, te, total // P, model=0, seed=1, s_freq=100, n_trace_slots=100, chain_offset=p * (total // P)) for p in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.init(); e.steps(64)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for rep in range(n // 64):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.steps(64)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print('slots=%s P=%d tiles=%d: %.1f us per iteration of all %d chains -> %.3e evals/s' % (os.environ.get('LR_SLOTS'), P, engs[0].layout.tiles, el / n * 1e6, total, n * 1e5 * total / el), flush=True)
    for e in engs: e.close()
run(int(os.environ.get('P', '2')))
