"""Device time of lr_loglik_small_kernel on the one-state calc_likelihood seam (metal_bands): 300 back-to-back calls between
HIP events, with the rates / result in device memory and in pinned host memory (the zero-copy session)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from literate_amd import ops, _hip
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "binning_lik.npz"))
name = "metal_bands"
ts, te = G[name + "/ts"], G[name + "/te"]
start, end = G[name + "/start_end"]
n_bins = len(G[name + "/sp"])
lam = np.full(n_bins, .3); mu = np.full(n_bins, .2)
for model in (0, 2):
    ses = ops.LoglikSession(ts, te, float(int(start)), n_bins, 1, model, G[name + "/br"], end)
    assert ses.zero_copy
    ses(lam, mu)
    host_args = ses.args
    dev_args = list(ses.args)
    ses.rates.copy_(ses.rates_host)
    dev_args[5], dev_args[6], dev_args[11] = _hip.ptr(ses.rates[0]), _hip.ptr(ses.rates[1]), _hip.ptr(ses.out)
    for label, args in (("device pointers", tuple(dev_args)), ("pinned host pointers", host_args)):
        with torch.cuda.stream(ses.stream):
            for _ in range(50):
                ses.lib.lr_bd_loglik_batch(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ses.stream)
            for _ in range(300):
                ses.lib.lr_bd_loglik_batch(*args)
            e1.record(ses.stream)
        ses.stream.synchronize()
        print("model %d, %-22s: %.2f us per call (device, back to back)" % (model, label, e0.elapsed_time(e1) / 300 * 1e3), flush=True)
