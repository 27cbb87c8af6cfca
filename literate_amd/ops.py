"""Tensor-level wrappers of the C ABI: torch tensors in HBM in, torch tensors out.

Every function enqueues on torch's current HIP stream and raises if the HIP library or a GPU is
missing (no CPU path).  Shapes follow include/literate_hip.h.
"""
import os

import numpy as np

from . import _hip

_ws_cache = {}


def _torch():
    return _hip.require_gpu()


def _dev(x, dtype, device=None):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        # a tensor that already lives on a GPU stays there unless the caller names another device
        t = x.to(device=device or (x.device if x.is_cuda else "cuda"), dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(device or "cuda")
    return t.contiguous()


def _host_f64(x):
    """host numpy float64 view / copy of a tensor or array-like"""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def _workspace(nbytes, device):
    """A reusable byte workspace per device (grown on demand)."""
    torch = _torch()
    key = str(device)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _unit_windows(win_lo, win_hi):
    """(t0, n_bins) when the windows are the unit bins [t0 + w, t0 + w + 1] with t0 integer valued - what the reference
    always bins into (LRF:519-523, lib create_bins) - else None.  Decided on host arrays only (no device read-back)."""
    if hasattr(win_lo, "data_ptr") or hasattr(win_hi, "data_ptr"):     # torch tensors: possibly on the device, not inspected
        return None
    lo, hi = np.asarray(win_lo, dtype=float).ravel(), np.asarray(win_hi, dtype=float).ravel()
    if lo.size < 1 or hi.size != lo.size or lo.size > _hip.LR_MAX_BINS:
        return None
    t0 = float(lo[0])
    if not (np.isfinite(t0) and t0 == np.floor(t0) and abs(t0) < 1e9):
        return None
    grid = t0 + np.arange(lo.size, dtype=float)
    if np.array_equal(lo, grid) and np.array_equal(hi, grid + 1.0):
        return t0, int(lo.size)
    return None


def bin_unit_events(ts, te, t0, n_bins):
    """(sp_events int64[n_bins], ex_events, br_length f64) of the unit windows [t0 + w, t0 + w + 1]: one pass over the
    lineages (lr_bin_unit_events; the loop LRF:519-523 / lib create_bins:231-245)."""
    torch = _torch()
    lib = _hip.load()
    ts = _dev(ts, torch.float64)
    te = _dev(te, torch.float64, ts.device)
    n = ts.numel()
    if te.numel() != n:
        raise ValueError("ts and te differ in length")
    sp = torch.empty(n_bins, dtype=torch.int64, device=ts.device)
    ex = torch.empty(n_bins, dtype=torch.int64, device=ts.device)
    br = torch.empty(n_bins, dtype=torch.float64, device=ts.device)
    nbytes = lib.lr_bin_unit_events_workspace_bytes(n, n_bins)
    if nbytes < 0:
        _hip.check(int(nbytes), "lr_bin_unit_events_workspace_bytes")
    ws = _workspace(nbytes, ts.device)
    rc = _hip.launch(lib.lr_bin_unit_events, ts.device, _hip.ptr(ts), _hip.ptr(te), n, float(t0), int(n_bins), _hip.ptr(sp),
                     _hip.ptr(ex), _hip.ptr(br), _hip.ptr(ws), ws.numel())
    _hip.check(rc, "lr_bin_unit_events")
    return sp, ex, br


def bin_events(ts, te, win_lo, win_hi):
    """(sp_events int64[W], ex_events int64[W], br_length f64[W]) for windows [lo_w, hi_w]
    (precompute_events / get_br, lib:74-85, for all windows in one launch).  Unit windows on an integer origin - every
    binning call of the reference's CLIs - take the one-pass kernel (bin_unit_events)."""
    torch = _torch()
    lib = _hip.load()
    unit = _unit_windows(win_lo, win_hi)
    if unit is not None:
        return bin_unit_events(ts, te, unit[0], unit[1])
    ts = _dev(ts, torch.float64)
    te = _dev(te, torch.float64, ts.device)
    lo, hi = _dev(win_lo, torch.float64, ts.device), _dev(win_hi, torch.float64, ts.device)
    n, w = ts.numel(), lo.numel()
    if te.numel() != n or hi.numel() != w:
        raise ValueError("ts/te or window arrays differ in length")
    sp = torch.empty(w, dtype=torch.int64, device=ts.device)
    ex = torch.empty(w, dtype=torch.int64, device=ts.device)
    br = torch.empty(w, dtype=torch.float64, device=ts.device)
    nbytes = lib.lr_bin_events_workspace_bytes(n, w)
    if nbytes < 0:
        _hip.check(int(nbytes), "lr_bin_events_workspace_bytes")
    ws = _workspace(nbytes, ts.device)
    rc = _hip.launch(lib.lr_bin_events, ts.device, _hip.ptr(ts), _hip.ptr(te), n, _hip.ptr(lo), _hip.ptr(hi), w, _hip.ptr(sp), _hip.ptr(ex),
                           _hip.ptr(br), _hip.ptr(ws), ws.numel())
    _hip.check(rc, "lr_bin_events")
    return sp, ex, br


def expand_rates(rates, times, K, n_bins, mode=0):
    """[C,n_bins] per-bin rates from K segment rates (get_rate_index + L[ind], LRF:125-135)."""
    torch = _torch()
    lib = _hip.load()
    rates, times = _dev(rates, torch.float64), _dev(times, torch.float64)
    K = _dev(K, torch.int32)
    C, kmax = rates.shape
    if times.shape != (C, kmax + 1) or K.shape != (C,):
        raise ValueError("shape mismatch: rates [C,kmax], times [C,kmax+1], K [C]")
    out = torch.empty((C, n_bins), dtype=torch.float64, device=rates.device)
    rc = _hip.launch(lib.lr_expand_rates, rates.device, _hip.ptr(rates), _hip.ptr(times), _hip.ptr(K), kmax, C, n_bins, mode, _hip.ptr(out))
    _hip.check(rc, "lr_expand_rates")
    return out


def bd_loglik_batch(ts, te, t0, lam_bins, mu_bins, model=2, br_length=None, end_time=0.0):
    """out[C]: per-lineage birth-death log-likelihood of C per-bin rate vectors (LRF:137-162,
    BDIx:124-146).  ts/te are scanned once per group of chains."""
    torch = _torch()
    lib = _hip.load()
    ts = _dev(ts, torch.float64)
    te = _dev(te, torch.float64, ts.device)
    lam, mu = _dev(lam_bins, torch.float64, ts.device), _dev(mu_bins, torch.float64, ts.device)
    if lam.dim() == 1:
        lam, mu = lam[None, :], mu[None, :]
    C, n_bins = lam.shape
    if mu.shape != lam.shape or te.numel() != ts.numel():
        raise ValueError("shape mismatch")
    br = None if br_length is None else _dev(br_length, torch.float64, ts.device)
    if br is not None and br.numel() != n_bins:
        raise ValueError("br_length must have n_bins entries")
    out = torch.empty(C, dtype=torch.float64, device=ts.device)
    nbytes = lib.lr_bd_loglik_workspace_bytes(ts.numel(), n_bins, C, model)
    if nbytes < 0:
        _hip.check(int(nbytes), "lr_bd_loglik_workspace_bytes")
    ws = _workspace(nbytes, ts.device)
    rc = _hip.launch(lib.lr_bd_loglik_batch, ts.device, _hip.ptr(ts), _hip.ptr(te), ts.numel(), float(t0), n_bins, _hip.ptr(lam), _hip.ptr(mu),
                                C, model, _hip.ptr(br), float(end_time), _hip.ptr(out), _hip.ptr(ws), ws.numel())
    _hip.check(rc, "lr_bd_loglik_batch")
    return out


class LoglikSession:
    """The calc_likelihood seam (LRF:305-308: one call per MCMC iteration) with everything that does not change
    between calls prepared once: lineages and workspace resident in HBM, ONE pinned staging buffer for what a call may
    change - the per-bin rates of `n_states` states and br_length (the reference's operator reads the module global
    br_length_bin on every call, LRF:150-162, so the session takes it per call too) - and one for the result.
    Few states on few lineages: one launch that reads the pinned buffer and writes its result into pinned host memory
    the host polls (no copy, no stream synchronisation).  Otherwise: one host-to-device copy, the three launches of
    lr_bd_loglik_batch, one device-to-host copy and one stream synchronisation."""

    def __init__(self, ts, te, t0, n_bins, n_states, model=2, br_length=None, end_time=0.0):
        torch = _torch()
        self.lib = _hip.load()
        self.ts = _dev(ts, torch.float64)
        self.te = _dev(te, torch.float64, self.ts.device)
        dev = self.device = self.ts.device
        self.n, self.n_bins, self.C, self.model = self.ts.numel(), int(n_bins), int(n_states), int(model)
        self.has_br = br_length is not None
        # [lam: C x n_bins | mu: C x n_bins | br_length: n_bins], one pinned block and its device mirror
        self.stage_host = torch.zeros((2 * self.C + 1) * self.n_bins, dtype=torch.float64).pin_memory()
        self.stage_np = self.stage_host.numpy()
        self.rates_np = self.stage_np[:2 * self.C * self.n_bins].reshape(2, self.C, self.n_bins)
        self.br_np = self.stage_np[2 * self.C * self.n_bins:]
        if self.has_br:
            self.br_np[:] = _host_f64(br_length)
        self.stage = torch.empty_like(self.stage_host, device=dev)
        self.out = torch.empty(self.C, dtype=torch.float64, device=dev)
        self.out_host = torch.empty(self.C, dtype=torch.float64).pin_memory()
        self.out_np = self.out_host.numpy()
        nbytes = self.lib.lr_bd_loglik_workspace_bytes(self.n, self.n_bins, self.C, self.model)
        if nbytes < 0:
            _hip.check(int(nbytes), "lr_bd_loglik_workspace_bytes")
        self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)   # the session's own: nothing else scribbles on it
        self.stream = torch.cuda.current_stream(dev)
        self.zero_copy = (self.C <= 16 and self.n <= (1 << 18) and self.n * self.C <= (1 << 21) and self.n_bins <= 900
                          and os.environ.get("LR_LOGLIK_SMALL", "1") != "0")
        self.out_bits = self.out_np.view(np.uint64)
        self._t0, self._end_time = float(t0), float(end_time)
        self.args_zero = self._args(self.stage_host, self.out_host)
        self.args_copy = self._args(self.stage, self.out)

    def _args(self, stage, out):
        nb, C = self.n_bins, self.C
        base = stage.data_ptr()
        br = _hip.c_vp(base + 16 * C * nb) if self.has_br else None
        return (_hip.ptr(self.ts), _hip.ptr(self.te), self.n, self._t0, nb, _hip.c_vp(base), _hip.c_vp(base + 8 * C * nb),
                C, self.model, br, self._end_time, _hip.ptr(out), _hip.ptr(self.ws), self.ws.numel(),
                _hip.c_vp(self.stream.cuda_stream))

    _SENTINEL = np.uint64(0x7FF8DEAD0000BEEF)      # a NaN payload no arithmetic produces: "not written yet"

    def __call__(self, L, M, br_length=None):
        """L, M: [n_states, n_bins] (or [n_bins] when n_states == 1) host arrays, br_length: this call's [n_bins]
        (None: the one given at construction) -> numpy [n_states] (a view of the session's pinned result buffer:
        valid until the next call)."""
        torch = _torch()
        self.rates_np[0] = L
        self.rates_np[1] = M
        if br_length is not None and self.has_br:
            self.br_np[:] = br_length
        # the launch goes to the stream captured at construction: that stream's device must be current for the call
        switch = torch.cuda.current_device() != self.device.index
        if switch:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(self.device)
        try:
            if self.zero_copy:
                self.out_bits[:] = self._SENTINEL
                rc = self.lib.lr_bd_loglik_batch(*self.args_zero)
                _hip.check(rc, "lr_bd_loglik_batch")
                bits, sent = self.out_bits, self._SENTINEL
                for _ in range(200000):                       # ~0.2 s of polling at most, then the ordinary wait
                    if not (bits == sent).any():
                        return self.out_np
                # the result did not become visible to the polling host (non-coherent pinned memory, a stalled device):
                # wait the ordinary way, and do not poll again in this session
                self.zero_copy = False
                self.stream.synchronize()
                if (bits == sent).any():
                    raise _hip.HipLibraryError("lr_bd_loglik_batch: the result never arrived in host memory")
                return self.out_np
            with torch.cuda.stream(self.stream):
                self.stage.copy_(self.stage_host, non_blocking=True)
                rc = self.lib.lr_bd_loglik_batch(*self.args_copy)
                _hip.check(rc, "lr_bd_loglik_batch")
                self.out_host.copy_(self.out, non_blocking=True)
            self.stream.synchronize()
            return self.out_np
        finally:
            if switch:
                torch.cuda.set_device(prev)


def rj_propose_score(rates, times, K, move, index, draws, mult_d=1.1):
    """Batched explicit-draw proposal scorer (LRF:29-69, 165-176).  Returns
    (rates'[C,kmax], times'[C,kmax+1], K'[C], score[C])."""
    torch = _torch()
    lib = _hip.load()
    rates, times = _dev(rates, torch.float64), _dev(times, torch.float64)
    K, move, index = _dev(K, torch.int32), _dev(move, torch.int32), _dev(index, torch.int32)
    draws = _dev(draws, torch.float64)
    C, kmax = rates.shape
    if times.shape != (C, kmax + 1) or draws.shape != (C, 2 * kmax):
        raise ValueError("shape mismatch: times [C,kmax+1], draws [C,2*kmax]")
    o_r, o_t = torch.empty_like(rates), torch.empty_like(times)
    o_k = torch.empty_like(K)
    o_s = torch.empty(C, dtype=torch.float64, device=rates.device)
    rc = _hip.launch(lib.lr_rj_propose_score, rates.device, _hip.ptr(rates), _hip.ptr(times), _hip.ptr(K), kmax, C, _hip.ptr(move), _hip.ptr(index),
                                 _hip.ptr(draws), float(mult_d), _hip.ptr(o_r), _hip.ptr(o_t), _hip.ptr(o_k),
                                 _hip.ptr(o_s))
    _hip.check(rc, "lr_rj_propose_score")
    return o_r, o_t, o_k, o_s


def log_priors(rates, K, shape, gamma_rate, poi_rate=None):
    """out[C] = prior_gamma(rates[:K], shape, gamma_rate) (+ Poisson_prior(K, poi_rate)) (LRF:198-202)."""
    torch = _torch()
    lib = _hip.load()
    rates, K = _dev(rates, torch.float64), _dev(K, torch.int32)
    C, kmax = rates.shape
    g = _dev(gamma_rate, torch.float64)
    p = None if poi_rate is None else _dev(poi_rate, torch.float64)
    out = torch.empty(C, dtype=torch.float64, device=rates.device)
    rc = _hip.launch(lib.lr_log_priors, rates.device, _hip.ptr(rates), _hip.ptr(K), kmax, C, float(shape), _hip.ptr(g), _hip.ptr(p), _hip.ptr(out))
    _hip.check(rc, "lr_log_priors")
    return out


def dd_rates(args, DT, m_birth=2, m_death=2):
    """DDRate per-bin (birth, death, niche, niche_frac), each [C,n_bins] (DD:71-100)."""
    torch = _torch()
    lib = _hip.load()
    args, DT = _dev(args, torch.float64), _dev(DT, torch.float64)
    if args.dim() == 1:
        args = args[None, :]
    C, n_bins = args.shape[0], DT.numel()
    outs = [torch.empty((C, n_bins), dtype=torch.float64, device=args.device) for _ in range(4)]
    rc = _hip.launch(lib.lr_dd_rates, args.device, _hip.ptr(args), _hip.ptr(DT), n_bins, C, m_birth, m_death, *[_hip.ptr(o) for o in outs])
    _hip.check(rc, "lr_dd_rates")
    return tuple(outs)


def ddv2_rates(args, DT, m_birth=2, m_death=2):
    """DDRatev2 per-bin (birth, death, niche, niche_frac), each [C,n_bins] (DDRatev2.py:73-104); args [C,9]."""
    torch = _torch()
    lib = _hip.load()
    args, DT = _dev(args, torch.float64), _dev(DT, torch.float64)
    if args.dim() == 1:
        args = args[None, :]
    if args.shape[1] != 9:
        raise ValueError("DDRatev2 takes 9 parameters per state")
    C, n_bins = args.shape[0], DT.numel()
    outs = [torch.empty((C, n_bins), dtype=torch.float64, device=args.device) for _ in range(4)]
    rc = _hip.launch(lib.lr_ddv2_rates, args.device, _hip.ptr(args), _hip.ptr(DT), n_bins, C, m_birth, m_death, *[_hip.ptr(o) for o in outs])
    _hip.check(rc, "lr_ddv2_rates")
    return tuple(outs)


def trend_rates(args, trend, const_birth=False, const_death=False):
    """trend_rate.py:73-88 per-bin (birth, death), each [C,n_bins]; args [C,6], trend = normalised covariate."""
    torch = _torch()
    lib = _hip.load()
    args, trend = _dev(args, torch.float64), _dev(trend, torch.float64)
    if args.dim() == 1:
        args = args[None, :]
    if args.shape[1] != 6:
        raise ValueError("trend_rate takes 6 parameters per state")
    C, n_bins = args.shape[0], trend.numel()
    outs = [torch.empty((C, n_bins), dtype=torch.float64, device=args.device) for _ in range(2)]
    rc = _hip.launch(lib.lr_trend_rates, args.device, _hip.ptr(args), _hip.ptr(trend), n_bins, C, int(bool(const_birth)), int(bool(const_death)),
                            *[_hip.ptr(o) for o in outs])
    _hip.check(rc, "lr_trend_rates")
    return tuple(outs)


def binned_keiding(birth, death, n_spec, n_exti, DT):
    """(birth_lik[C], death_lik[C]) = sum_b log(rate) * events - rate * DT (DD:86, 101) for C per-bin rate vectors."""
    torch = _torch()
    lib = _hip.load()
    birth, death, DT = _dev(birth, torch.float64), _dev(death, torch.float64), _dev(DT, torch.float64)
    n_spec, n_exti = _dev(n_spec, torch.int64), _dev(n_exti, torch.int64)
    if birth.dim() == 1:
        birth, death = birth[None, :], death[None, :]
    C, n_bins = birth.shape
    if death.shape != birth.shape or DT.numel() != n_bins or n_spec.numel() != n_bins or n_exti.numel() != n_bins:
        raise ValueError("shape mismatch")
    ob = torch.empty(C, dtype=torch.float64, device=birth.device)
    od = torch.empty_like(ob)
    rc = _hip.launch(lib.lr_binned_keiding, birth.device, _hip.ptr(birth), _hip.ptr(death), _hip.ptr(n_spec), _hip.ptr(n_exti), _hip.ptr(DT),
                               n_bins, C, _hip.ptr(ob), _hip.ptr(od))
    _hip.check(rc, "lr_binned_keiding")
    return ob, od


def simulate_bd(n_start, n_steps, seed, lam_steps=None, mu_steps=None, mode=0, l0=0.0, m0=0.0, K=1.0, scale=1.0,
                capacity=None, device=None):
    """Discrete-time birth-death simulation on the device (simulateRateABC.v2.py:103-234 / notebook 4 Simulator).
    Returns (ts, te, alive_trace) as device tensors: birth / death step per lineage (te = n_steps: extant), trimmed to
    the lineages created, in slot order, and the living count per step.  Raises OverflowError when `capacity`
    (default 64 x n_start, at least 1M) was hit."""
    torch = _torch()
    lib = _hip.load()
    dev = device or "cuda"
    capacity = int(capacity or max(64 * n_start, 1 << 20))
    lam = None if lam_steps is None else _dev(lam_steps, torch.float64, dev)
    mu = None if mu_steps is None else _dev(mu_steps, torch.float64, dev)
    if mode == 0 and (lam is None or mu is None or lam.numel() < n_steps or mu.numel() < n_steps):
        raise ValueError("mode 0 needs lam_steps and mu_steps with n_steps entries")
    ts = torch.empty(capacity, dtype=torch.float64, device=dev)
    te = torch.empty(capacity, dtype=torch.float64, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    trace = torch.zeros(int(n_steps), dtype=torch.int64, device=dev)
    ws = torch.zeros(64, dtype=torch.uint8, device=dev)
    rc = _hip.launch(lib.lr_simulate_bd, ts.device, _hip.ptr(lam), _hip.ptr(mu), int(n_steps), int(mode), float(l0), float(m0), float(K),
                            float(scale), int(n_start), capacity, int(seed) & 0xFFFFFFFFFFFFFFFF, _hip.ptr(ts), _hip.ptr(te),
                            _hip.ptr(counters), _hip.ptr(trace), _hip.ptr(ws), ws.numel())
    _hip.check(rc, "lr_simulate_bd")
    n, alive, overflow, _ = [int(v) for v in counters.cpu()]
    if overflow:
        raise OverflowError("simulate_bd: more than %d lineages; pass a larger capacity" % capacity)
    return ts[:n], te[:n], trace


def debug_draws(seed, chain, it, purpose, idx, kind, shape):
    """Device RNG probe: kind 0 u_a, 1 u_b, 2 normal, 3 gamma(shape) at (it, purpose, idx)."""
    torch = _torch()
    lib = _hip.load()
    it = _dev(it, torch.int64)
    purpose, idx, kind = _dev(purpose, torch.int32), _dev(idx, torch.int32), _dev(kind, torch.int32)
    shape = _dev(shape, torch.float64)
    out = torch.empty(it.numel(), dtype=torch.float64, device=it.device)
    rc = _hip.launch(lib.lr_debug_draws, it.device, int(seed), int(chain), _hip.ptr(it), _hip.ptr(purpose), _hip.ptr(idx), _hip.ptr(kind),
                            _hip.ptr(shape), it.numel(), _hip.ptr(out))
    _hip.check(rc, "lr_debug_draws")
    return out
