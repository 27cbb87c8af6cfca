"""Multi-chain RJMCMC engine on one GPU: the host mirror of runMCMC (LiteRateForward.py:216-373).

All chain state lives in one torch uint8 workspace in HBM whose layout the C ABI reports
(lr_mcmc_query_layout); this class only creates views on it and calls lr_mcmc_*.
"""
import ctypes as C
import hashlib
import os

import numpy as np

from . import _hip, ops


class ChainEngine:
    def __init__(self, ts, te, n_chains, model=0, seed=1, const_rates=0, const_death_rate=0, use_rate_HP=1,
                 poisson_HP=0.0, update_fraction=0.75, s_freq=1000, n_trace_slots=0, chain_offset=0,
                 device=None, stats=None, sort_lineages=True, unit_resolution=None, engine="auto", dd=None, team=0,
                 chains_per_team=0):
        """ts/te: lineage birth/death times (death_jitter already added, LRF:471).

        stats: optional (t0, n_bins, br_length) if the caller already binned the data; otherwise the
        binning kernel runs here (LRF:515-523).
        dd: None = the runMCMC sampler (LRF:216-373); dict(m_birth, m_death, present, init_death) = the DDRate.py
        sampler (DD:124-241) on create_bins statistics: stats = (ORIGIN, N_TIME_BINS, DT), model 2;
        dict(kind="trend", m_birth=const_B, m_death=const_D) = the trend_rate.py sampler with
        stats = (ORIGIN, N_TIME_BINS, TREND)."""
        torch = _hip.require_gpu()
        self.lib = _hip.load()
        self.device = torch.device(device or "cuda")
        ts_d = ops._dev(ts, torch.float64, self.device)
        te_d = ops._dev(te, torch.float64, self.device)
        if sort_lineages:
            # HBM layout choice: lineages ordered by (birth bin, te) - for year-resolution data that is (ts, te).
            # A wave's 64 lineages then hit the same or neighbouring table entries (LDS gathers broadcast / stay
            # conflict-free), runs of one birth bin share their birth gather and consecutive lineages of a run die
            # in the same or a neighbouring bin, which is what the packed pair slots need (csrc/lr_pack.hip).  The
            # log-likelihood is a sum over lineages: the order changes only its rounding.
            o1 = torch.sort(te_d, stable=True).indices
            o2 = torch.sort(torch.floor(ts_d[o1]), stable=True).indices
            order = o1[o2]
            ts_d, te_d = ts_d[order].contiguous(), te_d[order].contiguous()
        self.ts, self.te = ts_d, te_d
        self.start_time = float(self.ts.min().item())
        self.end_time = float(self.te.max().item())
        if stats is None:
            t0, n_bins = int(self.start_time), int(self.end_time) - int(self.start_time)
            # the unit windows [i, i + 1], i in range(int(min ts), int(max te)) (LRF:519-523): one pass over the lineages
            self.sp_events, self.ex_events, self.br_length = ops.bin_unit_events(self.ts, self.te, float(t0), n_bins)
        else:
            t0, n_bins, br = stats
            self.br_length = ops._dev(br, torch.float64, self.device)
            self.sp_events = self.ex_events = None
        self.t0, self.n_bins, self.model = float(t0), int(n_bins), int(model)
        # unit-resolution data (integer years + death_jitter: every dataset the reference ships): all lineages
        # share the in-bin fractions, which the kernels then fold into 8-byte lookup tables
        fs = self.ts - torch.floor(self.ts)
        fe = self.te - (torch.ceil(self.te) - 1.0)
        fs0, fe0 = float(fs.min().item()), float(fe.min().item())
        is_unit = bool((fs.max().item() == fs0) and (fe.max().item() == fe0))
        if unit_resolution is None:
            unit_resolution = is_unit
        if unit_resolution and not is_unit:
            raise ValueError("unit_resolution=True but the lineages do not share their in-bin fractions")
        self.unit_resolution = bool(unit_resolution)
        self.cfg = _hip.McmcConfig(
            n_lineages=self.ts.numel(), n_bins=self.n_bins, n_chains=int(n_chains), model=int(model),
            const_rates=int(const_rates), const_death_rate=int(const_death_rate), use_rate_HP=int(use_rate_HP),
            s_freq=int(s_freq), n_trace_slots=int(n_trace_slots), poisson_HP=float(poisson_HP),
            update_fraction=float(update_fraction), t0=self.t0, start_time=self.start_time, end_time=self.end_time,
            seed=int(seed), chain_offset=int(chain_offset), unit_resolution=int(self.unit_resolution),
            engine_mode={"auto": 0, "launch": 1, "persistent": 2, "persistent4": 3, "persistent2": 4, "spec": 5, "stream": 6, "packed": 7}[engine],
            frac_birth=fs0 if self.unit_resolution else 0.0, frac_death=fe0 if self.unit_resolution else 0.0,
            sampler=0 if dd is None else (2 if dd.get("kind") == "trend" else 1),
            m_birth=0 if dd is None else int(dd["m_birth"]), m_death=0 if dd is None else int(dd["m_death"]),
            team_request=(int(team) or (1 if os.environ.get("LR_SHARED_DEVICE", "0") == "1" else 0)) | (int(chains_per_team) << 8),
            dd_present=0.0 if dd is None else float(dd.get("present", 0.0)),
            dd_init_death=0.0 if dd is None else float(dd.get("init_death", 0.1)))
        self.dd = dd
        self.layout = _hip.McmcLayout()
        with torch.cuda.device(self.device):      # the planner asks the CURRENT device for its compute units
            _hip.check(self.lib.lr_mcmc_query_layout(C.byref(self.cfg), C.byref(self.layout)), "lr_mcmc_query_layout")
        self.workspace = torch.zeros(self.layout.total_bytes, dtype=torch.uint8, device=self.device)
        handle = C.c_void_p()
        br_ptr = _hip.ptr(self.br_length) if (model in (0, 1) or dd is not None) else None
        with torch.cuda.device(self.device):      # the library creates its streams / events on the current device
            _hip.check(self.lib.lr_mcmc_create(C.byref(self.cfg), _hip.ptr(self.ts), _hip.ptr(self.te), br_ptr,
                                               _hip.ptr(self.workspace), self.workspace.numel(), C.byref(handle)),
                       "lr_mcmc_create")
        self.handle = handle
        self.n_chains = int(n_chains)
        self.iterations = 0
        self._hash = None
        self._plan_args = dict(model=model, const_rates=const_rates, const_death_rate=const_death_rate, use_rate_HP=use_rate_HP,
                               poisson_HP=poisson_HP, update_fraction=update_fraction, unit_resolution=self.unit_resolution, dd=dd,
                               stats=(self.t0, self.n_bins, self.br_length))
        self.plan_report = None
        if engine == "auto" and os.environ.get("LR_PLAN_CHECK", "0") == "1" and not ChainEngine._in_plan_check:
            self.plan_report = self.plan_check()

    _in_plan_check = False

    def plan_check(self, n_iters=200, tolerance=0.10):
        """LR_PLAN_CHECK=1: the planner's choice against a measurement ON THIS DEVICE.  lr_persist_variant picks the engine
        from cost models fitted to sweeps on the build pool's boxes (csrc/lr_mcmc.hip); here every engine that accepts
        this configuration runs n_iters iterations after as many of warm-up, from the CLI's initial state, and a warning
        names any engine that beats the planner's choice by more than `tolerance`.  Returns {"auto": kernel name,
        "us_per_iter": {engine: us or None when the engine refuses the configuration}, "best": engine, "ok": bool}."""
        import warnings
        torch = _hip.require_gpu()
        ChainEngine._in_plan_check = True
        times = {}
        try:
            for name in ("auto", "persistent4", "persistent2", "spec", "packed", "launch"):
                try:
                    e = ChainEngine(self.ts, self.te, self.n_chains, seed=int(self.cfg.seed), s_freq=1 << 30, n_trace_slots=2,
                                    chain_offset=int(self.cfg.chain_offset), device=self.device, sort_lineages=False,
                                    engine=name, **self._plan_args)
                except (ValueError, _hip.HipLibraryError):
                    times[name] = None
                    continue
                try:
                    e.init()
                    e.steps(n_iters)
                    torch.cuda.synchronize(self.device)
                    times[name] = e.timed_steps(n_iters) / n_iters * 1e3
                    if name == "auto":
                        auto_kernel = e.kernel_name()
                finally:
                    e.close()
        finally:
            ChainEngine._in_plan_check = False
        valid = {k: v for k, v in times.items() if v is not None and k != "auto"}
        best = min(valid, key=valid.get)
        ok = times["auto"] <= (1.0 + tolerance) * valid[best]
        if not ok:
            warnings.warn("LR_PLAN_CHECK: the planner chose %s (%.2f us per iteration) but engine=%r runs this configuration "
                          "in %.2f us on this device: pass engine=%r, and refit the planner's model (csrc/lr_mcmc.hip)"
                          % (auto_kernel, times["auto"], best, valid[best], best), RuntimeWarning)
        return dict(auto=auto_kernel, us_per_iter=times, best=best, ok=bool(ok))

    @property
    def _data_hash(self):
        """sha1 of the lineage arrays as the engine holds them (sorted): identifies the data in a checkpoint."""
        if self._hash is None:
            h = hashlib.sha1(self.ts.cpu().numpy().tobytes())
            h.update(self.te.cpu().numpy().tobytes())
            self._hash = h.hexdigest()
        return self._hash

    # ---- views on the workspace ----
    def _view(self, off, dtype, shape):
        torch = _hip.require_gpu()
        n = int(np.prod(shape))
        item = torch.empty((), dtype=dtype).element_size()
        return self.workspace[off:off + n * item].view(dtype).view(*shape)

    @property
    def state_f64(self):
        import torch
        return self._view(self.layout.state_f64, torch.float64, (self.n_chains, _hip.LR_STATE_ROWS, _hip.LR_ROW))

    @property
    def state_i32(self):
        import torch
        return self._view(self.layout.state_i32, torch.int32, (self.n_chains, _hip.LR_ISTATE_ROWS, _hip.LR_ROW))

    @property
    def trace(self):
        import torch
        return self._view(self.layout.trace, torch.float64,
                          (self.cfg.n_trace_slots, self.n_chains, _hip.LR_TRACE_W))

    # ---- control ----
    def init(self, L=None, M=None, tL=None, tM=None):
        """Initial state: None = CLI init (K=1, Gamma(2,2) rates; LRF:580-583); else lists/arrays per
        chain (ragged allowed: pass 2-D arrays padded with anything plus K inferred from times)."""
        import torch
        if L is None:
            rc = _hip.launch(self.lib.lr_mcmc_init, self.device, self.handle, None, None, None, None, None, None, 0)
        elif self.dd is not None:
            # parametric samplers: L = [C, 8] (DD:161) or [C, 6] (trend_rate.py:74) parameter vectors
            npar = 6 if self.dd.get("kind") == "trend" else 8
            args = np.zeros((self.n_chains, _hip.LR_KMAX))
            args[:, :npar] = np.asarray(L, dtype=float).reshape(self.n_chains, npar)
            self._init_keep = [ops._dev(args, torch.float64, self.device)]
            rc = _hip.launch(self.lib.lr_mcmc_init, self.device, self.handle, _hip.ptr(self._init_keep[0]), None, None, None,
                             None, None, _hip.LR_KMAX)
        else:
            kmax = _hip.LR_KMAX
            Ls = np.zeros((self.n_chains, kmax)); Ms = np.zeros((self.n_chains, kmax))
            tLs = np.zeros((self.n_chains, kmax + 1)); tMs = np.zeros((self.n_chains, kmax + 1))
            KL = np.zeros(self.n_chains, dtype=np.int32); KM = np.zeros(self.n_chains, dtype=np.int32)
            for c in range(self.n_chains):
                l, m = np.atleast_1d(L[c]), np.atleast_1d(M[c])
                KL[c], KM[c] = len(l), len(m)
                Ls[c, :len(l)], Ms[c, :len(m)] = l, m
                tLs[c, :len(l) + 1], tMs[c, :len(m) + 1] = tL[c], tM[c]
            dev = [ops._dev(x, torch.float64, self.device) for x in (Ls, Ms, tLs, tMs)]
            dk = [ops._dev(x, torch.int32, self.device) for x in (KL, KM)]
            self._init_keep = dev + dk
            rc = _hip.launch(self.lib.lr_mcmc_init, self.device, self.handle, *[_hip.ptr(x) for x in dev + dk], kmax)
        _hip.check(rc, "lr_mcmc_init")
        self.iterations = 0

    def _launch(self, fn, *args):
        """An ABI call on this engine's device and torch's current stream there.  When that device already is the
        current one (the normal case) the call goes straight through: the device guard of _hip.launch costs the host
        tens of microseconds, which a short timed region (bench.py at --steps 20) would be charged for."""
        import torch
        idx = self.device.index
        if idx is None or torch.cuda.current_device() == idx:
            return fn(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return _hip.launch(fn, self.device, *args)

    def steps(self, n):
        _hip.check(self._launch(self.lib.lr_mcmc_steps, self.handle, int(n)), "lr_mcmc_steps")
        self.iterations += int(n)

    def time_scan(self, reps=20):
        """Average duration (ms) of the lineage-scan kernel, HIP events on the launch stream."""
        ms = C.c_float(0.0)
        _hip.check(_hip.launch(self.lib.lr_mcmc_time_scan, self.device, self.handle, int(reps), C.byref(ms)),
                   "lr_mcmc_time_scan")
        return float(ms.value)

    def check_status(self):
        """Synchronise and raise if the engine flagged an error on the device (a bounded wait between blocks ran out - a
        team exchange of the speculative kernel, a counter of the resident streaming kernel: the blocks were not all
        resident)."""
        st = C.c_int32(0)
        _hip.check(_hip.launch(self.lib.lr_mcmc_status, self.device, self.handle, C.byref(st)), "lr_mcmc_status")
        if st.value != 0:
            raise _hip.HipLibraryError("engine status %d: a wait between blocks timed out (the kernel's blocks were not all "
                                       "resident: is the GPU shared? LR_SHARED_DEVICE=1 runs without teams, LR_STREAM=0 "
                                       "without the resident streaming kernel), the run is void" % st.value)

    def warnings(self):
        """Synchronise and return the engine's warning bits (include/literate_hip.h: LR_WARN_*)."""
        w = C.c_int32(0)
        _hip.check(_hip.launch(self.lib.lr_mcmc_warnings, self.device, self.handle, C.byref(w)), "lr_mcmc_warnings")
        return int(w.value)

    def warning_text(self):
        """Human-readable form of warnings() for the CLIs ('' when there is none)."""
        w = self.warnings()
        if w & _hip.LR_WARN_KCAP:
            return ("WARNING: an add-shift move was proposed from a state with %d rates and rejected: the device holds at "
                    "most %d rates per process (the reference has no cap, LiteRateForward.py:29-47); the posterior of "
                    "the number of shifts is truncated there" % (_hip.LR_KMAX, _hip.LR_KMAX))
        return ""

    def kernel_name(self):
        """Name of the kernel steps() spends its time in, as rocprofv3's kernel trace prints it."""
        buf = C.create_string_buffer(128)
        _hip.check(self.lib.lr_mcmc_describe(self.handle, buf, 128), "lr_mcmc_describe")
        return buf.value.decode()

    def timed_steps(self, n):
        """steps(n) bracketed by HIP events on the launch stream; returns elapsed device ms (blocks)."""
        ms = C.c_float(0.0)
        _hip.check(self._launch(self.lib.lr_mcmc_time_steps, self.handle, int(n), C.byref(ms)), "lr_mcmc_time_steps")
        self.iterations += int(n)
        return float(ms.value)

    def prepared_timed_steps(self, n):
        """timed_steps(n) with every argument marshalled beforehand: returns a zero-argument callable that makes the one
        ABI call (blocking) and returns the elapsed device ms.  For short timed regions (bench.py at --steps 20) the
        Python side of a call is a measurable share of the wall clock."""
        import torch
        ms = C.c_float(0.0)
        fn, handle, n_c, ms_ref = self.lib.lr_mcmc_time_steps, self.handle, C.c_int64(int(n)), C.byref(ms)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

        def call():
            rc = fn(handle, n_c, ms_ref, stream)
            if rc:
                _hip.check(rc, "lr_mcmc_time_steps")
            self.iterations += int(n)
            return float(ms.value)
        return call

    # ---- checkpoint / resume (no reference counterpart) ----
    _CFG_KEYS = ("n_lineages", "n_bins", "n_chains", "model", "const_rates", "const_death_rate", "use_rate_HP",
                 "s_freq", "n_trace_slots", "poisson_HP", "update_fraction", "t0", "start_time", "end_time", "seed",
                 "chain_offset", "unit_resolution", "frac_birth", "frac_death", "sampler", "m_birth", "m_death",
                 "dd_present", "dd_init_death", "engine_mode")

    def _signature(self):
        """What a checkpoint must agree on: every configuration field, the workspace layout and a hash of the lineages.
        Values are kept as strings (repr), so 64-bit seeds and offsets compare exactly."""
        sig = {k: repr(getattr(self.cfg, k)) for k in self._CFG_KEYS}
        sig.update({"layout_" + f: repr(getattr(self.layout, f)) for f, _ in self.layout._fields_})
        sig["data_sha1"] = self._data_hash
        sig["format"] = "2: state in .npz (head, tail), trace rows in .npz.trace"
        return sig

    def _trace_span(self):
        """(offset, end) of the trace region in the workspace (the status word follows it)."""
        return int(self.layout.trace), int(self.layout.status)

    def checkpoint_begin(self):
        """First half of save(), asynchronous: a device-side copy of everything a checkpoint holds EXCEPT the trace rows
        (chain states, pending proposals, tables, packed lineages, carried sums), taken on the current stream behind the
        steps() calls so far.  The next window can be launched right away; checkpoint_write() puts the ticket on disk
        later, from the side (the trace rows of a finished window never change again)."""
        import torch
        t0, t1 = self._trace_span()
        with torch.cuda.device(self.device):
            head, tail = self.workspace[:t0].clone(), self.workspace[t1:].clone()
            ev = torch.cuda.Event()
            ev.record()
        return dict(head=head, tail=tail, ev=ev, iterations=int(self.iterations), samples=int(self.samples_done()))

    def checkpoint_write(self, ticket, path):
        """Second half of save(): the ticket's state -> `path` (.npz, written beside the target and renamed over it: a
        kill leaves the previous checkpoint intact - also when a run starts over onto an existing checkpoint: its trace
        file is then replaced by a rename next to the .npz's own, not truncated in place), the trace rows sampled since the last write -> appended to
        `path`.trace (raw float64 rows [chains, LR_TRACE_W] per sample: a checkpoint per window used to rewrite the whole
        trace buffer every time - 1.2 GB per window at 1000 samples x 1024 chains).  The caller has made sure the run is
        not void (check_status(), or TraceStreamer.collect() of the same window)."""
        import torch
        path = str(path)
        if not path.endswith(".npz"):
            path += ".npz"
        side = getattr(self, "_ckpt_stream", None)
        if side is None:
            side = self._ckpt_stream = torch.cuda.Stream(device=self.device)
        row_bytes = self.n_chains * _hip.LR_TRACE_W * 8
        s1 = ticket["samples"]
        sidecar = path + ".trace"
        saved = getattr(self, "_ckpt_saved", {}).get(path, 0)
        if saved > s1 or not os.path.exists(sidecar) or os.path.getsize(sidecar) < saved * row_bytes:
            saved = 0
        with torch.cuda.device(self.device), torch.cuda.stream(side):
            side.wait_event(ticket["ev"])
            head, tail = ticket["head"].cpu().numpy(), ticket["tail"].cpu().numpy()
            new_rows = self.trace[saved:s1].cpu().numpy() if s1 > saved else None
        side.synchronize()
        sig = self._signature()
        tmp = path + ".tmp.npz"
        np.savez(tmp, head=head, tail=tail, iterations=np.int64(ticket["iterations"]), n_trace_rows=np.int64(s1),
                 sig_keys=np.array(list(sig.keys())), sig_vals=np.array(list(sig.values())))
        if saved > 0:
            # the usual case: rows behind the ones the .npz on disk records are appended (a kill before the rename below
            # leaves that checkpoint with surplus rows it never reads)
            with open(sidecar, "r+b") as f:
                f.seek(saved * row_bytes)
                f.truncate()
                if new_rows is not None:
                    f.write(new_rows.tobytes())
                f.flush()
                os.fsync(f.fileno())
        else:
            # a run that starts over onto an existing checkpoint: the old trace file is never truncated in place - the
            # new one is written beside it and renamed over it right before the .npz is
            with open(sidecar + ".tmp", "wb") as f:
                if new_rows is not None:
                    f.write(new_rows.tobytes())
                f.flush()
                os.fsync(f.fileno())
            os.replace(sidecar + ".tmp", sidecar)
        os.replace(tmp, path)
        if not hasattr(self, "_ckpt_saved"):
            self._ckpt_saved = {}
        self._ckpt_saved[path] = s1

    def save(self, path):
        """Write the run (all chain states, pending proposals, trace rows, iteration count) to `path` (.npz + .npz.trace).
        The workspace IS the run between two steps() calls; draws are addressed by (seed, chain, iteration), so a
        run resumed with load() continues bit-identically."""
        ticket = self.checkpoint_begin()
        self.check_status()          # (synchronises) a void run must not replace the last good checkpoint
        self.checkpoint_write(ticket, path)

    def load(self, path):
        """Resume from save(): the engine must have been created on the same data with the same settings."""
        import torch
        path = str(path)
        with np.load(path) as z:
            saved = dict(zip([str(k) for k in z["sig_keys"]], [str(v) for v in z["sig_vals"]]))
            mine = self._signature()
            bad = [k for k in mine if k not in saved or mine[k] != saved[k]]
            if bad or len(saved) != len(mine):
                raise ValueError("checkpoint was written by a different configuration: " + ", ".join(bad))
            t0, t1 = self._trace_span()
            head, tail = torch.from_numpy(z["head"]), torch.from_numpy(z["tail"])
            if head.numel() != t0 or tail.numel() != self.workspace.numel() - t1:
                raise ValueError("checkpoint workspace size differs")
            n_rows = int(z["n_trace_rows"])
            row_bytes = self.n_chains * _hip.LR_TRACE_W * 8
            if n_rows > 0:
                sidecar = path + ".trace"
                if not os.path.exists(sidecar) or os.path.getsize(sidecar) < n_rows * row_bytes:
                    raise ValueError("checkpoint trace file %s is missing or shorter than the %d rows the checkpoint holds" % (sidecar, n_rows))
                rows = np.fromfile(sidecar, dtype=np.uint8, count=n_rows * row_bytes)
                self.workspace[t0:t0 + n_rows * row_bytes].copy_(torch.from_numpy(rows).to(self.device))
            self.workspace[:t0].copy_(head.to(self.device))
            self.workspace[t1:].copy_(tail.to(self.device))
            self.iterations = int(z["iterations"])
        self._ckpt_saved = {path: n_rows}
        _hip.check(_hip.launch(self.lib.lr_mcmc_restore, self.device, self.handle), "lr_mcmc_restore")

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.lr_mcmc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-side snapshots ----
    def snapshot(self):
        """Accepted state of every chain as numpy: dict(L, M, tL, tM (lists), likA, priorA, K_l, K_m, ...)."""
        self.check_status()
        return self.snapshot_from(self.state_f64.cpu().numpy(), self.state_i32.cpu().numpy())

    def snapshot_from(self, S, I):
        """snapshot() of host copies of the two state blocks (TraceStreamer takes them at a window's end)."""
        KL, KM = I[:, _hip.IROW_SCALARS, _hip.I_KL], I[:, _hip.IROW_SCALARS, _hip.I_KM]
        sc = S[:, _hip.ROW_SCALARS]
        n = S.shape[0]
        return dict(
            K_l=KL.copy(), K_m=KM.copy(),
            L=[S[c, _hip.ROW_L, :KL[c]].copy() for c in range(n)],
            M=[S[c, _hip.ROW_M, :KM[c]].copy() for c in range(n)],
            tL=[S[c, _hip.ROW_TL, :KL[c] + 1].copy() for c in range(n)],
            tM=[S[c, _hip.ROW_TM, :KM[c] + 1].copy() for c in range(n)],
            likA=sc[:, _hip.S_LIKA].copy(), priorA=sc[:, _hip.S_PRIORA].copy(),
            gamma_rate=sc[:, [_hip.S_GRATE_L, _hip.S_GRATE_M]].copy(), poi=sc[:, _hip.S_POI].copy(),
            accepted=I[:, _hip.IROW_SCALARS, _hip.I_ACCEPTED].copy(),
            it=(I[:, _hip.IROW_SCALARS, _hip.I_IT_LO].astype(np.int64) & 0xFFFFFFFF)
               | (I[:, _hip.IROW_SCALARS, _hip.I_IT_HI].astype(np.int64) << 32),
        )

    def samples_done(self):
        """Trace rows written so far: iterations 0, s, 2 s, ... below `iterations` (LRF:321)."""
        return min(self.cfg.n_trace_slots, (self.iterations + self.cfg.s_freq - 1) // self.cfg.s_freq)

    def trace_rows(self, n_samples=None):
        """Trace buffer as numpy [samples, chains, LR_TRACE_W] (see include/literate_hip.h)."""
        self.check_status()
        n_avail = self.samples_done()
        n = n_avail if n_samples is None else min(n_samples, n_avail)
        return self.trace[:n].cpu().numpy()


def split_trace_row(row):
    """One trace row -> (mcmc head[13], sp_rates row, ex_rates row) in the reference's log layout
    (LRF:343-359): rates then interior shift times."""
    K = _hip.LR_KMAX
    head = row[:_hip.LR_TRACE_HEAD]
    kl, km = int(head[6]), int(head[7])
    rl = row[_hip.LR_TRACE_HEAD:_hip.LR_TRACE_HEAD + 2 * K - 1]
    rm = row[_hip.LR_TRACE_HEAD + 2 * K - 1:]
    sp = np.concatenate([rl[:kl], rl[K:K + kl - 1]])
    ex = np.concatenate([rm[:km], rm[K:K + km - 1]])
    return head, sp, ex


class TraceStreamer:
    """Streams a run out window by window (a window = the iterations of one steps() call = one print block): the
    reference writes and flushes its logs at every sample (LRF:334-359, DD:236-238); here the rows a window sampled
    and a copy of the chain states at its end leave the device on a SIDE stream - gathered to rank 0 over RCCL when
    the chains are sharded - while the NEXT window already runs on the main stream.

        eng.steps(n); st.mark()           # window k enqueued; its end recorded
        eng.steps(n); st.mark()           # window k + 1 enqueued
        rows, snap, win = st.collect()    # window k: [samples, total chains, LR_TRACE_W] on rank 0 (None elsewhere)

    The trace keeps every sampled row resident (slot = sample number), so a window's rows are not touched by later
    windows; the state blocks are copied device-to-device at the window's end on the main stream (a later launch
    rewrites them).

    Co-residency: the copies and the RCCL gather of window k run on the side stream WHILE window k + 1's kernel is
    resident.  The speculative kernel's teams of blocks exchange partial sums and need every block of the launch
    resident at once (the planner never plans more blocks than the device has CUs; an exchange that stalls for 2 s
    raises the status word): RCCL's kernels take a few CUs' worth of workgroup slots beside it, not whole CUs, and the
    planner's blocks leave room for them (one 512-thread block per CU).  A device shared with OTHER processes gives no
    such guarantee: set LR_SHARED_DEVICE=1 there (no teams)."""

    def __init__(self, eng, total_chains=None, n_local=None, gather=True):
        """gather=False: every rank keeps (and writes) the rows of its own chains (the DDRate / trend_rate CLIs)."""
        import torch
        self.eng, self.torch, self.gather = eng, torch, bool(gather)
        self.total = eng.n_chains if total_chains is None else int(total_chains)
        self.n_local = eng.n_chains if n_local is None else int(n_local)
        self.side = torch.cuda.Stream(device=eng.device)
        self.pending = []
        self.done_samples = 0

    def rewind(self, n_samples=0):
        """Start (again) from sample `n_samples`: after load() the windows up to the checkpoint are re-read as one."""
        self.done_samples = int(n_samples)

    def mark(self):
        torch, eng = self.torch, self.eng
        with torch.cuda.device(eng.device):
            S, I = eng.state_f64.clone(), eng.state_i32.clone()          # on the main stream, behind the window
            st = eng.workspace[eng.layout.status:eng.layout.status + 8].clone()
            ev = torch.cuda.Event()
            ev.record()
        s1 = eng.samples_done()
        self.pending.append((self.done_samples, s1, eng.iterations, S, I, st, ev))
        self.done_samples = s1

    def collect(self):
        """Oldest marked window -> (rows or None, snapshot dict, (first sample, end sample, iterations at its end))."""
        from . import dist as lrd
        torch, eng = self.torch, self.eng
        s0, s1, its, S, I, st, ev = self.pending.pop(0)
        with torch.cuda.device(eng.device), torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            status = int(st.view(torch.int32).cpu()[0])
            if self.gather:
                # a collective of its own BEFORE the gather: every rank sees the worst status and raises alike - a rank
                # that raised alone would leave the others blocked in dist.gather for good
                status = lrd.agree_status(status, eng.device)
            if status != 0:
                raise _hip.HipLibraryError("engine status %d on at least one rank: a team exchange timed out, the run "
                                           "is void" % status)
            local = eng.trace[s0:s1][:, :self.n_local].contiguous()
            # (a window without a sample is empty on every rank alike: no collective for it)
            rows = lrd.gather_traces(local, self.total) if (self.gather and s1 > s0) else local
            rows = rows.cpu().numpy() if rows is not None else None
            snap = eng.snapshot_from(S.cpu().numpy(), I.cpu().numpy())
        self.side.synchronize()
        return rows, snap, (s0, s1, its)
