"""ctypes binding of libliterate_hip.so (include/literate_hip.h).  Fails loudly: no fallback."""
import ctypes as C
import os

from .build import LIB

c_i32, c_i64, c_f64, c_vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p

LR_ERR_NULL, LR_ERR_SIZE, LR_ERR_MODEL, LR_ERR_WORKSPACE, LR_ERR_T0, LR_ERR_STATE, LR_ERR_ORDER = -1, -2, -3, -4, -5, -6, -7
ERRORS = {-1: "LR_ERR_NULL", -2: "LR_ERR_SIZE", -3: "LR_ERR_MODEL", -4: "LR_ERR_WORKSPACE", -5: "LR_ERR_T0",
          -6: "LR_ERR_STATE", -7: "LR_ERR_ORDER (lineages must be sorted by birth time for the persistent engines)"}

LR_KMAX, LR_ROW, LR_MAX_BINS = 32, 64, 4094
LR_WARN_KCAP = 2
LR_STATE_ROWS, LR_ISTATE_ROWS = 9, 5
LR_TRACE_HEAD = 13
LR_TRACE_W = LR_TRACE_HEAD + 2 * (2 * LR_KMAX - 1)
# rows / scalar slots (include/literate_hip.h)
ROW_L, ROW_M, ROW_TL, ROW_TM, ROW_PL, ROW_PM, ROW_PTL, ROW_PTM, ROW_SCALARS = range(9)
(S_LIKA, S_PRIORA, S_PRIORPOIA, S_GRATE_L, S_GRATE_M, S_POI, S_HASTING, S_PRIOR_P, S_PRIORPOI_P, S_CONST_P,
 S_CONST_A, S_LIK_P, S_LOG_G0, S_LOG_G1, S_LOG_POI) = range(15)
IROW_EL, IROW_EM, IROW_PEL, IROW_PEM, IROW_SCALARS = range(5)
(I_KL, I_KM, I_PKL, I_PKM, I_GIBBS, I_INVALID, I_IT_LO, I_IT_HI, I_ACCEPTED, I_MOVE, I_NEXT_LO, I_NEXT_HI,
 I_SLOT) = range(13)


class McmcConfig(C.Structure):
    _fields_ = [("n_lineages", c_i64), ("n_bins", c_i32), ("n_chains", c_i32), ("model", c_i32),
                ("const_rates", c_i32), ("const_death_rate", c_i32), ("use_rate_HP", c_i32), ("s_freq", c_i32),
                ("n_trace_slots", c_i32), ("poisson_HP", c_f64), ("update_fraction", c_f64), ("t0", c_f64),
                ("start_time", c_f64), ("end_time", c_f64), ("seed", C.c_uint64), ("chain_offset", c_i64),
                ("unit_resolution", c_i32), ("engine_mode", c_i32), ("frac_birth", c_f64), ("frac_death", c_f64),
                ("sampler", c_i32), ("m_birth", c_i32), ("m_death", c_i32), ("team_request", c_i32),
                ("dd_present", c_f64), ("dd_init_death", c_f64)]


class McmcLayout(C.Structure):
    _fields_ = [("state_f64", c_i64), ("state_i32", c_i64), ("bin_consts", c_i64), ("lineage_idx", c_i64), ("args_blob", c_i64), ("tables", c_i64),
                ("partials", c_i64), ("trace", c_i64), ("total_bytes", c_i64), ("table_stride", c_i32),
                ("tiles", c_i32), ("chains_per_block", c_i32), ("trace_width", c_i32), ("n_parts", c_i32),
                ("pipelined", c_i32), ("persistent", c_i32), ("reserved1", c_i32), ("status", c_i64), ("xchg", c_i64),
                ("team_blocks", c_i32), ("table_mode", c_i32), ("lineage_frac", c_i64), ("pack_tmp", c_i64),
                ("spec_chains_per_team", c_i32), ("streaming", c_i32), ("packed_scan", c_i32), ("reserved3", c_i32)]


# name -> (restype, argtypes); exactly the symbols include/literate_hip.h declares (+ the RNG debug hook)
SIGNATURES = {
    "lr_version": (c_i32, []),
    "lr_bin_events_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "lr_bin_events": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "lr_bin_unit_events_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "lr_bin_unit_events": (c_i32, [c_vp, c_vp, c_i64, c_f64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "lr_expand_rates": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "lr_bd_loglik_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i32]),
    "lr_bd_loglik_plan": (c_i32, [c_i64, c_i32, c_i32, c_i32, C.POINTER(c_i32)]),
    "lr_bd_loglik_batch": (c_i32, [c_vp, c_vp, c_i64, c_f64, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_f64, c_vp,
                                   c_vp, c_i64, c_vp]),
    "lr_rj_propose_score": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp, c_vp, c_vp,
                                    c_vp]),
    "lr_log_priors": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_f64, c_vp, c_vp, c_vp, c_vp]),
    "lr_dd_rates": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "lr_ddv2_rates": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "lr_binned_keiding": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "lr_simulate_bd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_f64, c_f64, c_f64, c_f64, c_i64, c_i64, C.c_uint64, c_vp, c_vp,
                             c_vp, c_vp, c_vp, c_i64, c_vp]),
    "lr_trend_rates": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "lr_format_rows": (c_i64, [c_vp, c_vp, c_i64, C.c_uint64, c_i32, c_vp, c_i64]),
    "lr_mcmc_query_layout": (c_i32, [C.POINTER(McmcConfig), C.POINTER(McmcLayout)]),
    "lr_mcmc_create": (c_i32, [C.POINTER(McmcConfig), c_vp, c_vp, c_vp, c_vp, c_i64, C.POINTER(c_vp)]),
    "lr_mcmc_init": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "lr_mcmc_steps": (c_i32, [c_vp, c_i64, c_vp]),
    "lr_mcmc_time_scan": (c_i32, [c_vp, c_i32, C.POINTER(C.c_float), c_vp]),
    "lr_mcmc_time_steps": (c_i32, [c_vp, c_i64, C.POINTER(C.c_float), c_vp]),
    "lr_mcmc_restore": (c_i32, [c_vp, c_vp]),
    "lr_mcmc_describe": (c_i32, [c_vp, C.c_char_p, c_i32]),
    "lr_mcmc_status": (c_i32, [c_vp, C.POINTER(c_i32), c_vp]),
    "lr_mcmc_warnings": (c_i32, [c_vp, C.POINTER(c_i32), c_vp]),
    "lr_mcmc_destroy": (c_i32, [c_vp]),
    "lr_debug_draws": (c_i32, [C.c_uint64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "lr_debug_stream2": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library (built by literate_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise HipLibraryError("libliterate_hip.so not found at %s - run `python -m literate_amd.build` "
                              "(there is no CPU fallback)" % LIB)
    lib = C.CDLL(LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError("%s: %s" % (what, ERRORS.get(rc, rc)))
    raise HipLibraryError("%s: hipError_t %d" % (what, rc))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise HipLibraryError("no ROCm GPU visible: literate_amd has no CPU path")
    return torch


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_vp(t.data_ptr())


def stream_ptr(device=None):
    """torch's current HIP stream on `device` (None = the current device) as a void*."""
    import torch
    return c_vp(torch.cuda.current_stream(device).cuda_stream)


def launch(fn, device, *args):
    """Call an ABI entry point whose last argument is the stream: on `device`'s current torch stream, with `device`
    made current for the call (a launch on another device's stream is an error in HIP, and the library's own
    streams / events / graphs are created on the current device)."""
    import torch
    with torch.cuda.device(device):
        return fn(*args, stream_ptr(device))
