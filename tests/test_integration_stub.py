"""The ctypes stubs INTEGRATION.md section 2 shows a LiteRate maintainer, exercised as written: raw ctypes on
libliterate_hip.so (no literate_amd import), torch only for device memory, replacing calc_likelihood (LRF:430-431) and
the binning loop (LRF:519-523)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(golden_dir):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X: no ROCm device visible")
    return np.load(os.path.join(golden_dir, "binning_lik.npz"))


def test_integration_md_stub_matches_reference_likelihoods(G):
    import torch
    _lr = ctypes.CDLL(os.path.join(ROOT, "literate_amd", "csrc", "libliterate_hip.so"))
    _lr.lr_bd_loglik_workspace_bytes.restype = ctypes.c_int64
    name = "example_TBP"
    ts, te = G[name + "/ts"], G[name + "/te"]
    start_time, end_time = G[name + "/start_end"]
    n_bins = len(G[name + "/sp"])
    br_length_bin = G[name + "/br"]
    for model_BDI in (0, 2):
        _ts = torch.as_tensor(ts, dtype=torch.float64, device="cuda")
        _te = torch.as_tensor(te, dtype=torch.float64, device="cuda")
        _br = torch.as_tensor(br_length_bin, dtype=torch.float64, device="cuda")
        _out = torch.empty(1, dtype=torch.float64, device="cuda")
        _ws = torch.empty(_lr.lr_bd_loglik_workspace_bytes(ctypes.c_int64(len(ts)), n_bins, 1, model_BDI),
                          dtype=torch.uint8, device="cuda")
        P = lambda t: ctypes.c_void_p(t.data_ptr())

        def calc_likelihood(L_acc_vec, M_acc_vec):          # same signature as LRF:137 / LRF:150
            lam = torch.as_tensor(L_acc_vec, dtype=torch.float64, device="cuda")
            mu = torch.as_tensor(M_acc_vec, dtype=torch.float64, device="cuda")
            rc = _lr.lr_bd_loglik_batch(P(_ts), P(_te), ctypes.c_int64(len(ts)), ctypes.c_double(int(start_time)),
                                        n_bins, P(lam), P(mu), 1, model_BDI, P(_br), ctypes.c_double(end_time),
                                        P(_out), P(_ws), ctypes.c_int64(_ws.numel()), None)
            if rc:
                raise RuntimeError("lr_bd_loglik_batch rc=%d" % rc)
            return _out.item()

        # the reference's own worked state: L=[.6,.2], M=[.15,.19], shifts at 4.55 / 16.682
        L = np.where(np.arange(n_bins) < 4, .6, .2)
        M = np.where(np.arange(n_bins) < 16, .15, .19)
        want = 15.824528451812753 if model_BDI == 0 else -352.6785362157869
        assert calc_likelihood(L, M) == pytest.approx(want, rel=1e-9)


def test_integration_md_binning_stub_matches_reference_statistics(G):
    """The one-pass binning call of INTEGRATION.md section 2, as written, against the statistics the reference's own loop
    (LRF:519-523) produced for the shipped datasets (tests/golden/binning_lik.npz): bit for bit."""
    import torch
    _lr = ctypes.CDLL(os.path.join(ROOT, "literate_amd", "csrc", "libliterate_hip.so"))
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for name in ("example_TBP", "example_TAD", "metal_bands", "simulated"):
        ts, te = G[name + "/ts"], G[name + "/te"]
        _ts = torch.as_tensor(ts, dtype=torch.float64, device="cuda")
        _te = torch.as_tensor(te, dtype=torch.float64, device="cuda")
        _lr.lr_bin_unit_events_workspace_bytes.restype = ctypes.c_int64
        t0 = int(min(ts)); n_bins = int(max(te)) - t0                      # the windows [i, i + 1], i in range(int(min ts), int(max te))
        _sp = torch.empty(n_bins, dtype=torch.int64, device="cuda"); _ex = torch.empty_like(_sp)
        _brl = torch.empty(n_bins, dtype=torch.float64, device="cuda")
        _wsb = torch.empty(_lr.lr_bin_unit_events_workspace_bytes(ctypes.c_int64(len(ts)), n_bins), dtype=torch.uint8, device="cuda")
        rc = _lr.lr_bin_unit_events(P(_ts), P(_te), ctypes.c_int64(len(ts)), ctypes.c_double(t0), n_bins, P(_sp), P(_ex), P(_brl),
                                    P(_wsb), ctypes.c_int64(_wsb.numel()), None)
        sp_events_bin, ex_events_bin, br_length_bin = _sp.cpu().numpy(), _ex.cpu().numpy(), _brl.cpu().numpy()
        assert rc == 0 and n_bins == len(G[name + "/sp"])
        assert np.array_equal(sp_events_bin, G[name + "/sp"]) and np.array_equal(ex_events_bin, G[name + "/ex"])
        assert np.array_equal(br_length_bin, G[name + "/br"])
