"""The persistent kernels' scan loops issue their group loads from inline asm and wait for them by hand
(csrc/lr_scan.h, lr_gload16_async): the compiler does not know the registers are in flight.  This test compiles the
translation units that use them to device assembly and checks that no instruction touches such a register between a
load and the next vmcnt wait, on any path (literate_amd/check_async_loads.py).  No GPU needed: hipcc cross-compiles."""
import os
import shutil
from concurrent.futures import ThreadPoolExecutor

import pytest

from literate_amd import build, check_async_loads

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_no_read_of_a_register_in_flight(tmp_path):
    units = ["lr_mcmc.hip", "lr_spec.hip"]
    outs = [str(tmp_path / (u + ".s")) for u in units]
    with ThreadPoolExecutor(max_workers=2) as pool:
        list(pool.map(lambda p: build.device_asm(*p), zip(units, outs)))
    n_loads, bad = check_async_loads.check(outs, verbose=False)
    assert n_loads > 100, "the hand-placed loads were not found: has the asm changed?"
    assert bad == 0


def test_the_checker_sees_a_violation(tmp_path):
    good = """
.LBB0_1:
	global_load_dwordx4 v[4:7], v8, s[2:3]
	v_add_f64 v[10:11], v[12:13], v[14:15]
	s_cbranch_scc1 .LBB0_2
	s_waitcnt vmcnt(0)
	v_and_b32_e32 v9, 15, v4
.LBB0_2:
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v1, v5
	s_endpgm
"""
    bad = good.replace("	s_cbranch_scc1 .LBB0_2\n	s_waitcnt vmcnt(0)\n", "	s_cbranch_scc1 .LBB0_2\n	v_mov_b32_e32 v20, v6\n	s_waitcnt vmcnt(0)\n")
    bad2 = good.replace(".LBB0_2:\n	s_waitcnt vmcnt(0)\n", ".LBB0_2:\n")       # the taken branch reads v5 with no wait
    for name, text, want in (("good", good, 0), ("bad", bad, 1), ("bad2", bad2, 1)):
        p = tmp_path / (name + ".s")
        p.write_text(text)
        n, v = check_async_loads.check([str(p)], verbose=False)
        assert n == 1 and v == want, (name, n, v)
