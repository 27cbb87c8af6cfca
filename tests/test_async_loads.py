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
    """(build_hip() runs the same check on the assembly of the code objects it links and fails the build on a violation;
    here the assembly the last build left is checked again, or compiled if there is none.)"""
    units = ["lr_mcmc.hip", "lr_spec.hip"]
    outs = [build.device_asm_path(u) for u in units]
    if not all(os.path.exists(o) for o in outs):
        outs = [str(tmp_path / (u + ".s")) for u in units]
        with ThreadPoolExecutor(max_workers=2) as pool:
            list(pool.map(lambda p: build.device_asm(*p), zip(units, outs)))
    n_loads, bad = check_async_loads.check(outs, verbose=False)
    assert n_loads > 100, "the hand-placed loads were not found: has the asm changed?"
    assert bad == 0


def test_build_records_compiler_and_check(tmp_path):
    import json
    if not os.path.exists(build.BUILD_INFO):
        pytest.skip("library not built by this tree's build.py")
    info = json.load(open(build.BUILD_INFO))
    assert "clang version" in info["hipcc_version"] and info["async_load_violations"] == 0
    assert info["async_loads_checked"]["lr_mcmc.hip"] >= build.ASYNC_UNITS["lr_mcmc.hip"]


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
    # a vmcnt(1) wait right behind the load leaves it in flight (one outstanding operation is allowed: this one); only a
    # wait whose count is covered by the vector-memory instructions issued after the load makes it land
    bad3 = good.replace("	v_add_f64 v[10:11], v[12:13], v[14:15]\n", "	s_waitcnt vmcnt(1)\n	v_mov_b32_e32 v21, v7\n")
    good2 = good.replace("	v_add_f64 v[10:11], v[12:13], v[14:15]\n",
                         "	global_load_dwordx4 v[30:33], v8, s[4:5]\n	s_waitcnt vmcnt(1)\n	v_mov_b32_e32 v21, v7\n")
    # a FLAT operation behind the load may complete out of order with it (it can be served by LDS): it must not count as
    # an operation "behind" the load, so vmcnt(1) does not land the load here - the same lines with a global load do
    bad4 = good2.replace("global_load_dwordx4 v[30:33], v8, s[4:5]", "flat_load_dwordx4 v[30:33], v[40:41]")
    for name, text, want, n_want in (("good", good, 0, 1), ("bad", bad, 1, 1), ("bad2", bad2, 1, 1), ("bad3", bad3, 1, 1),
                                     ("good2", good2, 0, 2), ("bad4", bad4, 1, 1)):
        p = tmp_path / (name + ".s")
        p.write_text(text)
        n, v = check_async_loads.check([str(p)], verbose=False)
        assert n == n_want and v == want, (name, n, v)
