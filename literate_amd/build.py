"""Build libliterate_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libliterate_hip.so")
SOURCES = ["lr_stats.hip", "lr_loglik.hip", "lr_mcmc.hip", "lr_sim.hip"]
HEADERS = ["lr_device.h", "lr_chain.h", "lr_dd.h", "lr_scan.h", "lr_step.h", "lr_spec.h", "lr_internal.h", os.path.join("..", "..", "include", "literate_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_hip(force=False, verbose=False):
    """Compile the HIP sources into csrc/libliterate_hip.so; returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LR_EXTRA_FLAGS", "").split()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
