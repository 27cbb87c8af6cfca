"""Build libliterate_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libliterate_hip.so")
SOURCES = ["lr_stats.hip", "lr_loglik.hip", "lr_mcmc.hip", "lr_spec.hip", "lr_pack.hip", "lr_sim.hip"]
HEADERS = ["lr_device.h", "lr_chain.h", "lr_dd.h", "lr_scan.h", "lr_step.h", "lr_spec.h", "lr_engine.h", "lr_internal.h",
           os.path.join("..", "..", "include", "literate_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_hip(force=False, verbose=False):
    """Compile the HIP sources (one hipcc per translation unit, in parallel) and link csrc/libliterate_hip.so;
    returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LR_EXTRA_FLAGS", "").split()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra
    objs = [os.path.splitext(s)[0] + ".o" for s in SOURCES]

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, cwd=CSRC, check=True)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as pool:
        list(pool.map(compile_one, zip(SOURCES, objs)))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
