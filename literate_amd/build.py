"""Build libliterate_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libliterate_hip.so")
SOURCES = ["lr_stats.hip", "lr_loglik.hip", "lr_mcmc.hip", "lr_spec.hip", "lr_pack.hip", "lr_sim.hip"]
HEADERS = ["lr_device.h", "lr_math.h", "lr_chain.h", "lr_dd.h", "lr_scan.h", "lr_step.h", "lr_spec.h", "lr_engine.h", "lr_internal.h",
           os.path.join("..", "..", "include", "literate_hip.h")]
# Per translation unit.  The speculative kernel's loop body is ~8000 instructions at a 168-VGPR budget: machine LICM
# hoists every literal of the inlined log/exp polynomials out of it and the allocator then spills them (592 bytes of
# scratch, reloaded inside the candidate build); without the pass the kernel keeps 128 bytes and the few-chain shards
# run 8-17 % faster (cfg3 4.7 -> 4.0 us per iteration).  lr_mcmc.hip: the persistent kernels' stepper waves run their
# whole launch inside one function (lr_persist4_steppers) with the chain step inlined in its loop: with the pass that
# function reloads ~480 hoisted values from scratch per step, without it a dozen.
TU_FLAGS = {"lr_spec.hip": os.environ.get("LR_SPEC_FLAGS", "-mllvm -disable-machine-licm").split(),
            "lr_mcmc.hip": os.environ.get("LR_MCMC_FLAGS", "-mllvm -disable-machine-licm").split()}


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
        cmd = [hipcc] + flags + TU_FLAGS.get(src, []) + ["-c", src, "-o", obj]
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


def device_asm(src, out_path):
    """The device assembly of one translation unit, compiled with the flags of the library build (for
    literate_amd.check_async_loads)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LR_EXTRA_FLAGS", "").split()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17"] + extra + TU_FLAGS.get(src, []) + \
          ["-S", "--cuda-device-only", src, "-o", out_path]
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    return out_path


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
