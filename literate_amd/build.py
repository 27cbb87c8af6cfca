"""Build libliterate_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The build also VERIFIES what it ships: the persistent kernels' scan loops issue their group loads from inline asm and
wait for them by hand (csrc/lr_scan.h), which is only safe as long as the compiler places no read of those registers
before the wait.  Every translation unit is therefore compiled with -save-temps, and the device assembly of the very
code objects that are linked is walked by literate_amd.check_async_loads; a violation - or the hand-placed loads not
being found where they are expected - fails the build.  The compiler's version string is recorded beside the library
(libliterate_hip.build.json), so a library built by another hipcc is recognisable."""
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libliterate_hip.so")
BUILD_INFO = os.path.join(CSRC, "libliterate_hip.build.json")
OBJ_DIR = os.path.join(CSRC, "_build")
SOURCES = ["lr_stats.hip", "lr_loglik.hip", "lr_mcmc.hip", "lr_spec.hip", "lr_stream.hip", "lr_packscan.hip", "lr_pack.hip", "lr_sim.hip", "lr_format.hip"]
HEADERS = ["lr_device.h", "lr_math.h", "lr_chain.h", "lr_dd.h", "lr_scan.h", "lr_step.h", "lr_spec.h", "lr_engine.h", "lr_internal.h",
           os.path.join("..", "..", "include", "literate_hip.h")]
# Per translation unit.  The speculative kernel's loop body is ~8000 instructions at a 168-VGPR budget: machine LICM
# hoists every literal of the inlined log/exp polynomials out of it and the allocator then spills them (592 bytes of
# scratch, reloaded inside the candidate build); without the pass the kernel keeps 128 bytes and the few-chain shards
# run 8-17 % faster (cfg3 4.7 -> 4.0 us per iteration).  lr_mcmc.hip: the persistent kernels' stepper waves run their
# whole launch inside one function (lr_persist4_steppers) with the chain step inlined in its loop: with the pass that
# function reloads ~480 hoisted values from scratch per step, without it a dozen.
TU_FLAGS = {"lr_spec.hip": os.environ.get("LR_SPEC_FLAGS", "-mllvm -disable-machine-licm").split(),
            "lr_mcmc.hip": os.environ.get("LR_MCMC_FLAGS", "-mllvm -disable-machine-licm").split(),
            # (the resident streaming kernel: its stepper waves' loop holds the chain step twice - 486 VGPRs with the pass,
            # where four blocks per CU allow 128)
            "lr_stream.hip": os.environ.get("LR_STREAM_FLAGS", "-mllvm -disable-machine-licm").split()}
# translation units whose device assembly is checked, and the least number of saddr-form 16-byte loads the checker must
# find there (lr_mcmc.hip holds the hand-placed ones; lr_spec.hip includes the same scan header but its slices use plain
# loads - it is checked all the same)
ASYNC_UNITS = {"lr_mcmc.hip": 60, "lr_spec.hip": 0, "lr_packscan.hip": 16}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def hipcc_version():
    try:
        return subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError) as ex:
        return "unknown (%s)" % type(ex).__name__


def device_asm_path(src):
    """Where -save-temps leaves the gfx950 assembly of one translation unit of the library build."""
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def build_hip(force=False, verbose=False):
    """Compile the HIP sources (one hipcc per translation unit, in parallel), check the hand-placed loads in the
    device assembly of exactly those compiles, link csrc/libliterate_hip.so; returns the library path."""
    if not force and not _stale():
        return LIB
    from . import check_async_loads
    hipcc = _hipcc()
    extra = os.environ.get("LR_EXTRA_FLAGS", "").split()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj"] + extra
    shutil.rmtree(OBJ_DIR, ignore_errors=True)
    os.makedirs(OBJ_DIR)
    # a build that fails - a compile error, a hand-placed load the checker objects to - must not leave the library of an
    # EARLIER build behind: it would travel to the GPU box and be measured as if it were this source
    for stale in (LIB, BUILD_INFO):
        if os.path.exists(stale):
            os.remove(stale)
    objs = [os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o") for s in SOURCES]

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc] + flags + TU_FLAGS.get(src, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, cwd=CSRC, check=True)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as pool:
        list(pool.map(compile_one, zip(SOURCES, objs)))
    checked = {}
    for src, least in ASYNC_UNITS.items():
        n_loads, bad = check_async_loads.check([device_asm_path(src)], verbose=verbose)
        checked[src] = n_loads
        if bad:
            raise RuntimeError("%s: %d reads of registers a hand-placed load still has in flight (see above): this "
                               "compiler / flag set cannot build the scan loops safely" % (src, bad))
        if n_loads < least:
            raise RuntimeError("%s: only %d hand-placed loads found in the device assembly (expected >= %d): has the "
                               "asm of lr_gload16_async changed, or the checker's pattern?" % (src, n_loads, least))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, cwd=CSRC, check=True)
    with open(BUILD_INFO, "w") as f:
        json.dump({"hipcc": hipcc, "hipcc_version": hipcc_version(), "flags": flags, "tu_flags": TU_FLAGS,
                   "async_loads_checked": checked, "async_load_violations": 0}, f, indent=1)
    # the temporaries are large (preprocessed sources, bitcode): keep the objects' directory out of the snapshot
    for name in os.listdir(OBJ_DIR):
        if not name.endswith(".s") or "host" in name:
            os.remove(os.path.join(OBJ_DIR, name))
    return LIB


def device_asm(src, out_path):
    """The device assembly of one translation unit, compiled with the flags of the library build (for
    literate_amd.check_async_loads)."""
    extra = os.environ.get("LR_EXTRA_FLAGS", "").split()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17"] + extra + TU_FLAGS.get(src, []) + \
          ["-S", "--cuda-device-only", src, "-o", out_path]
    subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    return out_path


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
