"""
Build libbluest_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU is needed to build.

    python -m bluest_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["runtime.hip", "mirrors.hip", "plan.hip", "spg.hip", "intproj.hip", "xchg.hip", "newton.hip", "matfree.hip"]     # translation units of libbluest_hip.so
HEADERS = ["common.hpp", "solve.hpp", "plan.hpp", "spg_state.hpp"]
OBJDIR = os.path.join(CSRC, "_build")
HDR = os.path.join(ROOT, "include", "bluest_hip.h")
LIB = os.path.join(HERE, "libbluest_hip.so")
ARCH = "gfx950"


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _inputs():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [HDR]


STAMP = os.path.join(OBJDIR, "hipcc_flags.txt")      # the extra flags the library was built with: an experiment build is never "current"


def _extra_flags():
    return os.environ.get("BLUEST_EXTRA_HIPCC_FLAGS", "").split()


def up_to_date():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in _inputs()):
        return False
    try:
        built_with = open(STAMP).read().split()
    except OSError:
        built_with = []
    return built_with == _extra_flags()


SHIM_SRC = os.path.join(HERE, "csrc", "cmisc_shim.cpp")


def shim_path():
    import sysconfig
    return os.path.join(HERE, "shim", "_cmisc_bluest" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_shim(force=False, verbose=False):
    """pybind11 module `_cmisc_bluest` (same names as the reference's native module) over libbluest_hip.so"""
    out = shim_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(SHIM_SRC), os.path.getmtime(HDR)):
        return out
    import pybind11
    import sysconfig
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-O2", "-shared", "-std=c++17", "-fPIC", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.join(ROOT, "include"), SHIM_SRC, "-o", out, "-L" + HERE, "-lbluest_hip", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and up_to_date():
        build_shim(force=False, verbose=verbose)
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-pthread", "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC] + _extra_flags()

    def compile_one(name):
        obj = os.path.join(OBJDIR, os.path.splitext(name)[0] + ".o")
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:   # the units compile in parallel
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(" ".join(_extra_flags()))
    build_shim(force=True, verbose=verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
