"""Build the in-tree native libraries.

  gaussianvi_amd/libgvi_hip.so   HIP kernels + C-ABI (include/gvi_hip.h), hipcc --offload-arch=gfx950

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgvi_hip.so")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build gaussianvi_amd/libgvi_hip.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def lib_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    srcs.append(os.path.join(ROOT, "include", "gvi_hip.h"))
    return srcs


def build_lib(force: bool = False, verbose: bool = False, out: str | None = None, defines: tuple = ()) -> str:
    """out / defines: an A/B build beside the in-tree library (tools/build_variant.py)."""
    LIB = out or globals()["LIB"]
    if not force and not _stale(LIB, lib_sources()):
        return LIB
    tmp = LIB + f".tmp{os.getpid()}"          # written aside and renamed: a concurrent reader never sees a partial file
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"),
           os.path.join(CSRC, "gvi_hip.hip"), "-x", "hip", os.path.join(CSRC, "spgh.cpp"), os.path.join(CSRC, "table_io.cpp"),
           "-o", tmp]
    for define in list(defines) + os.environ.get("GVI_BUILD_DEFINES", "").split():      # profiling builds, e.g. GVI_FUSED_TIMING
        cmd.insert(1, "-D" + define)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        sys.stderr.write(r.stderr)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-8000:])
    os.replace(tmp, LIB)
    return LIB


def build_examples(force: bool = False) -> str:
    """C++ drivers over the host shim (include/gvi/gvi_host.hpp), linked against the C-ABI library."""
    build_lib()
    out_dir = os.path.join(ROOT, "examples", "bin")
    os.makedirs(out_dir, exist_ok=True)
    first = None
    for name in ("1d_example", "1d_example_prox", "planar_example", "factorwise_example", "ltv_chain_example"):
        src = os.path.join(ROOT, "examples", name + ".cpp")
        exe = os.path.join(out_dir, name)
        deps = [src, os.path.join(ROOT, "include", "gvi", "gvi_host.hpp"), os.path.join(ROOT, "include", "gvi", "factorized_opts_LTV.hpp"),
                os.path.join(ROOT, "include", "gvi_hip.h")]
        if force or _stale(exe, deps):
            cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src,
                   "-L", HERE, "-lgvi_hip", "-Wl,-rpath," + HERE, "-o", exe]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("g++ failed:\n" + " ".join(cmd) + "\n" + r.stderr[-4000:])
        first = first or exe
    exe = first
    return exe


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_examples(force="--force" in sys.argv))
