"""Builds the native parts of the package in-tree (hipcc cross-compiles gfx950 without a GPU).

  csrc/libgmgcoulomb.so  -- HIP kernels + the C-ABI of include/gmg_coulomb.h   (hipcc, gfx950)
  csrc/host/libstep50host.so / csrc/host/step50_mi355x -- host-side C++ mirror of the
      reference's LaplaceProblem (g++), which links against the C-ABI only.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(CSRC, "host")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")

LIB_DEVICE = os.path.join(CSRC, "libgmgcoulomb.so")
LIB_HOST = os.path.join(HOST, "libstep50host.so")
EXE_HOST = os.path.join(HOST, "step50_mi355x")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def build_device(force=False, verbose=False):
    srcs = _sources(CSRC, (".hip", ".hpp")) + [os.path.join(HERE, "..", "include", "gmg_coulomb.h")]
    if force or _newer(LIB_DEVICE, srcs):
        # -fno-jump-tables: the per-turn shape dispatch of the SSOR sweep as compares, not as a table read from memory
        cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-fno-jump-tables", "-std=c++17",
               "-Wall", "-Wno-unused-function", "-o", LIB_DEVICE, os.path.join(CSRC, "gmg_coulomb.hip"),
               "-pthread", "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_DEVICE


def build_host(force=False, verbose=False):
    if not os.path.isdir(HOST) or not _sources(HOST, (".cc",)):
        return None
    srcs = _sources(HOST, (".cc", ".h", ".inc")) + [os.path.join(HERE, "..", "include", "gmg_coulomb.h")]
    lib_srcs = [s for s in _sources(HOST, (".cc",)) if not s.endswith("main.cc")]
    if force or _newer(LIB_HOST, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-fopenmp",
               "-I" + os.path.join(HERE, "..", "include"), "-o", LIB_HOST] + lib_srcs + \
              ["-L" + CSRC, "-lgmgcoulomb", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    main_cc = os.path.join(HOST, "main.cc")
    if os.path.exists(main_cc) and (force or _newer(EXE_HOST, srcs)):
        cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-I" + os.path.join(HERE, "..", "include"), "-o", EXE_HOST, main_cc,
               "-L" + HOST, "-lstep50host", "-L" + CSRC, "-lgmgcoulomb", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..",
               "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_HOST


def build_all(force=False, verbose=False):
    build_device(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv, verbose=True)
