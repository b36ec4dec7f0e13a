"""Builds the host-only (g++) test harnesses on demand; returns ctypes handles."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsolver_harness.so")
SO_BATCH = os.path.join(HERE, "libbatch_harness.so")
CSRC = os.path.join(HERE, "..", "..", "getdist_amd", "csrc")


def _build(so, src, deps, extra=()):
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so] + list(extra),
                       check=True)
    return ctypes.CDLL(so)


def load():
    return _build(SO, os.path.join(HERE, "solver_harness.cpp"), [os.path.join(CSRC, "solvers.hpp")])


def load_batch():
    """batch2d.hpp (the plan and choreography of gd_density2d_batch) compiled for the host."""
    return _build(SO_BATCH, os.path.join(HERE, "batch_harness.cpp"),
                  [os.path.join(CSRC, "batch2d.hpp"), os.path.join(CSRC, "batch1d.hpp"), os.path.join(HERE, "..", "..", "include", "gdhip.h")], ["-pthread"])
