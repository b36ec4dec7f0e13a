"""Builds tests/native/libsolver_harness.so (g++, host only) on demand; returns the ctypes handle."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsolver_harness.so")


def load():
    src = os.path.join(HERE, "solver_harness.cpp")
    hdr = os.path.join(HERE, "..", "..", "getdist_amd", "csrc", "solvers.hpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", SO], check=True)
    return ctypes.CDLL(SO)
