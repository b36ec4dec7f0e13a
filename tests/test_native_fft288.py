"""The register codelets of getdist_amd/csrc/fft288.hpp (9-, 16-, 18-point transforms and their composition into the
288-point transform of the convolution's column kernel) against the definition of the discrete Fourier transform in long
double, on the host: the header compiles with g++ as it does with hipcc."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fft288_codelets_against_the_definition():
    src = os.path.join(HERE, "native", "fft288_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fft288_check")
        subprocess.run(["g++", "-O2", "-std=c++17", src, "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
