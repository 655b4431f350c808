"""CPU: the fast arithmetic used by the event kernels (sigtk_amd/csrc/tstat_math.h: constant
division by FMA correction, certified-rsqrt t-statistic tail) is bit-identical to plain IEEE
division / sqrt.  Runs oracle/verify_math.cpp in its sampled ('quick') mode; the exhaustive mode
(every float, 4e9 doubles; ~10 min on 8 cores) was run during development: ALL EXACT."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_fast_math_is_exact(tmp_path):
    exe = str(tmp_path / "verify_math")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-o", exe,
                           os.path.join(ROOT, "oracle", "verify_math.cpp")])
    p = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ALL EXACT" in p.stdout, p.stdout[-2000:]
