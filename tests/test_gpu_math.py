"""GPU: the hardware assumptions of the certified fast arithmetic (sigtk_amd/csrc/tstat_math.h): v_rsq_f32 within
2^-22 relative on every positive normal float, v_rcp_f32 within 2 ulp on the range event lengths live in.
(tools/rsq_check.hip, exhaustive; built by __graft_entry__.build().)"""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rsq_and_rcp_accuracy(gpu):
    exe = os.path.join(ROOT, "tools", "rsq_check")
    assert os.path.exists(exe), "tools/rsq_check not built (run __graft_entry__.build())"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    m = re.search(r"v_rsq_f32: max relative error ([0-9.e+-]+)", p.stdout)
    assert m and float(m.group(1)) < 2.0 ** -22
    m = re.search(r"v_rcp_f32: max relative error ([0-9.e+-]+)", p.stdout)
    assert m and float(m.group(1)) < 2.0 ** -22
