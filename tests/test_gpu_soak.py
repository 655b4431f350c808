"""GPU: a short run of the randomised differential test (tests/soak_parity.py): all six subtools through the job
API against the oracle on random batches (random lengths / kinds / scalings, svb-zd and int16 input, adversarial
and guard-failing reads).  Longer runs are recorded under profiles/."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_soak(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_parity.py"), "--minutes", "0.4", "--seed", "3"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    stats = json.loads(p.stdout.strip().splitlines()[-1])
    assert stats["mismatches"] == [] and stats["reads"] > 100 and stats["fallback_reads"] > 0
