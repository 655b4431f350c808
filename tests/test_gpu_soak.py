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


def test_soak_regression_stale_samples_before_a_read():
    """seed 77, batch 6919 of tests/soak_parity.py (found after 1.04 million reads): RNA parameters on a 1362-sample
    read with tiny pA values, staged into a recycled job right behind a neighbour whose samples, scaled with THIS
    read's offset, were 1e4 times larger.  The warm-up of the first chunk let them through the running window sums,
    where their squares left a rounding residue: one spurious boundary.  Replays the 60 batches up to that one
    through one job, as the soak did."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "soak_replay.py"), "--seed", "77", "--batch", "6919",
                        "--seq", "60"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "MISMATCH" not in p.stdout and p.stdout.strip().endswith("bad 0"), p.stdout[-2000:]
