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
    # a fixed number of batches: the same coverage on every box (a time box would not give that)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_parity.py"), "--batches", "120", "--seed", "3"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    stats = json.loads(p.stdout.strip().splitlines()[-1])
    assert stats["mismatches"] == [] and stats["batches"] == 120 and stats["reads"] > 100 and stats["fallback_reads"] > 0


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


def test_soak_regression_lane_local_rounding_in_the_flagged_pass(gpu, oracle):
    """seed 2024, batch 6003, read 1407 of tests/soak_parity.py (found after 0.7 million reads): a 270-sample read
    with one sample at 2e-5 pA (raw = -offset + 1e-4) fails the magnitude guard and goes to k_event_fallback; there
    the lazy pass' OWN running prefix rounded 22 indices later (its sum crossed a binade with the tiny term's low bits
    set) -- a place the reference's sequential scan, with another origin, adds exactly, so the repair list did not
    have it -- and a plateau of two equal t-statistics at 255 / 256 turned that last-bit difference into a missing
    boundary.  The flagged pass now checks its own additions (TwoSum residual)."""
    import numpy as np
    from sigtk_amd import api
    z = np.load(os.path.join(ROOT, "tests", "golden", "soak_seed2024_b6003_r1407.npz"))
    x = z["samples"].astype(np.int16)
    dig, off, rng = float(z["dig"][0]), float(z["off"][0]), float(z["rng"][0])
    for variant in (x, x[:262], np.concatenate([x, x[-5:]])):
        for rna in (0, 1):
            job = api.Job(0)
            job.stage([variant], np.array([dig]), np.array([off]), np.array([rng]), None)
            job.launch(api.TOOL_EVENT, rna=rna)
            res = job.wait()
            assert int(res["status"].n_fallback_reads) == 1
            g = res["events"][0]
            e = oracle.event_raw(variant, dig, off, rng, rna)
            assert g.start.size == e.start.size and np.array_equal(g.start.astype(np.uint64), e.start.astype(np.uint64))
            assert np.array_equal(g.mean.view(np.uint32), e.mean.view(np.uint32))
            assert np.array_equal(g.stdv.view(np.uint32), e.stdv.view(np.uint32))
            job.close()


def test_short_wave_vs_lane_soak(gpu):
    """a short run of tests/soak_wave_vs_lane.py: stat / jnn / prefix of the wave-per-read kernels against the
    lane-per-read kernels on batches of 1 000 - 6 000 ragged reads (longer runs are recorded under profiles/)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_wave_vs_lane.py"), "--batches", "40", "--seed", "9"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    stats = json.loads(p.stdout.strip().splitlines()[-1])
    assert stats["mismatches"] == [] and stats["batches"] == 40 and stats["reads"] > 5000
