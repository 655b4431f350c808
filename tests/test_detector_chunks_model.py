"""CPU: tools/proto/detector_chunks_proto.py -- the rules of the speculative chunk / segment scheme of the event
detector (warm-up from the fresh state, verified hand-over, re-runs, chunks that stop at their range's end and leave a
pending peak to the chunk behind, seams between segments) as plain Python, against the sequential automaton of the
oracle on the statistics of the reference's fixture and of synthetic edge signals.  The HIP kernels are compared with
the oracle in tests/test_gpu_event*.py; this pins the scheme where no GPU is needed."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proto"))
import detector_chunks_proto as dp  # noqa: E402


def _tstats(oracle, raw, dig, off, rng, rna):
    pa = oracle.pa(raw, dig, off, rng)
    s, q = oracle.prefix_sums(pa)
    w1, w2 = (7, 14) if rna else (3, 6)
    return oracle.tstat(s, q, w1), oracle.tstat(s, q, w2)


def _signals(sp1):
    rs = np.random.RandomState(5)
    for r in sp1.reads[:6]:
        yield r.raw, r.digitisation, r.offset, r.range
    n = 3000
    yield np.full(n, 500, dtype=np.int16), 8192.0, 10.0, 1402.882324                                       # constant
    yield np.repeat(rs.randint(300, 700, size=n // 10), 10).astype(np.int16), 8192.0, 3.0, 1402.882324    # noiseless steps
    flat = np.full(n, 480, dtype=np.int16)
    for p in (100, 900, 903, 2500):
        flat[p:p + 40] = rs.randint(300, 700, size=40)
    yield flat, 8192.0, 6.0, 1402.882324                                                                   # bursts on a flat line
    yield rs.randint(380, 640, size=257).astype(np.int16), 8192.0, 2.0, -1402.882324                       # short, negative range


@pytest.mark.parametrize("rna", [0, 1])
@pytest.mark.parametrize("seg_len,lanes,lead", [(1 << 30, 8, 32), (1024, 8, 32), (512, 4, 16), (2048, 64, 16), (1024, 1, 0),
                                                (1024, 8, 0), (304, 3, 0), (1 << 30, 64, 0)])
def test_chunked_detector_equals_the_sequential_one(oracle, sp1, rna, seg_len, lanes, lead):
    stats = {}
    for raw, dig, off, rng in _signals(sp1):
        t1, t2 = _tstats(oracle, raw, dig, off, rng, rna)
        exp = sorted(int(p) for p in oracle.peaks(t1, t2, rna))
        got = dp.detect_read(t1, t2, rna, seg_len, lanes, lead, stats)
        assert got == exp, (raw.size, seg_len, lanes, lead, sorted(set(got) ^ set(exp))[:6])
    if lead <= 16 and lanes > 1:
        assert stats.get("rerun", 0) > 0          # short warm-ups: speculation does fail, and is repaired
    if lead == 0 and seg_len < 10000:
        assert stats.get("seam_rerun", 0) > 0     # no warm-up at all: the seams fail as well, spans are run again in order
