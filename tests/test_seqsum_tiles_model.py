"""CPU: tools/proto/seqsum_tiles_proto.py -- the model of k_long_chains (a long read's sequential float32 sum from
per-tile summaries: predicted binades, T0 / T1 for either entering parity, composition 64 tiles per step, tiles the
prediction missed evaluated from the true accumulator) against the plain loop, bit for bit.  The kernel itself is
checked against the oracle by tests/test_gpu_stat_long.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proto"))
import seqsum_proto as sq  # noqa: E402
import seqsum_tiles_proto as tp  # noqa: E402


def _reads():
    rs = np.random.RandomState(13)
    unit = np.float32(np.float32(1402.882324) / np.float32(8192.0))
    for n in (1, 255, 1025, 70000, 300001):
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.float32)
        pa = ((raw + np.float32(7)).astype(np.float32) * unit).astype(np.float32)
        yield "raw%d" % n, raw
        yield "pa%d" % n, pa
        m = np.float32(sq.seq_ref(pa) / np.float32(n))
        d = (pa - m).astype(np.float32)
        yield "dev%d" % n, (d * d).astype(np.float32)             # the deviation pass of stdvf
        yield "const%d" % n, np.full(n, 517, dtype=np.float32)    # every addition rounds the same way: predictions drift
        yield "ties%d" % n, np.full(n, 0.5, dtype=np.float32)      # every addition a tie once the sum is large
        yield "zeros%d" % n, np.zeros(n, dtype=np.float32)
        yield "negative%d" % n, (-raw).astype(np.float32)          # oriented by the sign of the total
        mixed = rs.randint(-2000, 2000, size=n).astype(np.float32)
        yield "mixed%d" % n, mixed                                 # negative terms: marked tiles, evaluated one by one


@pytest.mark.parametrize("waves", [64, 7])
def test_composed_tiles_equal_the_plain_loop(waves):
    composed = evaluated = 0
    for name, x in _reads():
        st = {}
        got = tp.compose_read(x, waves=waves, stats=st)
        ref = sq.seq_ref(x)
        assert int(sq.bits(got)) == int(sq.bits(ref)), (name, waves, st, got, ref)
        if name.startswith(("raw", "pa", "dev")) and x.size > 200000:
            # nanopore-like terms: all but the binade crossings (and the first tile) compose
            assert st.get("evaluated", 0) <= 24, (name, st)
            composed += st.get("composed", 0)
            evaluated += st.get("evaluated", 0)
    assert composed > 10 * evaluated
