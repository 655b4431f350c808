"""CPU: tools/proto/seqsum_segments_proto.py -- a sequential float32 sum over a read cut into segments that are
summarised independently (two surrogate starts per segment) and composed, against the plain loop, bit for bit.
The round-3 sketch of long reads on several wavefronts in `stat` / `jnn` / `prefix`; the kernel that came of it
(k_long_chains, round 4) summarises per 1024-term tile: tools/proto/seqsum_tiles_proto.py, tests/test_seqsum_tiles_model.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proto"))
import seqsum_segments_proto as sp  # noqa: E402


def _reads():
    rs = np.random.RandomState(11)
    unit = np.float32(np.float32(1402.882324) / np.float32(8192.0))
    for n in (1, 1000, 65536, 200001):
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.float32)
        pa = ((raw + np.float32(7)).astype(np.float32) * unit).astype(np.float32)
        yield "raw%d" % n, raw
        yield "pa%d" % n, pa
        m = np.float32(sp.seq(pa) / np.float32(n))
        d = (pa - m).astype(np.float32)
        yield "dev%d" % n, (d * d).astype(np.float32)            # the deviation pass of stdvf
        yield "ties%d" % n, np.full(n, 0.5, dtype=np.float32)     # every addition a tie once the sum is large
        yield "zeros%d" % n, np.zeros(n, dtype=np.float32)
        sp_ = pa.copy(); sp_[::97] = 0; sp_[5::1013] *= np.float32(4096)
        yield "spiky%d" % n, sp_


@pytest.mark.parametrize("seg", [1024, 4096, 16384])
def test_composed_segments_equal_the_plain_loop(seg):
    composed = 0
    for name, x in _reads():
        st = {}
        got = sp.compose_read(x, seg, st)
        ref = sp.seq(x)
        assert sp.bits(got) == sp.bits(ref), (name, seg, st)
        composed += st.get("composed", 0)
    assert composed > (50 if seg < 16384 else 30)   # (most segments of the long reads take the parallel route)


def test_serial_share_is_logarithmic():
    rs = np.random.RandomState(3)
    x = np.clip(np.rint(rs.normal(520, 75, size=400000)), 0, 4000).astype(np.float32)
    st = {}
    got = sp.compose_read(x, 4096, st)
    assert sp.bits(got) == sp.bits(sp.seq(x))
    G = st["composed"] + st["serial"]
    assert G == 98 and st["serial"] <= 4 + int(np.log2(G)) + 2, st
