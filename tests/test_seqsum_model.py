"""CPU: the numpy model of sigtk_amd/csrc/seqsum.h (tools/proto/seqsum_proto.py) against the plain sequential float32 loop.
The HIP implementation is compared with the oracle and with the lane-per-read kernels in tests/test_gpu_stat.py; this
test pins the ALGORITHM (surrogate starts, parity maps, binade crossings, native fallbacks) where no GPU is needed."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proto"))
import seqsum_proto as sp  # noqa: E402


def _same(a, b):
    return (np.isnan(a) and np.isnan(b)) or int(sp.bits(a)) == int(sp.bits(b))


def _cases():
    rs = np.random.RandomState(7)
    unit = np.float32(np.float32(1402.882324) / np.float32(8192.0))
    for n in (0, 1, 63, 64, 65, 1023, 1025, 5000, 40000):
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.float32)
        pa = ((raw + np.float32(7)).astype(np.float32) * unit).astype(np.float32)
        yield "raw%d" % n, raw
        yield "pa%d" % n, pa
        yield "negpa%d" % n, (-pa).astype(np.float32)
        if n:
            m = np.float32(sp.seq_ref(pa) / np.float32(n))
            d = (pa - m).astype(np.float32)
            yield "dev%d" % n, (d * d).astype(np.float32)
    yield "ties", np.full(70000, 333, dtype=np.float32)            # integer sum beyond 2^24: a tie at every step
    yield "mixed_sign", rs.randint(-2000, 2000, size=6000).astype(np.float32)
    z = np.zeros(5000, dtype=np.float32); z[3000] = 7; z[4000:] = 1
    yield "zeros_then_ones", z
    o = rs.normal(100, 10, size=20000).astype(np.float32); o[9000] = 3e7; o[15000] = 1e-30
    yield "outliers", o
    nn = rs.normal(100, 10, size=4000).astype(np.float32); nn[2500] = np.nan
    yield "nan", nn
    ii = rs.normal(100, 10, size=4000).astype(np.float32); ii[2500] = np.inf; ii[3000] = -np.inf
    yield "inf", ii
    yield "tiny", (rs.rand(3000) * 1e-35).astype(np.float32)
    yield "huge", (rs.rand(3000) * 1e35).astype(np.float32)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[c[0] for c in _cases()])
def test_wave_model_matches_sequential_sum(name, x):
    with np.errstate(all="ignore"):
        got, st = sp.seq_sum_wave(x)
        exp = sp.seq_ref(x)
    assert _same(got, exp), "%s: model %r sequential %r" % (name, got, exp)
    if name.startswith("pa") and x.size >= 40000:
        assert st.serial_tiles == 0 and st.crossings < 16   # the parallel path did the work
