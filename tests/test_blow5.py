"""CPU: the BLOW5 reader/writer helpers against the reference's bundled file."""
import os

import numpy as np

from sigtk_amd import blow5


def test_read_sp1(sp1):
    assert len(sp1.reads) == 100
    assert sum(r.raw.size for r in sp1.reads) == 472511
    assert sp1.record_press == 1 and sp1.signal_press == 1
    assert sp1.attr("experiment_type") == "genomic_dna" and sp1.attr("sequencing_kit") == "sqk-lsk109"
    r0 = sp1.reads[0]
    assert r0.read_id == "00011a60-dd92-4aad-be1d-59a33545ab1d" and r0.raw.size == 4710
    assert (r0.digitisation, r0.offset, r0.range) == (8192.0, 8.0, 1402.88232421875)


def test_svb_zd_roundtrip():
    rs = np.random.RandomState(0)
    for n in (0, 1, 3, 4, 5, 1000, 4097):
        x = rs.randint(-32768, 32767, size=n).astype(np.int16)
        assert np.array_equal(blow5.svb_zd_decode(blow5.svb_zd_encode(x)), x)
    x = (500 + rs.randint(-5, 5, size=777)).astype(np.int16)
    assert np.array_equal(blow5.svb_zd_decode(blow5.svb_zd_encode(x)), x)


def test_write_read_roundtrip(tmp_path, sp1):
    for rp, sp in ((0, 0), (1, 1), (1, 0), (0, 1)):
        p = str(tmp_path / ("t%d%d.blow5" % (rp, sp)))
        blow5.write_blow5(p, sp1.reads[:7], {"experiment_type": "rna", "sequencing_kit": "sqk-rna004"}, rp, sp)
        b = blow5.read_blow5(p)
        assert b.attr("experiment_type") == "rna"
        assert [r.read_id for r in b.reads] == [r.read_id for r in sp1.reads[:7]]
        for a, c in zip(b.reads, sp1.reads[:7]):
            assert np.array_equal(a.raw, c.raw) and a.offset == c.offset and a.range == c.range
