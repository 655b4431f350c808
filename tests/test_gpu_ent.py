"""GPU: `ent` -- histograms on the device + the reference-ordered finish on the host, against the
oracle (doubles compared bit for bit: same counts, same operations, same order)."""
import numpy as np
import pytest

from sigtk_amd import blow5

pytestmark = pytest.mark.gpu


def _run(gpu, reads, svb=False):
    n = len(reads)
    job = gpu.Job(0)
    sig = [blow5.svb_zd_encode(r) for r in reads] if svb else reads
    job.submit(gpu.TOOL_ENT, sig, np.full(n, 8192.0), np.zeros(n), np.full(n, 1400.0),
               counts=[r.size for r in reads] if svb else None)
    ent = job.wait()["ent"]
    job.close()
    return ent


def _check(oracle, reads, ent):
    for r, raw in enumerate(reads):
        exp = oracle.ent(raw)
        assert np.array_equal(ent[r].view(np.uint64), exp.view(np.uint64)), "read %d: %r vs oracle %r" % (r, ent[r], exp)


def test_ent_sp1_dna(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads]
    _check(oracle, reads, _run(gpu, reads))
    _check(oracle, reads, _run(gpu, reads, svb=True))


def test_ent_synthetic_and_ragged(gpu, oracle):
    lens = [1, 2, 3, 255, 256, 257, 1000, 30000, 100000, 100001]
    for kind in (0, 1):
        reads, _, _, _ = gpu.synth_reads_host(len(lens), lens, seed=31 + kind, kind=kind)
        _check(oracle, reads, _run(gpu, reads))


def test_ent_values_outside_the_windows(gpu, oracle):
    """negative samples, samples >= 8192 and jumps whose zigzag code is >= 4096 take the overflow lists"""
    rs = np.random.RandomState(9)
    reads = [rs.randint(-32768, 32767, size=50000).astype(np.int16),                    # everything overflows
             (500 + rs.randint(-40, 40, size=20000)).astype(np.int16),                  # nothing does
             np.where(rs.rand(7000) < 0.01, -5, 700).astype(np.int16),                  # a few negatives
             np.concatenate([np.full(3000, 100), np.full(3000, 9000), np.full(10, -32768)]).astype(np.int16),
             np.array([32767, -32768, 32767, -32768, 0], dtype=np.int16)]
    ent = _run(gpu, reads)
    _check(oracle, reads, ent)
    assert ent[0][0] > 14.0          # ~log2(50000 distinct-ish values): sanity that the lists were really used


def test_ent_empty_read_and_empty_batch(gpu):
    ent = _run(gpu, [np.zeros(0, dtype=np.int16), np.array([5], dtype=np.int16)])
    assert np.array_equal(ent, np.zeros((2, 3)))     # n = 0: defined as zeros; n = 1: one value, p = 1 -> 0, no deltas
    assert _run(gpu, []).shape == (0, 3)
