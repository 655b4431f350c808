"""GPU parity: the HIP `event` and `pa` paths against the oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RAGGED = [0, 1, 2, 5, 11, 12, 13, 14, 27, 28, 29, 63, 64, 65, 127, 128, 129, 199, 200, 250, 1000,
          4095, 4096, 4097, 8191, 8192, 8200, 20000, 33333]


def _check_events(oracle, reads, dig, off, rng, rna, got):
    for r, raw in enumerate(reads):
        exp = oracle.event_raw(raw, dig[r], off[r], rng[r], rna)
        g = got[r]
        assert g.start.size == exp.start.size, "read %d (n=%d): %d events, oracle %d" % (
            r, raw.size, g.start.size, exp.start.size)
        np.testing.assert_array_equal(g.start.astype(np.uint64), exp.start, err_msg="read %d start" % r)
        np.testing.assert_array_equal(g.length.astype(np.float32), exp.length, err_msg="read %d length" % r)
        # same arithmetic as the reference: bit-exact (the contract only asks for 1e-5 relative)
        np.testing.assert_array_equal(g.mean.view(np.uint32), exp.mean.view(np.uint32), err_msg="read %d mean" % r)
        np.testing.assert_array_equal(g.stdv.view(np.uint32), exp.stdv.view(np.uint32), err_msg="read %d stdv" % r)


def test_event_sp1_dna(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads]
    dig = np.array([r.digitisation for r in sp1.reads])
    off = np.array([r.offset for r in sp1.reads])
    rng = np.array([r.range for r in sp1.reads])
    got, status = gpu.event(reads, dig, off, rng, 0)
    _check_events(oracle, reads, dig, off, rng, 0, got)
    # the fixture contains a read with a |pA| = 0.34 sample, which must take the exact fallback
    assert status.n_fallback_reads >= 1
    assert status.n_capacity_overflow == 0


def test_event_sp1_rna_params(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads[:40]]
    dig = np.array([r.digitisation for r in sp1.reads[:40]])
    off = np.array([r.offset for r in sp1.reads[:40]])
    rng = np.array([r.range for r in sp1.reads[:40]])
    got, status = gpu.event(reads, dig, off, rng, 1)
    _check_events(oracle, reads, dig, off, rng, 1, got)


@pytest.mark.parametrize("kind,rna", [(0, 0), (1, 1), (0, 1), (1, 0)])
def test_event_synthetic_100k(gpu, oracle, kind, rna):
    reads, dig, off, rng = gpu.synth_reads_host(6, 100000, seed=1 + kind, kind=kind)
    got, status = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)
    assert status.n_capacity_overflow == 0


@pytest.mark.parametrize("rna", [0, 1])
def test_event_ragged(gpu, oracle, rna):
    reads, dig, off, rng = gpu.synth_reads_host(len(RAGGED), RAGGED, seed=7, kind=rna)
    got, status = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)
    assert got[0].start.size == 0  # empty read -> no events


def test_event_edge_signals(gpu, oracle):
    rs = np.random.RandomState(3)
    n = 6000
    const = np.full(n, 500, dtype=np.int16)                       # variance floor -> FLT_MIN
    steps = np.repeat(rs.randint(300, 700, size=n // 10), 10).astype(np.int16)  # noiseless steps
    noise = rs.randint(-32768, 32767, size=n).astype(np.int16)    # full int16 range
    tiny = (rs.randint(400, 600, size=n)).astype(np.int16)
    tiny[1234] = -9                                                # raw = -offset+2 -> tiny |pA| -> fallback
    zeros = np.zeros(n, dtype=np.int16)
    # a level change every 3 samples: the densest boundaries the detector can emit (> 512 per 2048 samples, so the
    # builder needs more than one round of boundary records per tile)
    dense = (np.repeat(np.tile([420, 610], n // 6 + 1), 3)[:n] + rs.randint(-2, 3, size=n)).astype(np.int16)
    reads = [const, steps, noise, tiny, zeros, dense]
    dig = np.full(6, 8192.0); off = np.array([10.0, 3.0, 0.0, 11.0, 0.0, 5.0]); rng = np.full(6, 1402.882324)
    for rna in (0, 1):
        got, status = gpu.event(reads, dig, off, rng, rna)
        _check_events(oracle, reads, dig, off, rng, rna, got)
    # negative range flips the sign of every pA value
    got, _ = gpu.event(reads[:2], dig[:2], off[:2], -rng[:2], 0)
    _check_events(oracle, reads[:2], dig[:2], off[:2], -rng[:2], 0, got)


def test_pa_matches_oracle(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads[:20]] + [np.zeros(0, dtype=np.int16), np.array([5], dtype=np.int16)]
    recs = sp1.reads[:20]
    dig = np.array([r.digitisation for r in recs] + [8192.0, 8192.0])
    off = np.array([r.offset for r in recs] + [3.0, -4.5])
    rng = np.array([r.range for r in recs] + [1400.0, 1467.61])
    got = gpu.pa(reads, dig, off, rng)
    for r, raw in enumerate(reads):
        exp = oracle.pa(raw, dig[r], off[r], rng[r])
        np.testing.assert_array_equal(got[r].view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("rna", [0, 1])
def test_event_guard_failing_reads_repair_path(gpu, oracle, rna):
    """Reads with tiny-|pA| samples make the reference's sequential double prefix sums round; the
    fallback kernel must reproduce exactly that (event-local repair, and the all-dirty mode when a
    read has more rounding events than the repair list holds)."""
    reads, dig, off, rng = gpu.synth_reads_host(8, 100000, seed=41, kind=rna)
    tiny = [int(1 - off[r]) for r in range(8)]   # raw + offset == 1 -> |pA| = 0.17
    for r, positions in enumerate([[50000], [1599], [1600], [99990], [7], [30000, 30001, 70000],
                                   list(range(1000, 99000, 1000)),          # ~98 tiny samples -> all-dirty mode
                                   []]):
        for p in positions:
            reads[r][p] = tiny[r]
    got, status = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)
    assert status.n_fallback_reads >= 7


@pytest.mark.parametrize("rna", [0, 1])
def test_event_ragged_lognormal(gpu, oracle, rna):
    """mixed read lengths (log-normal, sigma 0.8, as bench.py --ragged 0.8): short reads run on 16-sample chunk
    granularity with a short warm-up, long ones on the full-length layout, in one batch"""
    rs = np.random.RandomState(11)
    lens = (6000 * np.exp(rs.normal(-0.32, 0.8, size=48))).astype(np.int64).clip(1, 90000)
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens.tolist(), seed=9, kind=rna)
    got, status = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)
    assert status.n_capacity_overflow == 0
