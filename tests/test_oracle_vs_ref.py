"""CPU: the oracle restatement against the REAL reference (oracle/_ref/libsigtk_ref.so, compiled
from /root/reference by oracle/Makefile), function by function on seeded inputs.  Skipped when the
reference library has not been built."""
import numpy as np
import pytest

from sigtk_amd import api


@pytest.fixture(scope="module")
def ref(reflib):
    if reflib is None:
        pytest.skip("oracle/_ref/libsigtk_ref.so not built (needs /root/reference)")
    return reflib


def _random_reads(seed):
    rs = np.random.RandomState(seed)
    reads, dig, off, rng = api.synth_reads_host(6, [250, 1000, 5000, 20000, 50000, 100000], seed, seed % 2)
    reads = list(reads)
    reads.append(rs.randint(-3000, 3000, size=3000).astype(np.int16))
    reads.append(np.repeat(rs.randint(300, 700, size=400), 10).astype(np.int16))
    dig = np.concatenate([dig, [8192.0, 2048.0]])
    off = np.concatenate([off, [12.0, -3.0]])
    rng = np.concatenate([rng, [1402.882324, -748.5]])
    return reads, dig, off, rng


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pa_event_stat_bitwise(oracle, ref, seed):
    reads, dig, off, rng = _random_reads(seed)
    for r, raw in enumerate(reads):
        assert np.array_equal(oracle.pa(raw, dig[r], off[r], rng[r]).view(np.uint32),
                              ref.pa(raw, dig[r], off[r], rng[r]).view(np.uint32))
        for rna in (0, 1):
            a = oracle.event_raw(raw, dig[r], off[r], rng[r], rna)
            b = ref.event_raw(raw, dig[r], off[r], rng[r], rna)
            assert np.array_equal(a.start, b.start) and np.array_equal(a.length, b.length)
            assert np.array_equal(a.mean.view(np.uint32), b.mean.view(np.uint32))
            assert np.array_equal(a.stdv.view(np.uint32), b.stdv.view(np.uint32))
        sa = oracle.stat(raw, dig[r], off[r], rng[r])
        sb = ref.stat(raw, dig[r], off[r], rng[r])
        assert sa[4] == sb[4]
        for k in (0, 1, 2, 3, 5):
            assert np.float32(sa[k]).view(np.uint32) == np.float32(sb[k]).view(np.uint32)


@pytest.mark.parametrize("seed", [4, 5])
def test_jnn_adaptor_polya(oracle, ref, seed):
    reads, dig, off, rng = api.synth_reads_host(5, [1500, 2500, 30000, 100000, 100000], seed, 1)
    for r, raw in enumerate(reads):
        for rna in (0, 1):
            ax, ay = oracle.jnn_raw(raw, rna)
            bx, by = ref.jnn_raw(raw, rna)
            assert np.array_equal(ax, bx) and np.array_equal(ay, by)
        for pore in (0, 2):
            assert oracle.find_adaptor(raw, pore) == ref.find_adaptor(raw, pore)
        pa = oracle.pa(raw, dig[r], off[r], rng[r])
        for top, bot in ((120.0, 80.0), (110.5, 90.25)):
            assert oracle.find_polya(pa, top, bot, 0) == ref.find_polya(pa, top, bot, 0)


def test_soak_regression_fixtures_are_pinned(oracle, ref):
    """the reads the round-2 soaks found GPU bugs on (tests/golden/soak_*.npz): what the GPU tests compare against --
    the oracle -- equals the real reference on them (event, stat and jnn)."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name in ("soak_seed2024_b6003_r1407.npz", "soak_wvl_seed5_b431_r1525.npz", "soak_wvl_seed5_b797_r3354.npz"):
        z = np.load(os.path.join(gold, name))
        raw = z["samples"].astype(np.int16)
        dig, off, rng = float(np.ravel(z["dig"])[0]), float(np.ravel(z["off"])[0]), float(np.ravel(z["rng"])[0])
        for rna in (0, 1):
            a = oracle.event_raw(raw, dig, off, rng, rna)
            b = ref.event_raw(raw, dig, off, rng, rna)
            assert np.array_equal(a.start, b.start) and np.array_equal(a.mean.view(np.uint32), b.mean.view(np.uint32))
            ax, ay = oracle.jnn_raw(raw, rna)
            bx, by = ref.jnn_raw(raw, rna)
            assert np.array_equal(ax, bx) and np.array_equal(ay, by)
        sa, sb = oracle.stat(raw, dig, off, rng), ref.stat(raw, dig, off, rng)
        assert sa[4] == sb[4] and all(np.float32(sa[k]).view(np.uint32) == np.float32(sb[k]).view(np.uint32) for k in (0, 1, 2, 3, 5))
