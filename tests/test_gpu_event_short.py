"""GPU parity: several short reads per wavefront (k_event_multi), against the oracle (bit-exact).

Reads shorter than 16 384 samples (65 536 with RNA parameters) in a large batch get `lanes` lanes each instead of a
wavefront of their own
(the lanes_per_short_read option forces the number here; by default it is chosen per batch and small batches keep 64).
"""
import numpy as np
import pytest

from test_gpu_event import _check_events

pytestmark = pytest.mark.gpu

SHORT = [0, 1, 2, 5, 11, 12, 13, 14, 27, 28, 29, 63, 64, 65, 127, 128, 129, 199, 200, 250, 1000, 4095, 4096, 4097, 8191,
         8192, 8200, 12000, 16383, 333, 77, 5000, 5001, 4999, 6, 16000]


@pytest.fixture()
def lanes_cfg(gpu):
    def f(lanes, lead=0):
        gpu.event_configure_short(lanes)
        gpu.event_configure(0, 0, lead)
    yield f
    gpu.event_configure_short(0)
    gpu.event_configure(0, 0, 0)


@pytest.mark.parametrize("rna", [0, 1])
@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32])
def test_short_reads_share_a_wavefront(gpu, oracle, lanes_cfg, rna, lanes):
    reads, dig, off, rng = gpu.synth_reads_host(len(SHORT), SHORT, seed=17 + rna, kind=rna)
    lanes_cfg(lanes)
    got, st = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)
    assert got[0].start.size == 0 and st.n_capacity_overflow == 0
    assert st.n_events_total == sum(g.start.size for g in got)
    lanes_cfg(lanes, 16)   # a 16-sample warm-up: speculation fails, lanes are re-run inside their read's group
    got, st = gpu.event(reads, dig, off, rng, rna)
    _check_events(oracle, reads, dig, off, rng, rna, got)


@pytest.mark.parametrize("rna", [0, 1])
def test_fixture_reads_packed(gpu, oracle, sp1, lanes_cfg, rna):
    recs = sp1.reads
    reads = [r.raw for r in recs]
    dig = np.array([r.digitisation for r in recs]); off = np.array([r.offset for r in recs])
    rng = np.array([r.range for r in recs])
    for lanes in (4, 16):
        lanes_cfg(lanes)
        got, st = gpu.event(reads, dig, off, rng, rna)
        _check_events(oracle, reads, dig, off, rng, rna, got)
        assert st.n_fallback_reads >= 1   # (the fixture's read with a |pA| = 0.34 sample)


@pytest.mark.parametrize("rna", [0, 1])
def test_sorted_batch_short_tail_and_long_head(gpu, oracle, lanes_cfg, rna):
    """>= 1024 reads of mixed lengths: the dispatch order exists, the reads under 16 384 samples are its tail and go to
    k_event_multi, the others keep a wavefront each (one of them long enough to be cut into segments)"""
    rs = np.random.RandomState(5)
    lens = np.exp(rs.uniform(np.log(1), np.log(40000), size=1100)).astype(np.int64)
    lens[7] = 300000; lens[100] = 16384; lens[101] = 16383; lens[500] = 0
    lens[102] = 65536; lens[103] = 65535; lens[104] = 50000   # (RNA parameters: reads under 65 536 samples are short)
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens.tolist(), seed=23, kind=rna)
    for lanes in (0, 4, 32):   # 0: the library's own choice for this batch
        lanes_cfg(lanes)
        got, st = gpu.event(reads, dig, off, rng, rna)
        assert st.n_split_reads == 1 and st.n_capacity_overflow == 0
        pick = list(range(0, 1100, 37)) + [7, 100, 101, 102, 103, 104, 500]
        _check_events(oracle, [reads[i] for i in pick], dig[pick], off[pick], rng[pick], rna, [got[i] for i in pick])
        assert st.n_events_total == sum(g.start.size for g in got)


def test_edge_signals_packed(gpu, oracle, lanes_cfg):
    rs = np.random.RandomState(3)
    n = 6000
    const = np.full(n, 500, dtype=np.int16)
    steps = np.repeat(rs.randint(300, 700, size=n // 10), 10).astype(np.int16)
    noise = rs.randint(-32768, 32767, size=n).astype(np.int16)
    tiny = (rs.randint(400, 600, size=n)).astype(np.int16)
    tiny[1234] = -9
    zeros = np.zeros(n, dtype=np.int16)
    dense = (np.repeat(np.tile([420, 610], n // 6 + 1), 3)[:n] + rs.randint(-2, 3, size=n)).astype(np.int16)
    flat = np.full(n, 254, dtype=np.int16)
    reads = [const, steps, noise, tiny, zeros, dense, flat, steps[:777]]
    R = len(reads)
    dig = np.full(R, 8192.0); off = np.array([10.0, 3.0, 0.0, 11.0, 0.0, 5.0, 0.0, 4.0]); rng = np.full(R, 1402.882324)
    for lanes, lead in ((1, 0), (4, 0), (8, 16), (32, 0)):
        lanes_cfg(lanes, lead)
        for rna in (0, 1):
            got, st = gpu.event(reads, dig, off, rng, rna)
            _check_events(oracle, reads, dig, off, rng, rna, got)
            assert st.n_fallback_reads >= 1


def test_packed_reads_on_odd_addresses_decline_one_by_one(gpu, oracle, lanes_cfg):
    from test_gpu_device_api import _run_layout
    lanes_cfg(16)
    reads, _, _, _ = gpu.synth_reads_host(5, [3000, 2001, 999, 1500, 1200], seed=31, kind=0)
    offsets = [0, 3001, 5008, 6032, 0]
    offsets[4] = 7552
    n_samples = (offsets[4] + 1200 + 7) // 8 * 8
    b, arena = _run_layout(gpu, oracle, reads, offsets, n_samples, 0)
    st = arena.status()
    assert st.n_capacity_overflow == 0 and 2 <= st.n_fallback_reads < 5
    for r, raw in enumerate(reads):
        exp = oracle.event_raw(raw, 8192.0, 7.0, 1402.882324, 0)
        got = arena.read_events(r)
        assert got.start.size == exp.start.size, "read %d" % r
        assert np.array_equal(got.start.astype(np.uint64), exp.start)
        assert np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32))
        assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32))


@pytest.mark.parametrize("rna", [0, 1])
def test_a_read_that_is_short_and_long_belongs_to_its_segments(gpu, oracle, lanes_cfg, rna):
    """with test-sized segments a read can be under the short threshold and over the long one: the packed kernel must
    leave it to its segments (the two run side by side on different streams)"""
    L = gpu.load_library()
    lens = [5000, 3000, 900, 1500, 12000, 700, 2048, 1025, 1024, 40000]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=3 + rna, kind=rna)
    for lanes in (4, 16):
        lanes_cfg(lanes)
        gpu.event_configure(1024, 1025, 0)
        for _ in range(3):
            got, st = gpu.event(reads, dig, off, rng, rna)
            _check_events(oracle, reads, dig, off, rng, rna, got)
            assert st.n_split_reads == sum(1 for n in lens if n >= 1025)


@pytest.mark.parametrize("rna", [0, 1])
def test_packing_threshold_is_no_cliff(gpu, rna):
    """VERDICT r04 task 8b: reads under short_max samples (16 384; 65 536 with the RNA preset) share wavefronts in large
    batches (api.hip: event_multi_plan).  Just under and just over that length the batch is timed packed and one read
    per wavefront: what the rule picks may be at most 1.10 x the other (+ 30 us)."""
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    short_max = 65536 if rna else 16384

    def timed(b, lanes, smax):
        o = gpu.EVENT_OPTIONS
        old = (o.lanes_per_short_read, o.short_max)
        o.lanes_per_short_read, o.short_max = lanes, smax
        try:
            arena = device.EventArena(b)
        finally:
            o.lanes_per_short_read, o.short_max = old
        for _ in range(3):
            device.event(b, arena, rna)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); device.event(b, arena, rna); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    report = []
    for length in (short_max - 384, short_max + 64):
        n = max(int(1.0e9 // length), 26000)   # (packing also wants >= 4 rounds of wavefronts: 24 576 reads at 32 lanes each)
        b = device.synth_reads(n, length, seed=11, kind=rna, device=dev)
        packed = gpu.event_plan(n, b.total_samples, length, rna).lanes_per_short_read != 0
        t_whole, t_packed = timed(b, -1, 0), timed(b, 0, 2 * short_max)
        t_rule, t_other = (t_packed, t_whole) if packed else (t_whole, t_packed)
        report.append((length, "packed" if packed else "whole", round(t_packed, 3), round(t_whole, 3)))
        assert t_rule <= 1.10 * t_other + 0.03, report
        del b
        torch.cuda.empty_cache()
    assert [r[1] for r in report] == ["packed", "whole"], report
    print(report)
