"""GPU: the device-pointer API (sgk_event on caller-laid-out HBM buffers) -- odd layouts, overflow
reporting, and size-independent properties at a BASELINE-sized batch."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


# what recycled device memory looked like in the gaps of a job's sample arena when this was found: pointer-like
# words, i.e. a few huge values between runs of zeros
STALE = np.array([4096, 961, 31966, 0, -8192, -12610, 31957, 0, 16, 0, 0, 0, 0, 0, 16, 0, -8176, -12610, 31957, 0, 512,
                  0, 512, 0, 9, 0, 13, 0, 0, -26880, 31957, 0] + [0] * 32, dtype=np.int16)


def _run_layout(gpu, oracle, reads, offsets, n_samples, rna, slots_for=None, gap_fill=None):
    """Place reads at the given sample offsets of one buffer and run sgk_event through the device API."""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    host = np.zeros(n_samples, dtype=np.int16)
    host[:] = 777  # gap samples are arbitrary data, not zeros
    if gap_fill is not None:
        host[:] = np.resize(gap_fill, n_samples)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    for r, o in zip(reads, offsets):
        host[o:o + len(r)] = r
    b = device.DeviceReads(
        samples=torch.from_numpy(host).to(dev), offsets=torch.tensor(offsets, dtype=torch.int64, device=dev),
        lengths=torch.tensor(lens, dtype=torch.int32, device=dev),
        dig=torch.full((len(reads),), 8192.0, dtype=torch.float64, device=dev),
        off=torch.full((len(reads),), 7.0, dtype=torch.float64, device=dev),
        rng=torch.full((len(reads),), 1402.882324, dtype=torch.float64, device=dev),
        n_reads=len(reads), max_read_len=int(lens.max()), n_samples=n_samples,
        offsets_host=np.array(offsets, dtype=np.uint64), lengths_host=lens.astype(np.uint32))
    arena = device.EventArena(b)
    if slots_for is not None:
        slots = np.zeros(len(reads) + 1, dtype=np.int64)
        np.cumsum([slots_for(int(n)) for n in lens], out=slots[1:])
        arena.slots_host = slots
        arena.slots = torch.from_numpy(slots).to(dev)
    if gap_fill is not None:  # nothing may depend on what the workspace or the output arena held before
        arena.ws.fill_(0xA5)
        arena.events.fill_(-1); arena.n_events.fill_(-1)
    device.event(b, arena, rna)
    torch.cuda.synchronize()
    return b, arena


def test_unaligned_and_edge_reads_take_the_exact_fallback(gpu, oracle):
    reads, _, _, _ = gpu.synth_reads_host(5, [30000, 20001, 9999, 15000, 12000], seed=31, kind=0)
    # read 0 at the very start of the buffer (no head room: fine since round 2, no lane runs in front of a read),
    # read 1 on an odd sample, read 2 on a 2-byte-but-not-16-byte boundary, read 3 aligned with room, read 4 flush
    # against the buffer end
    offsets = [0, 30001, 50008, 60032, 0]
    offsets[4] = 75040
    n_samples = offsets[4] + len(reads[4])
    n_samples = (n_samples + 7) // 8 * 8
    b, arena = _run_layout(gpu, oracle, reads, offsets, n_samples, 0)
    st = arena.status()
    assert st.n_capacity_overflow == 0
    assert st.n_fallback_reads >= 2  # reads 1 and 4 cannot use the fast path
    for r, raw in enumerate(reads):
        exp = oracle.event_raw(raw, 8192.0, 7.0, 1402.882324, 0)
        got = arena.read_events(r)
        assert got.start.size == exp.start.size, "read %d" % r
        assert np.array_equal(got.start.astype(np.uint64), exp.start)
        assert np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32))
        assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32))


def test_capacity_overflow_is_reported_not_silent(gpu, oracle):
    reads, _, _, _ = gpu.synth_reads_host(2, [20000, 20000], seed=32, kind=0)
    offsets = [256, 256 + 20032]
    b, arena = _run_layout(gpu, oracle, reads, offsets, 256 + 2 * 20032 + 320, 0,
                           slots_for=lambda n: 100)  # far too few slots
    st = arena.status()
    assert st.n_capacity_overflow == 2
    exp = oracle.event_raw(reads[0], 8192.0, 7.0, 1402.882324, 0)
    assert int(arena.n_events[0].item()) == exp.start.size  # the true count is still reported
    got = arena.read_events(0)
    assert np.array_equal(got.start[:100].astype(np.uint64), exp.start[:100])  # the slots that exist are right


def test_baseline_sized_batch_properties(gpu, oracle):
    """BASELINE config 2 at its full size (10 000 reads x 100 000 samples): properties that do not need the
    oracle on every read, plus the oracle on a sample of reads."""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    b = device.synth_reads(10000, 100000, seed=1, kind=0, device=dev)
    arena = device.EventArena(b)
    device.event(b, arena, 0)
    torch.cuda.synchronize()
    st = arena.status()
    nev = arena.n_events[:b.n_reads].to(torch.int64)
    assert st.n_capacity_overflow == 0 and int(nev.sum().item()) == st.n_events_total
    slots = torch.from_numpy(arena.slots_host[:-1]).to(dev)
    # every read: first event starts at 0, events are contiguous, lengths sum to n, last ends at n
    first = arena.start[slots].to(torch.int64)
    assert bool((first == 0).all())
    last = slots + nev - 1
    assert bool(((arena.start[last].to(torch.int64) + arena.length[last].to(torch.int64)) == 100000).all())
    for r in (0, 4999, 9999):
        s = int(arena.slots_host[r]); k = int(nev[r].item())
        st_r = arena.start[s:s + k].to(torch.int64); ln_r = arena.length[s:s + k].to(torch.int64)
        assert bool((st_r[1:] == st_r[:-1] + ln_r[:-1]).all()) and int(ln_r.sum().item()) == 100000
        assert bool((ln_r >= 1).all()) and bool(torch.isfinite(arena.mean[s:s + k]).all())
    # idempotence: a second pass over the same batch gives identical bytes
    snap = (arena.start.clone(), arena.length.clone(), arena.mean.clone(), arena.stdv.clone(), nev.clone())
    device.event(b, arena, 0)
    torch.cuda.synchronize()
    nev2 = arena.n_events[:b.n_reads].to(torch.int64)
    assert bool((nev2 == snap[4]).all())
    for r in (5, 1234):
        s = int(arena.slots_host[r]); k = int(nev2[r].item())
        assert bool((arena.start[s:s + k] == snap[0][s:s + k]).all())
        assert bool((arena.mean[s:s + k].view(torch.int32) == snap[2][s:s + k].view(torch.int32)).all())
    # oracle on a sample
    for r in (0, 777, 5555, 9999):
        o = int(b.offsets_host[r]); n = int(b.lengths_host[r])
        raw = b.samples[o:o + n].cpu().numpy()
        exp = oracle.event_raw(raw, float(b.dig[r]), float(b.off[r]), float(b.rng[r]), 0)
        got = arena.read_events(r)
        assert np.array_equal(got.start.astype(np.uint64), exp.start)
        assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32))


def test_getevents_shim_matches_reference_signature(gpu, oracle, sp1):
    """sgk_getevents(nsample, pA*, rna): the drop-in for getevents() (src/events.c:553) fed with pA floats."""
    L = gpu.load_library()

    class Ev(C.Structure):
        _fields_ = [("start", C.c_uint64), ("length", C.c_float), ("mean", C.c_float), ("stdv", C.c_float)]

    class Tab(C.Structure):
        _fields_ = [("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t), ("event", C.POINTER(Ev))]

    L.sgk_getevents.restype = Tab
    L.sgk_getevents.argtypes = [C.c_size_t, C.c_void_p, C.c_int8]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for r in sp1.reads[:3]:
        pa = oracle.pa(r.raw, r.digitisation, r.offset, r.range)
        exp = oracle.getevents(pa, 0)
        t = L.sgk_getevents(pa.size, pa.ctypes.data, 0)
        assert t.n == exp.start.size and t.start == 0 and t.end == t.n
        got = np.array([(t.event[i].start, t.event[i].length, t.event[i].mean, t.event[i].stdv)
                        for i in range(t.n)], dtype=np.float64)
        assert np.array_equal(got[:, 0].astype(np.uint64), exp.start)
        assert np.array_equal(got[:, 1].astype(np.float32), exp.length)
        assert np.array_equal(got[:, 2].astype(np.float32).view(np.uint32), exp.mean.view(np.uint32))
        libc.free(t.event)
    # pA input through the chained segments (round 4): a batch of one read is a partial round of wavefronts, so the tail
    # split cuts a 100 000-sample read into 8 segments; a 700 000-sample read is long anyway (6 segments of 131 072)
    for n, rna in ((100000, 0), (700000, 0), (300000, 1)):
        reads, dig, off, rng = gpu.synth_reads_host(1, n, seed=5 + rna, kind=rna)
        pa = oracle.pa(reads[0], dig[0], off[0], rng[0])
        exp = oracle.getevents(pa, rna)
        t = L.sgk_getevents(pa.size, pa.ctypes.data, rna)
        assert t.n == exp.start.size, (n, rna, t.n, exp.start.size)
        ev = np.ctypeslib.as_array(C.cast(t.event, C.POINTER(C.c_uint8)), shape=(t.n * C.sizeof(Ev),)).copy()
        rec = np.frombuffer(ev.tobytes(), dtype=np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4"),
                                                          ("pad", "<u4")]))
        assert np.array_equal(rec["start"], exp.start) and np.array_equal(rec["length"], exp.length)
        assert np.array_equal(rec["mean"].view(np.uint32), exp.mean.view(np.uint32))
        assert np.array_equal(rec["stdv"].view(np.uint32), exp.stdv.view(np.uint32))
        libc.free(t.event)


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("fill", ["stale", "random", "extremes"])
def test_gap_content_never_matters(gpu, oracle, kind, fill):
    """The fast path reads up to 64 (DNA) / 256 (RNA) samples before a read and 16 after it without bounds
    checks; whatever sits there must not change a single event.  Regression: with EXACTLY 256 samples of head
    room the RNA pass had to redirect its first history groups, and initial window sums taken from memory
    disagreed with the redirected copies that later left the windows (only visible with non-constant gaps)."""
    lens = [100000, 5000, 70001, 300]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=21, kind=kind)
    rs = np.random.RandomState(4)
    gap = {"stale": STALE, "random": rs.randint(-32768, 32767, size=4099).astype(np.int16),
           "extremes": np.where(rs.rand(997) < 0.5, -32768, 32767).astype(np.int16)}[fill]
    for head in (256, 64 if kind == 0 else 256, 320):
        offsets, o = [], head
        for n in lens:
            offsets.append(o)
            o += (n + 63) // 64 * 64
        n_samples = o + 64
        b, arena = _run_layout(gpu, oracle, reads, offsets, n_samples, kind, gap_fill=gap)
        st = arena.status()
        assert st.n_fallback_reads == 0          # all of them took the fast path
        for r, raw in enumerate(reads):
            exp = oracle.event_raw(raw, 8192.0, 7.0, 1402.882324, kind)
            got = arena.read_events(r)
            assert got.start.size == exp.start.size, "head %d read %d: %d events, oracle %d" % (head, r, got.start.size, exp.start.size)
            assert np.array_equal(got.start.astype(np.uint64), exp.start)
            assert np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32))
            assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32))


def test_many_short_reads_next_to_one_very_long_read(gpu):
    """pa / qts / the synthetic generator launch one workgroup per (read, 8192-sample slab) pair with the slab count
    of the LONGEST read: 70 000 short reads next to one 2 000 000-sample read are 17 million pairs, more than one
    launch can hold (gridDim.x * 256 threads < 2^32) -- every pair must still be processed"""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    lens = np.full(70001, 100, dtype=np.int64)
    lens[12345] = 2_000_000
    assert len(lens) * ((int(lens.max()) + 8191) // 8192) * 256 >= 2 ** 32
    b = device.synth_reads(len(lens), 100, seed=9, kind=0, device=dev, lengths=lens)
    torch.cuda.synchronize()
    dig = b.dig[:b.n_reads].cpu().numpy(); off = b.off[:b.n_reads].cpu().numpy(); rng = b.rng[:b.n_reads].cpu().numpy()
    assert (dig > 0).all() and (rng > 0).all()          # the generator reached every read
    out = torch.zeros(b.n_samples, dtype=torch.float32, device=dev)
    device.pa(b, out)
    torch.cuda.synchronize()
    raw_all = b.samples.cpu().numpy()
    got_all = out.cpu().numpy()
    for r in (0, 1, 12344, 12345, 12346, 35000, 69999, 70000):
        o = int(b.offsets_host[r]); n = int(b.lengths_host[r])
        raw = raw_all[o:o + n].astype(np.float32)
        exp = (raw + np.float32(off[r])) * np.float32(rng[r] / dig[r])
        assert np.array_equal(exp.view(np.uint32), got_all[o:o + n].view(np.uint32)), "pa, read %d" % r
    device.qts(b, 3, 0)                                  # floor: clear the low 3 bits
    torch.cuda.synchronize()
    q_all = b.samples.cpu().numpy()
    for r in (0, 12345, 70000):
        o = int(b.offsets_host[r]); n = int(b.lengths_host[r])
        assert np.array_equal(q_all[o:o + n], (raw_all[o:o + n] >> 3) << 3), "qts, read %d" % r


def test_baseline_config4_and_5_at_their_shard_size(gpu, oracle):
    """BASELINE configs 4 and 5 at their per-GPU size (125 000 DNA reads x 100 000 samples: 25 GB of samples, 50 GB
    of pA, 67 GB of event slots): properties that hold for every read, idempotence, and the oracle bit for bit on a
    sample of reads -- fused stat + pA (config 4), then event on the same resident pool (config 5)."""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 170 * (1 << 30):
        pytest.skip("needs 170 GB of free HBM (this device has %.0f GB free)" % (free / (1 << 30)))
    R, N = 125000, 100000
    b = device.synth_reads(R, N, seed=3, kind=0, device=dev)
    pa = torch.empty(b.n_samples, dtype=torch.float32, device=dev)
    rec, _ = device.stat_pa(b, pa)
    torch.cuda.synchronize()
    got = np.frombuffer(rec.cpu().numpy().tobytes(), dtype=gpu.STAT_DTYPE)[:R]
    # every read: the record carries its length, the statistics are finite, the pA median is the pA of the raw median
    assert bool((got["n"] == N).all())
    for name in ("raw_mean", "pa_mean", "raw_std", "pa_std", "pa_median"):
        assert bool(np.isfinite(got[name]).all()), name
    unit = (np.float32(1402.882324) / np.float32(8192.0)).astype(np.float32)
    offs = b.off.cpu().numpy().astype(np.float32)
    exp_med = ((got["raw_median"].astype(np.float32) + offs[:R]) * unit).astype(np.float32)
    assert np.array_equal(exp_med.view(np.uint32), got["pa_median"].view(np.uint32))
    # idempotence: a second pass gives identical records and identical pA (checksum of the whole 50 GB)
    cks = lambda t: (int(t.view(torch.int32).to(torch.int64).sum().item()), int(t.view(torch.int32)[::4097].to(torch.int64).sum().item()))
    c1 = cks(pa)
    rec2, _ = device.stat_pa(b, pa)
    torch.cuda.synchronize()
    assert rec2.cpu().numpy().tobytes() == rec.cpu().numpy().tobytes() and cks(pa) == c1
    # the oracle on a sample
    for r in (0, 1, 62499, 124998, 124999):
        o = int(b.offsets_host[r])
        raw = b.samples[o:o + N].cpu().numpy()
        d, of, rg = float(b.dig[r]), float(b.off[r]), float(b.rng[r])
        e = oracle.stat(raw, d, of, rg)
        g = got[r]
        assert int(g["raw_median"]) == e[4]
        for name, ev in (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]), ("pa_median", e[5])):
            assert np.float32(g[name]).view(np.uint32) == np.float32(ev).view(np.uint32), (r, name)
        assert np.array_equal(pa[o:o + N].cpu().numpy().view(np.uint32), oracle.pa(raw, d, of, rg).view(np.uint32))
    # ---- config 5: the pipeline over the same pool (sgk_pipeline: fused stat + pA, then event): the same 50 GB of pA
    # (checksum) and the same records as config 4's call
    pa.zero_()
    arena = device.EventArena(b)
    rec5, _ = device.pipeline(b, arena, 0, pa)
    torch.cuda.synchronize()
    assert cks(pa) == c1
    assert rec5.cpu().numpy().tobytes() == rec.cpu().numpy().tobytes()
    del pa
    torch.cuda.empty_cache()
    st = arena.status()
    nev = arena.n_events[:R].to(torch.int64)
    assert st.n_capacity_overflow == 0 and int(nev.sum().item()) == st.n_events_total
    slots = torch.from_numpy(arena.slots_host[:-1]).to(dev)
    assert bool((arena.start[slots] == 0).all())
    last = slots + nev - 1
    assert bool(((arena.start[last].to(torch.int64) + arena.length[last].to(torch.int64)) == N).all())
    snap_n = nev.clone()
    first_ev = arena.events[slots].clone()
    device.event(b, arena, 0)
    torch.cuda.synchronize()
    assert bool((arena.n_events[:R].to(torch.int64) == snap_n).all()) and bool((arena.events[slots] == first_ev).all())
    for r in (0, 77777, 124999):
        o = int(b.offsets_host[r])
        raw = b.samples[o:o + N].cpu().numpy()
        exp = oracle.event_raw(raw, float(b.dig[r]), float(b.off[r]), float(b.rng[r]), 0)
        g = arena.read_events(r)
        assert g.start.size == exp.start.size and np.array_equal(g.start.astype(np.uint64), exp.start)
        assert np.array_equal(g.mean.view(np.uint32), exp.mean.view(np.uint32))
        assert np.array_equal(g.stdv.view(np.uint32), exp.stdv.view(np.uint32))


def test_pipeline_gives_what_the_three_calls_give(gpu):
    """sgk_pipeline: pA (bit-identical to sgk_pa), events (as sgk_event_opt), stat records (as sgk_stat_opt) from one
    call, over a batch in which every way a read can go through `event` occurs: whole reads, reads cut into segments
    (long; and the whole batch cut: the tail split of a small batch), packed short reads, reads the exactness guard
    sends to the fallback kernel (a sample at 0 pA), empty and tiny reads."""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    for lens, kind, opts_ in (([100000, 5000, 70001, 300, 0, 1, 2047, 2048, 2049, 33333], 0, {}),
                              ([300001, 100000, 5000, 140000], 1, {"segment_len": 32768, "long_min": 120000}),
                              ([5000] * 3000 + [40000, 100000], 0, {}),
                              ([100000] * 40, 0, {"warmup": 16})):
        b = device.synth_reads(len(lens), 0, seed=17, kind=kind, device=dev, lengths=np.asarray(lens, dtype=np.int64))
        # one read that crosses 0 pA (guard fails -> k_event_fallback): offset = -(a sample of it)
        if lens[0] >= 1000:
            o0 = int(b.offsets_host[0])
            b.off[0] = -float(int(b.samples[o0 + 500].item()))
        gpu.event_configure(opts_.get("segment_len", 0), opts_.get("long_min", 0), opts_.get("warmup", 0))
        try:
            want_pa = torch.zeros(b.n_samples, dtype=torch.float32, device=dev)
            device.pa(b, want_pa)
            a1 = device.EventArena(b)
            device.event(b, a1, kind)
            want_stat = device.stat(b).cpu().numpy().tobytes()
            a2 = device.EventArena(b)
            got_pa = torch.full((b.n_samples,), float("nan"), dtype=torch.float32, device=dev)
            got_stat, _ = device.pipeline(b, a2, kind, got_pa)
            torch.cuda.synchronize()
        finally:
            gpu.event_configure(0, 0, 0)
        assert a2.status().n_fallback_reads == a1.status().n_fallback_reads
        if lens[0] >= 1000:
            assert a1.status().n_fallback_reads >= 1
        assert got_stat.cpu().numpy().tobytes() == want_stat
        n1, n2 = a1.n_events.cpu().numpy(), a2.n_events.cpu().numpy()
        assert (n1 == n2).all()
        for r, n in enumerate(lens):
            o = int(b.offsets_host[r])
            assert torch.equal(got_pa[o:o + n].view(torch.int32), want_pa[o:o + n].view(torch.int32)), "read %d pA" % r
            s0, k = int(a1.slots_host[r]), int(n1[r])
            assert torch.equal(a1.events[s0:s0 + k], a2.events[s0:s0 + k]), "read %d events" % r


def test_baseline_config3_at_its_size(gpu, oracle):
    """BASELINE config 3 at its size (50 000 RNA-headed reads x 100 000 samples: 10 GB of samples, 27 GB of event
    slots): `event` with RNA parameters + `prefix`.  Properties of every read, idempotence, the tail split at work
    (50 000 reads are 24.4 rounds of RNA wavefronts: no split; the same call on the first 9 000 reads has one), and the
    oracle bit for bit on a sample of reads for both subtools."""
    torch = _torch()
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free HBM (this device has %.0f GB free)" % (free / (1 << 30)))
    R, N = 50000, 100000
    b = device.synth_reads(R, N, seed=2, kind=1, device=dev)
    arena = device.EventArena(b)
    device.event(b, arena, 1)
    torch.cuda.synchronize()
    st = arena.status()
    nev = arena.n_events[:R].to(torch.int64)
    assert st.n_capacity_overflow == 0 and int(nev.sum().item()) == st.n_events_total and st.n_split_reads == 0
    slots = torch.from_numpy(arena.slots_host[:-1]).to(dev)
    assert bool((arena.start[slots].to(torch.int64) == 0).all())
    last = slots + nev - 1
    assert bool(((arena.start[last].to(torch.int64) + arena.length[last].to(torch.int64)) == N).all())
    # events per read in the range the RNA-like generator gives (25.4 samples per event on average)
    assert 3000 < float(nev.double().mean().item()) < 5000
    snap_n = nev.clone()
    snap = {r: arena.read_events(r) for r in (5, 31234)}
    device.event(b, arena, 1)
    torch.cuda.synchronize()
    assert bool((arena.n_events[:R].to(torch.int64) == snap_n).all())
    for r, e in snap.items():
        g = arena.read_events(r)
        assert np.array_equal(g.start, e.start) and np.array_equal(g.stdv.view(np.uint32), e.stdv.view(np.uint32))
    for r in (0, 1, 24999, 49998, 49999):
        o = int(b.offsets_host[r])
        raw = b.samples[o:o + N].cpu().numpy()
        exp = oracle.event_raw(raw, float(b.dig[r]), float(b.off[r]), float(b.rng[r]), 1)
        got = arena.read_events(r)
        assert np.array_equal(got.start.astype(np.uint64), exp.start)
        assert np.array_equal(got.length.astype(np.float32), exp.length)
        assert np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32))
        assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32))
    # prefix over the same resident batch
    rec = device.prefix(b, 1, 0)
    torch.cuda.synchronize()
    pf = np.frombuffer(rec.cpu().numpy().tobytes(), dtype=gpu.PREFIX_DTYPE)[:R]
    assert bool((pf["n"] == N).all())
    found = pf["adapt_y"] > 0
    assert found.mean() > 0.9                       # the synthetic RNA reads carry an adaptor stall
    assert bool((pf["adapt_x"][found] < pf["adapt_y"][found]).all()) and bool((pf["adapt_y"][found] <= N).all())
    rec2 = device.prefix(b, 1, 0)
    torch.cuda.synchronize()
    assert rec2.cpu().numpy().tobytes() == rec.cpu().numpy().tobytes()
    for r in (0, 24999, 49999):
        o = int(b.offsets_host[r])
        raw = b.samples[o:o + N].cpu().numpy()
        e = oracle.prefix(raw, float(b.dig[r]), float(b.off[r]), float(b.rng[r]), 1, 0)
        g = pf[r]
        for name in ("adapt_x", "adapt_y", "polya_x", "polya_y"):
            assert int(g[name]) == int(getattr(e, name)), (r, name)
        if e.adapt_y > 0:
            for name in ("adapt_mean", "adapt_std", "adapt_median"):
                assert np.float32(g[name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32), (r, name)
        if e.polya_y > 0:
            for name in ("polya_mean", "polya_std", "polya_median"):
                assert np.float32(g[name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32), (r, name)
    # the tail split: the first 1 700 reads of the same batch are 0.83 of a round of RNA wavefronts -- every read is cut
    del arena
    torch.cuda.empty_cache()
    b9 = device.synth_reads(1700, N, seed=2, kind=1, device=dev)
    a9 = device.EventArena(b9)
    device.event(b9, a9, 1)
    torch.cuda.synchronize()
    s9 = a9.status()
    assert s9.n_split_reads == 1700 and s9.n_fallback_reads == 0
    assert bool((a9.n_events[:1700].to(torch.int64) == snap_n[:1700]).all())
