"""GPU parity: long reads in stat / jnn / prefix (k_long_chains: the sequential float sums of a long read on 16
wavefronts, composed from per-tile summaries) against the oracle, bit for bit, and against the one-wave path."""
import numpy as np
import pytest

from test_gpu_stat import _check_jnn, _check_prefix, _check_stat

pytestmark = pytest.mark.gpu


@pytest.fixture
def long_min(gpu):
    """sets sgk_stat_options_t::long_min for the test's calls and restores the default afterwards"""
    def set_(v):
        gpu.stat_configure(2, v)
    yield set_
    gpu.stat_configure(0, 0)


def _hostile(gpu, lens, seed):
    """synthetic reads + the shapes the summaries do not cover or cover with ties: signed ADC codes (negative terms
    and a negative running sum), a constant read (every addition of the raw chain rounds the same way: the predicted
    binades run ahead of the true ones), an all-zero read, a read of +-32767 alternating, a negative range"""
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=seed, kind=0)
    reads = [r.copy() for r in reads]
    rs = np.random.RandomState(seed)
    rng = rng.copy()
    if len(reads) > 1:
        reads[1] = rs.randint(-2000, 2000, size=reads[1].size).astype(np.int16)
    if len(reads) > 2:
        reads[2][:] = 517
    if len(reads) > 3:
        reads[3][:] = 0
    if len(reads) > 4:
        reads[4][::2] = 32767
        reads[4][1::2] = -32767
    if len(reads) > 5:
        rng[5] = -rng[5]
    if len(reads) > 6:
        reads[6] = (-np.abs(reads[6].astype(np.int32)) - 3).astype(np.int16)   # every term negative
    return reads, dig, off, rng


def test_read_of_3000001_samples_all_three_subtools(gpu, oracle, long_min):
    """VERDICT r03 task 6: a 3 000 001-sample read next to ordinary ones, stat / jnn / prefix against the oracle"""
    lens = [3000001, 100000, 5000, 262144, 262143]
    for kind in (0, 1):
        reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=41 + kind, kind=kind)
        long_min(0)
        st = gpu.stat(reads, dig, off, rng)
        _check_stat(oracle, reads, dig, off, rng, st)
        for rna in (0, 1):
            _check_jnn(oracle, reads, rna, gpu.jnn(reads, dig, off, rng, rna))
        for pore in (0, 2):
            _check_prefix(oracle, reads, dig, off, rng, kind, pore, gpu.prefix(reads, dig, off, rng, kind, pore))
        long_min(-1)   # the one-wave path gives the same records
        st1 = gpu.stat(reads, dig, off, rng)
        assert st.tobytes() == st1.tobytes()


@pytest.mark.parametrize("threshold", [8192, 20000])
def test_many_long_reads_of_every_shape(gpu, oracle, long_min, threshold):
    """a low threshold makes every read above it long: ragged lengths around tile and round boundaries, hostile
    contents; two rounds (more than 2 097 152 samples) in one read"""
    lens = [8192, 8193, 16384, 16385, 20000, 20001, 33333, 65535, 65536, 100000, 131071, 250000, 1024 * 2048 + 5000,
            9000, 12000, 8191, 100, 0, 1]
    reads, dig, off, rng = _hostile(gpu, lens, 7)
    long_min(threshold)
    _check_stat(oracle, reads, dig, off, rng, gpu.stat(reads, dig, off, rng))
    for rna in (0, 1):
        _check_jnn(oracle, reads, rna, gpu.jnn(reads, dig, off, rng, rna))
    for pore in (0, 2):
        _check_prefix(oracle, reads, dig, off, rng, 0, pore, gpu.prefix(reads, dig, off, rng, 0, pore))
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=9, kind=1)
    for pore in (0, 2):
        _check_prefix(oracle, reads, dig, off, rng, 1, pore, gpu.prefix(reads, dig, off, rng, 1, pore))


def test_long_path_composes_most_tiles_and_equals_the_one_wave_path(gpu):
    """device API: the fused stat + pA records and the pA output are identical with and without the long path; the
    status says how many tile sums were composed from summaries (all but the binade crossings and the first tile)"""
    import torch
    from sigtk_amd import device
    lens = np.asarray([1500000, 300000, 100000, 700001, 4000], dtype=np.int64)
    dev = torch.device("cuda", 0)
    b = device.synth_reads(len(lens), 0, seed=5, kind=0, device=dev, lengths=lens)
    gpu.stat_configure(2, 0)
    try:
        rec, pa = device.stat_pa(b)
        st = device.long_status(b, "stat")
        rec, pa = rec.cpu().numpy().copy(), pa.cpu().numpy().copy()
        pre = device.prefix(b, 0, 0).cpu().numpy().copy()
        sp = device.long_status(b, "prefix")
        arena = device.SegArena(b)
        device.jnn(b, arena, 0)
        sj = device.long_status(b, ws=arena.ws)
        segs = (arena.n_segs.cpu().numpy().copy(), arena.x.cpu().numpy().copy(), arena.y.cpu().numpy().copy())
        gpu.stat_configure(2, -1)
        rec1, pa1 = device.stat_pa(b)
        assert device.long_status(b, "stat").n_long_reads == 0
        pre1 = device.prefix(b, 0, 0).cpu().numpy()
        arena1 = device.SegArena(b)
        device.jnn(b, arena1, 0)
    finally:
        gpu.stat_configure(0, 0)
    assert rec.tobytes() == rec1.cpu().numpy().tobytes()
    pa1 = pa1.cpu().numpy()
    for r in range(len(lens)):   # (the gaps between the reads are not written)
        o, n = int(b.offsets_host[r]), int(lens[r])
        assert pa[o:o + n].tobytes() == pa1[o:o + n].tobytes(), "read %d pA" % r
    assert pre.tobytes() == pre1.tobytes()
    n1 = arena1.n_segs.cpu().numpy()
    assert (segs[0] == n1).all()
    x1, y1, slots = arena1.x.cpu().numpy(), arena1.y.cpu().numpy(), arena1.slots_host
    for r in range(len(lens)):
        s0, n = int(slots[r]), int(n1[r])
        assert (segs[1][s0:s0 + n] == x1[s0:s0 + n]).all() and (segs[2][s0:s0 + n] == y1[s0:s0 + n]).all()
    for s, chains in ((st, 4), (sp, 2), (sj, 2)):
        assert s.n_long_reads == 3 and s.n_timeouts == 0
        tiles = sum((int(n) + 1023 + 7) // 1024 for n in lens[[0, 1, 3]])
        assert chains * (tiles - 12) <= s.n_tiles <= chains * (tiles + 3)
        assert 0 < s.n_true_tiles <= s.n_tiles // 20, (s.n_tiles, s.n_true_tiles)


def test_smaller_workspaces_are_accepted(gpu):
    """a caller that sizes the workspace as ABI 0.2.0 did (dispatch order only), or passes none, gets the same records:
    its long reads run on one wavefront"""
    import ctypes as C
    import torch
    from sigtk_amd import device
    from sigtk_amd.device import _ptr, _stream_ptr
    lens = np.asarray([400000, 3000, 100000] + [5000] * 1100, dtype=np.int64)   # (>= 1024 reads: the order is used)
    dev = torch.device("cuda", 0)
    b = device.synth_reads(len(lens), 0, seed=8, kind=0, device=dev, lengths=lens)
    L = gpu.load_library()
    full = device.stat(b).cpu().numpy().copy()
    assert device.long_status(b, "stat").n_long_reads == 1
    view = b.view()
    order_only = 64 + (len(lens) * 4 + 128 * 4 + 63) // 64 * 64
    for nbytes in (order_only, 64, 0):
        out = torch.zeros(len(lens) * gpu.STAT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        ws = torch.zeros(max(nbytes, 1), dtype=torch.uint8, device=dev)
        gpu.check(L.sgk_stat_opt(C.byref(view), _ptr(out), _ptr(ws) if nbytes else None, nbytes, _stream_ptr(),
                                 C.byref(gpu.STAT_OPTIONS)), "sgk_stat_opt")
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == full.tobytes(), "workspace of %d bytes" % nbytes


def test_a_batch_of_similar_long_reads_is_left_to_the_wave_kernels(gpu):
    """the per-batch threshold lists the reads one wave would outlast the batch on; 200 reads of 300 000 samples are
    all over it (262 144) and none is an outlier: the list is dropped (LC_AUTO_MAX_READS).  An explicit threshold
    keeps them on the long path; same records either way."""
    import torch
    from sigtk_amd import device
    lens = np.full(200, 300000, dtype=np.int64)
    b = device.synth_reads(len(lens), 0, seed=12, kind=0, device=torch.device("cuda", 0), lengths=lens)
    try:
        gpu.stat_configure(2, 0)
        auto = device.stat(b).cpu().numpy().copy()
        assert device.long_status(b, "stat").n_long_reads == 0
        gpu.stat_configure(2, 262144)
        forced = device.stat(b).cpu().numpy().copy()
        st = device.long_status(b, "stat")
        assert st.n_long_reads == 200 and st.n_tiles > 0
    finally:
        gpu.stat_configure(0, 0)
    assert auto.tobytes() == forced.tobytes()


@pytest.fixture
def fault(gpu):
    """sgk_stat_options_t::debug_fault for the test's calls (fault injection into the long path's barriers)"""
    def set_(v):
        gpu.STAT_OPTIONS.debug_fault = int(v)
    yield set_
    gpu.STAT_OPTIONS.debug_fault = 0


# mode 1: workgroup `part` of every long read never arrives at its barrier number `phase` (stat and prefix meet at 6
# barriers per read, jnn at 7); mode 2: nobody is withheld, the spin bound is `bound` polls (1: whoever is not last at a
# barrier gives up at once -- timeouts at every barrier of every read, the last one included)
def _withhold(part, phase):
    return 1 | (part << 8) | (phase << 16)


@pytest.mark.parametrize("fault_word", [_withhold(5, 1), _withhold(0, 2), _withhold(15, 3), _withhold(3, 4), _withhold(9, 6),
                                        _withhold(2, 7), 2 | (1 << 8), 2 | (48 << 8)])
def test_a_barrier_timeout_declines_the_read_and_one_wavefront_redoes_it(gpu, oracle, long_min, fault, fault_word):
    """VERDICT r04 missing 3 / ADVICE r04: a workgroup of a long read that waits in vain at a barrier used to leave the
    read's sums WRONG (and a counter nobody read).  Now the read is declined -- no workgroup of it writes anything of the
    subtool's output -- and the wave-per-read kernel takes it behind the join: records identical to the oracle's,
    sgk_long_status_t::n_timeouts says how many reads went that way."""
    import torch
    from sigtk_amd import device
    lens = [8192, 20001, 33333, 65536, 100000, 250000, 1024 * 2048 + 5000, 9000, 100, 0, 1, 700001]
    reads, dig, off, rng = _hostile(gpu, lens, 13)
    n_long = sum(1 for n in lens if n >= 20000)
    long_min(20000)
    fault(fault_word)
    _check_stat(oracle, reads, dig, off, rng, gpu.stat(reads, dig, off, rng))
    for rna in (0, 1):
        _check_jnn(oracle, reads, rna, gpu.jnn(reads, dig, off, rng, rna))
    for pore in (0, 2):
        _check_prefix(oracle, reads, dig, off, rng, 0, pore, gpu.prefix(reads, dig, off, rng, 0, pore))
    # the device API tells how many reads were declined: with a withheld workgroup every long read that reaches that
    # barrier (stat / prefix have 6 per read, jnn 7)
    lens_a = np.asarray(lens, dtype=np.int64)
    b = device.synth_reads(len(lens), 0, seed=13, kind=0, device=torch.device("cuda", 0), lengths=lens_a)
    rec, pa = device.stat_pa(b)
    st = device.long_status(b, "stat")
    rec, pa = rec.cpu().numpy().copy(), pa.cpu().numpy().copy()
    pre = device.prefix(b, 0, 0).cpu().numpy().copy()
    sp = device.long_status(b, "prefix")
    arena = device.SegArena(b)
    device.jnn(b, arena, 0)
    sj = device.long_status(b, ws=arena.ws)
    segs = (arena.n_segs.cpu().numpy().copy(), arena.x.cpu().numpy().copy(), arena.y.cpu().numpy().copy())
    mode, phase = fault_word & 0xff, (fault_word >> 16) & 0xff
    for s, nbar in ((st, 6), (sp, 6), (sj, 7)):
        assert s.n_long_reads == n_long
        if mode == 1:
            assert s.n_timeouts == (n_long if phase <= nbar else 0), (s.n_timeouts, n_long, phase, nbar)
        elif (fault_word >> 8) == 1:
            assert s.n_timeouts > 0
    # ... and the same batch without the fault (and without the long path) gives the same bytes
    fault(0)
    long_min(-1)
    rec1, pa1 = device.stat_pa(b)
    assert rec.tobytes() == rec1.cpu().numpy().tobytes()
    pa1 = pa1.cpu().numpy()
    for r in range(len(lens)):
        o, n = int(b.offsets_host[r]), int(lens[r])
        assert pa[o:o + n].tobytes() == pa1[o:o + n].tobytes(), "read %d pA" % r
    assert pre.tobytes() == device.prefix(b, 0, 0).cpu().numpy().tobytes()
    arena1 = device.SegArena(b)
    device.jnn(b, arena1, 0)
    n1, x1, y1, slots = arena1.n_segs.cpu().numpy(), arena1.x.cpu().numpy(), arena1.y.cpu().numpy(), arena1.slots_host
    assert (segs[0] == n1).all()
    for r in range(len(lens)):
        s0, n = int(slots[r]), int(n1[r])
        assert (segs[1][s0:s0 + n] == x1[s0:s0 + n]).all() and (segs[2][s0:s0 + n] == y1[s0:s0 + n]).all()


def test_a_job_reports_the_reads_its_long_path_declined(gpu, oracle, fault):
    """sgk_job_wait fetches the long path's header with the results: sgk_job_long_declined() (the CLI warns on it)"""
    lens = [300000, 5000, 400001]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=3, kind=0)
    job = gpu.Job(0)
    L = gpu.load_library()
    try:
        for word, want in ((0, 0), (_withhold(4, 2), 2), (0, 0)):
            fault(word)
            job.set_options()
            job.submit(gpu.TOOL_STAT, reads, dig, off, rng)
            st = job.wait()["stat"]
            assert L.sgk_job_long_declined(job.h) == want
            for r, raw in enumerate(reads):
                e = oracle.stat(raw, dig[r], off[r], rng[r])
                assert int(st[r]["raw_median"]) == e[4]
                for name, ev in (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]), ("pa_median", e[5])):
                    assert np.float32(st[r][name]).view(np.uint32) == np.float32(ev).view(np.uint32), (word, r, name)
            job.submit(gpu.TOOL_PREFIX, reads, dig, off, rng)
            job.wait()
            assert L.sgk_job_long_declined(job.h) == want
    finally:
        job.close()


def test_long_read_threshold_is_no_cliff(gpu):
    """The length from which a read gets 16 workgroups follows the batch and the tool (csrc/stat_args.h: LongRule).  A few
    reads just over and just under that length among 1 000 / 3 000 / 10 000 reads of 100 000 samples are timed on the path
    the library picks and on the other (long_min = -1: never; long_min = their length: all of them): the pick may be at
    most 1.12 x the other (+ 30 us)."""
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)

    def timed(fn, lm):
        gpu.stat_configure(0, lm)
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        finally:
            gpu.stat_configure(0, 0)

    report = []
    for n in (1000, 3000, 10000):
        for tool in ("stat", "jnn", "prefix"):
            thr = int(gpu.stat_plan(tool, n + 6, (n + 6) * 100000 + 6 * 400000, 500000).long_min)
            assert thr > 0
            for L in (int(thr * 0.9) // 64 * 64, int(thr * 1.12) // 64 * 64 + 64):
                lens = np.full(n + 6, 100000, dtype=np.int64)
                lens[n // 2:n // 2 + 6] = L
                b = device.synth_reads(n + 6, 100000, seed=17, kind=0, device=dev, lengths=lens)
                ar = device.SegArena(b)
                fn = {"stat": lambda: device.stat(b), "jnn": lambda: device.jnn(b, ar, 0), "prefix": lambda: device.prefix(b, 0, 0)}[tool]
                picked_long = int(gpu.stat_plan(tool, n + 6, b.total_samples, L).long_min) != 0
                t_never, t_long = timed(fn, -1), timed(fn, L)
                t_pick, t_other = (t_long, t_never) if picked_long else (t_never, t_long)
                report.append((tool, n, L, "long" if picked_long else "wave", round(t_never, 3), round(t_long, 3)))
                assert t_pick <= 1.12 * t_other + 0.03, report[-1]
                del b, ar
                torch.cuda.empty_cache()
    print(report)
