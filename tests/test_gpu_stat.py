"""GPU parity: stat / jnn / prefix HIP paths against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5  # tolerance stated by BASELINE.json's north_star for float means/stds


def _scal(recs):
    return (np.array([r.digitisation for r in recs]), np.array([r.offset for r in recs]),
            np.array([r.range for r in recs]))


def _close(a, b, what):
    a = np.float32(a); b = np.float32(b)
    if (np.isnan(a) and np.isnan(b)) or (np.isinf(a) and a == b):
        return
    assert abs(float(a) - float(b)) <= REL * max(abs(float(b)), 1e-30), "%s: gpu %r oracle %r" % (what, a, b)


def _check_stat(oracle, reads, dig, off, rng, got, exact=True):
    for r, raw in enumerate(reads):
        e = oracle.stat(raw, dig[r], off[r], rng[r])
        g = got[r]
        assert int(g["n"]) == raw.size
        assert int(g["raw_median"]) == e[4], "read %d raw_median" % r
        for name, ev in (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]),
                         ("pa_median", e[5])):
            _close(g[name], ev, "read %d %s" % (r, name))
            if exact:  # same sequential float order as the reference: expect identical bits
                assert np.float32(g[name]).view(np.uint32) == np.float32(ev).view(np.uint32), \
                    "read %d %s not bit-exact: %r vs %r" % (r, name, g[name], ev)


def test_stat_sp1_dna(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads]
    dig, off, rng = _scal(sp1.reads)
    _check_stat(oracle, reads, dig, off, rng, gpu.stat(reads, dig, off, rng))


def test_stat_synthetic_100k(gpu, oracle):
    reads, dig, off, rng = gpu.synth_reads_host(5, 100000, seed=3, kind=0)
    _check_stat(oracle, reads, dig, off, rng, gpu.stat(reads, dig, off, rng))


def test_stat_ragged_and_negative_range(gpu, oracle):
    lens = [1, 2, 3, 7, 8, 9, 63, 64, 65, 100, 1000, 4097, 70001]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=5, kind=0)
    rs = np.random.RandomState(1)
    reads[4] = rs.randint(-32768, 32767, size=8).astype(np.int16)
    reads[10] = rs.randint(-2000, 2000, size=1000).astype(np.int16)
    rng = rng.copy(); rng[5] = -rng[5]; rng[11] = -rng[11]
    _check_stat(oracle, reads, dig, off, rng, gpu.stat(reads, dig, off, rng))


def _check_jnn(oracle, reads, rna, got):
    for r, raw in enumerate(reads):
        ex, ey = oracle.jnn_raw(raw, rna)
        gx, gy = got[r]
        np.testing.assert_array_equal(gx.astype(np.int64), ex, err_msg="read %d seg x" % r)
        np.testing.assert_array_equal(gy.astype(np.int64), ey, err_msg="read %d seg y" % r)


def test_jnn_sp1_dna(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads]
    dig, off, rng = _scal(sp1.reads)
    for rna in (0, 1):
        _check_jnn(oracle, reads, rna, gpu.jnn(reads, dig, off, rng, rna))


@pytest.mark.parametrize("kind", [0, 1])
def test_jnn_synthetic(gpu, oracle, kind):
    lens = [0, 1, 149, 150, 151, 1000, 5000, 30000, 100000, 100000]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=9, kind=kind)
    for rna in (0, 1):
        _check_jnn(oracle, reads, rna, gpu.jnn(reads, dig, off, rng, rna))


def _check_prefix(oracle, reads, dig, off, rng, rna, pore, got):
    for r, raw in enumerate(reads):
        e = oracle.prefix(raw, dig[r], off[r], rng[r], rna, pore)
        g = got[r]
        assert (int(g["adapt_x"]), int(g["adapt_y"])) == (e.adapt_x, e.adapt_y), "read %d adaptor" % r
        assert (int(g["polya_x"]), int(g["polya_y"])) == (e.polya_x, e.polya_y), "read %d polyA" % r
        if e.adapt_y > 0:
            for name in ("adapt_mean", "adapt_std", "adapt_median"):
                _close(g[name], getattr(e, name), "read %d %s" % (r, name))
                assert np.float32(g[name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32)
        if e.polya_y > 0:
            for name in ("polya_mean", "polya_std", "polya_median"):
                _close(g[name], getattr(e, name), "read %d %s" % (r, name))
                assert np.float32(g[name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32)


def test_prefix_sp1_dna(gpu, oracle, sp1):
    reads = [r.raw for r in sp1.reads]
    dig, off, rng = _scal(sp1.reads)
    _check_prefix(oracle, reads, dig, off, rng, 0, 0, gpu.prefix(reads, dig, off, rng, 0, 0))
    # RNA handling of the same reads exercises find_polya on real signal
    _check_prefix(oracle, reads, dig, off, rng, 1, 0, gpu.prefix(reads, dig, off, rng, 1, 0))


@pytest.mark.parametrize("pore", [0, 2])
def test_prefix_synthetic_rna(gpu, oracle, pore):
    lens = [100, 2000, 2001, 2500, 20000, 50000, 100000, 100000, 100000, 100000]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=21, kind=1)
    got = gpu.prefix(reads, dig, off, rng, 1, pore)
    _check_prefix(oracle, reads, dig, off, rng, 1, pore, got)
    assert (int(got[0]["adapt_x"]), int(got[0]["adapt_y"])) == (-1, -1)   # too short (jnn.c:173-177)
    assert any(int(g["adapt_y"]) > 0 and int(g["polya_y"]) > 0 for g in got)  # the structure is found


def test_fused_stat_pa_matches_stat_and_pa(gpu, oracle):
    """sgk_stat_pa (BASELINE config 4): same records as sgk_stat, pA bit-identical to sgk_pa / the oracle;
    ragged lengths exercise the unaligned head/tail of the vectorised median pass; one read spans more than
    8192 distinct raw values (two-level select path), one has a negative range (mirrored pA median)."""
    import torch
    from sigtk_amd import device
    lens = [1, 7, 8, 9, 63, 64, 65, 1000, 4097, 30000, 70001]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=11, kind=0)
    rs = np.random.RandomState(3)
    reads[8] = rs.randint(-32768, 32767, size=4097).astype(np.int16)
    rng = rng.copy(); rng[9] = -rng[9]
    dev = torch.device("cuda", 0)
    b = device.alloc_reads(np.asarray(lens, dtype=np.int64), dev, align=1)   # reads start at odd offsets too
    host = np.zeros(b.n_samples, dtype=np.int16)
    for r, raw in enumerate(reads):
        o = int(b.offsets_host[r]); host[o:o + raw.size] = raw
    b.samples.copy_(torch.from_numpy(host).to(dev))
    b.dig.copy_(torch.from_numpy(np.asarray(dig, dtype=np.float64)).to(dev))
    b.off.copy_(torch.from_numpy(np.asarray(off, dtype=np.float64)).to(dev))
    b.rng.copy_(torch.from_numpy(np.asarray(rng, dtype=np.float64)).to(dev))
    for kernels in (0, 1, 2):   # chosen per batch / one read per lane (k_moments + k_median) / one read per wavefront
        gpu.stat_configure(kernels)
        try:
            rec, pa = device.stat_pa(b)
            rec2 = device.stat(b)
            torch.cuda.synchronize()
        finally:
            gpu.stat_configure(0)
        got = np.frombuffer(rec.cpu().numpy().tobytes(), dtype=gpu.STAT_DTYPE)[:len(lens)]
        got2 = np.frombuffer(rec2.cpu().numpy().tobytes(), dtype=gpu.STAT_DTYPE)[:len(lens)]
        assert got.tobytes() == got2.tobytes()
        _check_stat(oracle, reads, dig, off, rng, got)
        pa_h = pa.cpu().numpy()
        for r, raw in enumerate(reads):
            o = int(b.offsets_host[r])
            exp = oracle.pa(raw, dig[r], off[r], rng[r])
            assert np.array_equal(pa_h[o:o + raw.size].view(np.uint32), exp.view(np.uint32)), "kernels %d read %d pA" % (kernels, r)


def _adversarial_reads(gpu, seed):
    """ragged + hostile reads for the wave-per-read sums (seqsum.h): every slow path of the chain is visited"""
    rs = np.random.RandomState(seed)
    lens = [0, 1, 2, 15, 16, 17, 63, 64, 65, 66, 127, 128, 129, 1000, 1023, 1024, 1025, 1040, 2047, 2048, 2049,
            5000, 5000, 5000, 5000, 5000, 5000, 20000, 33000, 100000, 100000, 250000, 40000, 40000, 40000, 3000,
            3000, 3000, 3000, 3000, 3000, 3000]
    lens += [int(v) for v in np.clip(np.exp(rs.normal(8.5, 1.2, size=60)), 1, 150000)]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=seed, kind=0)
    reads = [r.copy() for r in reads]
    dig = dig.copy(); off = off.copy(); rng = rng.copy()
    reads[22] = np.zeros(5000, dtype=np.int16)                              # all-zero raw
    reads[23] = np.full(5000, 333, dtype=np.int16)                          # constant (deviations are 0 or tiny)
    reads[24] = rs.randint(-32768, 32767, size=5000).astype(np.int16)       # full range, both signs
    reads[25] = rs.randint(-2000, 2000, size=5000).astype(np.int16)
    z = np.zeros(5000, dtype=np.int16); z[3000] = 7; z[4000:] = 1
    reads[26] = z
    reads[32] = np.full(40000, 32767, dtype=np.int16)                       # sums far beyond 2^24: ties every step
    reads[33] = np.full(40000, 4095, dtype=np.int16)
    o = rs.normal(500, 60, size=40000); o[20000] = 32000; o[30000] = -32000
    reads[34] = np.clip(np.rint(o), -32768, 32767).astype(np.int16)         # outliers comparable to the sum
    rng[35] = -rng[35]                                                       # negative unit: pA sum runs negative
    off[36] = -600.0                                                         # pA changes sign inside the read
    off[37] = -float(np.median(reads[37]))                                   # pA hovers around zero
    dig[38] = 0.0                                                            # unit = inf
    rng[39] = 0.0                                                            # unit = 0: every pA is 0
    rng[40] = 1e-30                                                          # tiny pA
    rng[41] = 1e38                                                           # huge pA: the sum overflows to inf
    # every pA is -0.0 (raw = -offset, negative unit): the reference's sum stays +0 (soak seed 2024, batch 2376)
    reads[2] = np.full(2, -18, dtype=np.int16); off[2] = 18.0; rng[2] = -abs(rng[2])
    reads[3] = np.full(15, -2, dtype=np.int16); off[3] = 2.0; rng[3] = -abs(rng[3])
    return reads, dig, off, rng


def test_stat_wave_matches_lane_per_read_and_oracle(gpu, oracle, monkeypatch):
    """the wave-per-read kernels (default) and the lane-per-read kernels of round 1 (sgk_stat_options_t::kernels = 1) are two
    independent implementations of the same sequential float sums: identical records, and identical to the oracle"""
    reads, dig, off, rng = _adversarial_reads(gpu, 17)
    gpu.stat_configure(2)
    wave = gpu.stat(reads, dig, off, rng)
    gpu.stat_configure(1)
    lane = gpu.stat(reads, dig, off, rng)
    gpu.stat_configure(0)
    for r in range(len(reads)):
        assert wave[r].tobytes() == lane[r].tobytes(), "read %d (n=%d): wave %r lane %r" % (r, reads[r].size, wave[r], lane[r])
    idx = [i for i in range(len(reads)) if reads[i].size > 0]
    with np.errstate(all="ignore"):
        _check_stat(oracle, [reads[i] for i in idx], dig[idx], off[idx], rng[idx], [wave[i] for i in idx])


def test_prefix_wave_matches_lane_per_read(gpu, monkeypatch):
    lens = [100, 2000, 2001, 2500, 20000, 50000, 100000, 100000, 100000, 100000, 30000, 70000]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=23, kind=1)
    for pore in (0, 2):
        gpu.stat_configure(2)
        wave = gpu.prefix(reads, dig, off, rng, 1, pore)
        gpu.stat_configure(1)
        lane = gpu.prefix(reads, dig, off, rng, 1, pore)
        gpu.stat_configure(0)
        for r in range(len(reads)):
            assert wave[r].tobytes() == lane[r].tobytes(), "read %d: wave %r lane %r" % (r, wave[r], lane[r])


def test_jnn_wave_matches_lane_per_read_and_oracle(gpu, oracle, monkeypatch):
    """k_jnn_wave (chunks between data-determined sync points) against k_jnn and the oracle: ragged lengths, reads
    with no sync point at all (one lane ends up running everything), reads that are all in or all out of range"""
    rs = np.random.RandomState(4)
    lens = [1, 2, 100, 127, 128, 129, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000, 5000, 5000, 5000, 5000,
            20000, 65536, 65537, 100000, 100000, 100000, 100000, 250000]
    lens += [int(v) for v in np.clip(np.exp(rs.normal(9.0, 1.0, size=40)), 1, 200000)]
    for kind in (0, 1):
        reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=31 + kind, kind=kind)
        reads = [r.copy() for r in reads]
        reads[14] = np.full(5000, 500, dtype=np.int16)                            # constant: never out of range... or always
        reads[15] = np.where(np.arange(5000) % 2 == 0, 400, 900).astype(np.int16)  # alternating far-apart levels
        reads[16] = (500 + 3 * ((np.arange(5000) // 4) % 2)).astype(np.int16)       # short streaks only: no sync point
        sq = np.where((np.arange(100000) // 700) % 2 == 0, 480, 620) + rs.randint(-3, 4, size=100000)
        reads[23] = sq.astype(np.int16)                                           # long in-range stretches, clean edges
        sm = (520 + 8 * np.sin(np.arange(100000) / 37.0)); sm[::40] += 500         # in range but for every 40th sample:
        reads[22] = np.rint(sm).astype(np.int16)                                  # a segment every 240 samples, NO sync point
        nz = rs.normal(520, 40, size=100000); nz[::7] += 400                      # every 7th sample an outlier
        reads[24] = np.clip(np.rint(nz), 0, 4000).astype(np.int16)
        for rna in (0, 1):
            gpu.stat_configure(2)
            wave = gpu.jnn(reads, dig, off, rng, rna)
            gpu.stat_configure(1)
            lane = gpu.jnn(reads, dig, off, rng, rna)
            gpu.stat_configure(0)
            for r in range(len(reads)):
                np.testing.assert_array_equal(wave[r][0], lane[r][0], err_msg="kind %d rna %d read %d (n=%d) x" % (kind, rna, r, reads[r].size))
                np.testing.assert_array_equal(wave[r][1], lane[r][1], err_msg="kind %d rna %d read %d (n=%d) y" % (kind, rna, r, reads[r].size))
            sub = list(range(0, 27))
            _check_jnn(oracle, [reads[i] for i in sub], rna, [wave[i] for i in sub])


def test_wave_kernels_longest_first_dispatch(gpu, oracle, monkeypatch):
    """>= 1024 reads through the job API: the wave-per-read kernels take the reads in the order of a device-side
    counting sort of their lengths (longest first).  Same records as the lane-per-read kernels, and as the oracle."""
    rs = np.random.RandomState(12)
    nr = 1500
    lens = [int(v) for v in np.clip(np.exp(rs.normal(7.5, 1.3, size=nr)), 0, 60000)]
    lens[7] = 0; lens[8] = 1; lens[9] = 150000
    reads, dig, off, rng = gpu.synth_reads_host(nr, lens, seed=77, kind=1)
    out = {}
    for mode in ("wave", "lane"):
        gpu.stat_configure(1 if mode == "lane" else 2)
        job = gpu.Job(0)
        job.stage(reads, dig, off, rng, None)
        job.launch(gpu.TOOL_STAT); st = job.wait()["stat"]
        job.launch(gpu.TOOL_JNN, rna=1); sg = job.wait()["segs"]
        job.launch(gpu.TOOL_PREFIX, rna=1, pore=0); pf = job.wait()["prefix"]
        out[mode] = (st.copy(), [(x.copy(), y.copy()) for x, y in sg], pf.copy())
        if mode == "wave":  # k_event takes the reads in the same order
            job.launch(gpu.TOOL_EVENT, rna=1)
            ev = job.wait()["events"]
            for i in [i for i in range(0, nr, 41) if lens[i] > 0] + [9]:
                e = oracle.event_raw(reads[i], dig[i], off[i], rng[i], 1)
                g = ev[i]
                assert g.start.size == e.start.size and np.array_equal(g.start.astype(np.uint64), e.start.astype(np.uint64)), "event read %d" % i
                assert np.array_equal(g.mean.view(np.uint32), e.mean.view(np.uint32)), "event means read %d" % i
        job.close()
    gpu.stat_configure(0)
    for r in range(nr):
        assert out["wave"][0][r].tobytes() == out["lane"][0][r].tobytes(), "stat read %d" % r
        np.testing.assert_array_equal(out["wave"][1][r][0], out["lane"][1][r][0], err_msg="jnn x read %d" % r)
        np.testing.assert_array_equal(out["wave"][1][r][1], out["lane"][1][r][1], err_msg="jnn y read %d" % r)
        w, l = out["wave"][2][r], out["lane"][2][r]
        for name in ("adapt_x", "adapt_y", "polya_x", "polya_y"):
            assert int(w[name]) == int(l[name]), "prefix read %d %s" % (r, name)
        if int(w["adapt_y"]) > 0:
            for name in ("adapt_mean", "adapt_std", "adapt_median"):
                assert np.float32(w[name]).view(np.uint32) == np.float32(l[name]).view(np.uint32), "prefix read %d %s" % (r, name)
    sub = [i for i in range(0, nr, 37) if lens[i] > 0] + [9]
    with np.errstate(all="ignore"):
        _check_stat(oracle, [reads[i] for i in sub], dig[sub], off[sub], rng[sub], [out["wave"][0][i] for i in sub])
        _check_jnn(oracle, [reads[i] for i in sub], 1, [out["wave"][1][i] for i in sub])
        _check_prefix(oracle, [reads[i] for i in sub], dig[sub], off[sub], rng[sub], 1, 0, [out["wave"][2][i] for i in sub])


def test_jnn_wave_regression_sync_sample_at_block_end(gpu, oracle):
    """tests/soak_wave_vs_lane.py, seed 5, batches 431 and 797: the sync sample that ends a lane's run was the LAST
    sample of a 32-sample block (position 32 = "whole block" was mistaken for "no end found"), the lane ran on into its
    neighbour's chunk and both reported the segments there; the merge then produced (next start, previous end)."""
    for name in ("soak_wvl_seed5_b431_r1525.npz", "soak_wvl_seed5_b797_r3354.npz"):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
        x = z["samples"].astype(np.int16)
        dig, off, rng = np.array([float(z["dig"])]), np.array([float(z["off"])]), np.array([float(z["rng"])])
        for rna in (0, 1):
            _check_jnn(oracle, [x], rna, gpu.jnn([x], dig, off, rng, rna))


def test_stat_negative_raw_sum_is_exact_and_not_slow(gpu, oracle):
    """reads whose running RAW sum is negative (signed ADC codes) used to fail the fast walk's sign test
    on every tile (term-by-term fallback: exact, 10-50x slower, and with longest-first dispatch one such read sets the
    kernel time).  The raw chain is oriented like the pA chain now: same bits as the oracle, and a batch of such reads
    costs what its mirror image costs."""
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(21)
    n, R = 150000, 96
    pos = [(600 + rs.randint(-60, 61, size=n)).astype(np.int16) for _ in range(R)]
    neg = [(-x).astype(np.int16) for x in pos]
    dig = np.full(R, 8192.0); off = np.full(R, 5.0); rng = np.full(R, 1402.882324)
    # (a chain whose TERMS change sign while the accumulator stays on one side -- first half +600s, second half -600s --
    # is outside the monotone fast walk by construction and is added term by term: exact, checked below, not timed)
    mixed = np.concatenate([pos[1][:n // 2], (-pos[1][n // 2:]).astype(np.int16)])
    zeros = neg[2].copy(); zeros[::2] = 0       # raw chain: +0 terms on the negated chain; pA chain: mixed signs
    hard = [mixed, zeros]
    bm = device.upload_reads(hard, np.full(2, 8192.0), np.full(2, 5.0), np.full(2, 1402.882324), dev)
    recm = device.stat(bm)
    torch.cuda.synchronize()
    _check_stat(oracle, hard, np.full(2, 8192.0), np.full(2, 5.0), np.full(2, 1402.882324),
                np.frombuffer(recm.cpu().numpy().tobytes(), dtype=gpu.STAT_DTYPE)[:2])
    times = {}
    for name, reads in (("pos", pos), ("neg", neg)):
        b = device.upload_reads(reads, dig, off, rng, dev)
        rec = device.stat(b)
        torch.cuda.synchronize()
        got = np.frombuffer(rec.cpu().numpy().tobytes(), dtype=gpu.STAT_DTYPE)[:R]
        _check_stat(oracle, reads[:6] + reads[-2:], dig, off, rng, np.concatenate([got[:6], got[-2:]]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            device.stat(b)
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / 5
    assert times["neg"] < 1.5 * times["pos"] + 0.05, times


@pytest.mark.parametrize("kind", [0, 1])
def test_large_batch_choices_match_the_wave_kernels(gpu, oracle, kind):
    """what kernels = 0 picks for a large batch (stat: one read per lane with the median out of k_moments' second pass;
    prefix: the wave finders + the lane kernels for the region statistics, medians likewise) against the wave kernels
    alone, record for record; reads that defeat the 32-value median window (constant, two-valued, wide) included --
    and against the ORACLE on 64 of the reads, the hostile ones first (VERDICT r04 task 8a: the default for >= 81 920
    reads was only compared GPU against GPU at that size)"""
    import torch
    from sigtk_amd import device
    rs = np.random.RandomState(17 + kind)
    n = 90000
    lens = rs.randint(2500, 4200, size=n).astype(np.int64)
    lens[rs.randint(0, n, size=300)] = rs.randint(0, 12, size=300)       # empty and tiny reads
    dev = torch.device("cuda", 0)
    b = device.synth_reads(n, 0, seed=29 + kind, kind=kind, device=dev, lengths=lens)
    rng = b.rng.cpu().numpy().copy()
    rng[rs.randint(0, n, size=2000)] *= -1.0                              # negative unit: the pA median mirrors the raw ranks
    b.rng.copy_(torch.from_numpy(rng).to(dev))
    host = b.samples.cpu().numpy().copy()
    hostile = [int(r) for r in rs.randint(0, n, size=400)]
    for r in hostile:
        o, m = int(b.offsets_host[r]), int(lens[r])
        u = rs.rand()
        if m == 0: continue
        if u < 0.25: host[o:o + m] = rs.randint(-100, 2000)
        elif u < 0.5: host[o:o + m] = np.where(rs.rand(m) < 0.5, 300, 900)
        elif u < 0.75: host[o:o + m] = rs.randint(-2000, 2000, size=m)
        else: host[o:o + m] = np.where(rs.rand(m) < 0.5, 500, 500 + rs.randint(1, 40))   # the median on a window edge
    b.samples.copy_(torch.from_numpy(host).to(dev))
    out = {}
    for kernels in (0, 2):
        gpu.stat_configure(kernels)
        try:
            st = device.stat(b).cpu().numpy().copy()
            pf = device.prefix(b, kind, 0).cpu().numpy().copy()
        finally:
            gpu.stat_configure(0)
        out[kernels] = (st, pf)
    assert gpu.stat_plan("stat", n, int(lens.sum()), int(lens.max())).kernels == 1
    assert out[0][0].tobytes() == out[2][0].tobytes()
    assert out[0][1].tobytes() == out[2][1].tobytes()
    # the oracle at the size the choice is made at: 44 hostile reads + 20 others
    pick = hostile[:44] + [int(r) for r in rs.randint(0, n, size=20)]
    got = np.frombuffer(out[0][0].tobytes(), dtype=gpu.STAT_DTYPE)[:n]
    dig, off = b.dig.cpu().numpy(), b.off.cpu().numpy()
    reads = [host[int(b.offsets_host[r]):int(b.offsets_host[r]) + int(lens[r])] for r in pick]
    _check_stat(oracle, [x for x in reads if x.size], [dig[r] for r in pick if lens[r]], [off[r] for r in pick if lens[r]],
                [rng[r] for r in pick if lens[r]], [got[r] for r in pick if lens[r]])


def test_each_cut_of_the_choice_is_no_cliff(gpu):
    """VERDICT r04 task 8b/c: kernels = 0 chooses between two implementations by ONE table (sgk_stat_lane_rules =
    LANE_RULES in csrc/stat_args.h, the numbers sgk_stat_plan and the launchers read): a line in (reads, samples per
    read) per tool.  Either side of that line -- 8 % under and over it at several batch sizes, and 64 reads under / at
    min_reads -- both implementations are timed and the one the library picks may be at most 1.12 x the other (+ 50 us:
    batches of a few thousand tiny reads are a tenth of a millisecond either way).
    A line that drifts with a kernel change shows up here instead of as a cliff in somebody's batch.  (Its first run, on
    round 4's five hand-placed steps, found stat at 49 088 x 32 768 on the 30 % slower implementation.)"""
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    names = {0: "stat", 1: "jnn", 3: "stat_pa"}
    pa_buf = torch.empty(int(7e9), dtype=torch.float32, device=dev)

    def timed(fn, kernels):
        gpu.stat_configure(kernels)
        try:
            fn(); fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        finally:
            gpu.stat_configure(0)

    report = []
    for tool, min_reads, max_len in gpu.stat_lane_rules():
        if tool not in names:
            continue     # (tool 4, the region statistics inside prefix, has no kernels switch of its own)
        shapes = [(min_reads - 64, max(512, max_len(min_reads) // 2)), (min_reads, max(512, max_len(min_reads) // 2))]
        for n in (min_reads, 2 * min_reads, 4 * min_reads, 8 * min_reads):
            line = max_len(n)
            if line * 1.08 * n > 6.5e9 or line < 1024:
                continue
            shapes += [(n, int(line * 0.92) // 64 * 64), (n, int(line * 1.08) // 64 * 64 + 64)]
        for n_reads, length in shapes:
            b = device.synth_reads(n_reads, length, seed=5, kind=0, device=dev)
            if tool == 0: fn = lambda: device.stat(b)
            elif tool == 3: fn = lambda: device.stat_pa(b, pa_buf[:b.n_samples])
            else:
                ar = device.SegArena(b)
                fn = lambda: device.jnn(b, ar, 0)
            tl, tw = timed(fn, 1), timed(fn, 2)
            picked = gpu.stat_plan(names[tool], n_reads, b.total_samples, length).kernels
            t_pick, t_other = (tl, tw) if picked == 1 else (tw, tl)
            report.append((names[tool], n_reads, length, round(tl, 3), round(tw, 3), picked))
            assert t_pick <= 1.12 * t_other + 0.05, report[-1]
            del b
            torch.cuda.empty_cache()
    print(report)
