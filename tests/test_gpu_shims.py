"""GPU parity: the per-read shims with the reference's own signatures (include/sigtk_gpu.h, "per-read shims";
src/jnn.h:104-109, src/stat.h:17-73), each diffed against the oracle through ctypes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(gpu, kind, lens, seed):
    return gpu.synth_reads_host(len(lens), lens, seed=seed, kind=kind)


@pytest.mark.parametrize("kind", [0, 1])
def test_stat_shims(gpu, oracle, kind):
    reads, dig, off, rng = _reads(gpu, kind, [1, 2, 777, 20000, 100000], 41)
    for r, raw in enumerate(reads):
        m, s, med = gpu.shim_stat_i16(raw)
        exp = oracle.stat(raw, dig[r], off[r], rng[r])
        assert np.float32(m).view(np.uint32) == np.float32(exp[0]).view(np.uint32)
        assert np.float32(s).view(np.uint32) == np.float32(exp[2]).view(np.uint32)
        assert med == exp[4]
        pa = oracle.pa(raw, dig[r], off[r], rng[r])
        got = gpu.shim_stat_f32(pa)
        want = oracle.statf(pa)
        for g, w in zip(got, want):
            assert np.float32(g).view(np.uint32) == np.float32(w).view(np.uint32)
    # negative values and ties in the float median
    x = np.array([3.5, -2.0, 3.5, 0.0, -7.25, 1e-3, 3.5, -2.0], dtype=np.float32)
    assert [np.float32(v).view(np.uint32) for v in gpu.shim_stat_f32(x)] == \
           [np.float32(v).view(np.uint32) for v in oracle.statf(x)]


@pytest.mark.parametrize("kind", [0, 1])
def test_jnn_raw_and_pa_shims_any_parameters(gpu, oracle, kind):
    reads, dig, off, rng = _reads(gpu, kind, [3000, 40000, 100000], 43)
    params = [dict(std_scale=0.75, corrector=50, seg_dist=50, window=150, stall_len=0.25, error=5),     # cDNA preset
              dict(std_scale=0.75, corrector=50, seg_dist=50, window=1000, stall_len=1.0, error=5),     # dRNA preset
              dict(std_scale=0.4, corrector=17, seg_dist=120, window=60, stall_len=0.5, error=2),       # none of them
              dict(std_scale=-1.0, corrector=50, seg_dist=200, window=250, stall_len=1.0, error=30, top=560.0, bot=420.0)]
    for r, raw in enumerate(reads):
        clamped = np.clip(raw, 0, 1200).astype(np.float32)              # rm_outlier, src/jnn.c:61-77
        pa = oracle.pa(raw, dig[r], off[r], rng[r])
        for kw in params:
            po = oracle.jnn_param(**kw)
            pg = gpu.JnnParam(po.std_scale, po.corrector, po.seg_dist, po.window, po.stall_len, po.error, po.top, po.bot)
            ex, ey = oracle.jnn_core(clamped, po)
            assert gpu.shim_jnn_raw(raw, pg) == list(zip(ex.tolist(), ey.tolist()))
            kwp = dict(kw)
            if kw["std_scale"] < 0:
                kwp.update(top=float(np.median(pa)) + 8.0, bot=float(np.median(pa)) - 8.0)
            po = oracle.jnn_param(**kwp)
            pg = gpu.JnnParam(po.std_scale, po.corrector, po.seg_dist, po.window, po.stall_len, po.error, po.top, po.bot)
            ex, ey = oracle.jnn_pa(pa, po)
            assert gpu.shim_jnn_pa(pa, pg) == list(zip(ex.tolist(), ey.tolist()))
    assert gpu.shim_jnn_raw(np.zeros(0, dtype=np.int16), gpu.JnnParam(0.75, 50, 50, 150, 0.25, 5, 0, 0)) == []


def test_adaptor_and_polya_shims(gpu, oracle):
    reads, dig, off, rng = _reads(gpu, 1, [1500, 30000, 100000, 100000], 47)
    for r, raw in enumerate(reads):
        for pore in (0, 2):
            assert gpu.shim_find_adaptor(raw, pore) == oracle.find_adaptor(raw, pore)
        p9 = gpu.Jnnv2Param(0.5, 1500, 2000, 0.0, 200000, 2000)        # JNNV2_RNA_R9_ADAPTOR, src/jnn.h:84-90
        (xy, rc) = gpu.shim_jnnv2(raw, p9)
        assert rc == 0 and xy == oracle.find_adaptor(raw, 0)
        pa = oracle.pa(raw, dig[r], off[r], rng[r])
        ax, ay = oracle.find_adaptor(raw, 0)
        if ay > 0:
            m_a = float(oracle.statf(pa[ax:ay])[0])
            tail = pa[ay:]
            assert gpu.shim_find_polya(tail, m_a + 50.0, m_a + 10.0, 0) == oracle.find_polya(tail, m_a + 50.0, m_a + 10.0, 0)
    # a window the kernels do not implement is refused, not approximated
    (xy, rc) = gpu.shim_jnnv2(reads[2], gpu.Jnnv2Param(0.5, 1500, 1000, 0.0, 200000, 2000))
    assert xy == (-1, -1) and rc != 0


def test_seqsum_chains_on_hostile_float_arrays(gpu, oracle):
    """sgk_meanf / sgk_stdvf run the reference's sequential float sums through seqsum.h (surrogate starts, parity maps,
    binade crossings, native fallbacks) on ARBITRARY float input: mixed signs, cancellation, ties at every step, sums
    through zero, denormals, overflow to inf, NaN.  Bit-identical to the oracle's plain loops."""
    rs = np.random.RandomState(21)
    arrays = []
    for n in (1, 2, 15, 16, 17, 255, 256, 257, 1023, 1024, 1025, 4999, 70000, 300000):
        arrays.append(rs.normal(90, 12, size=n).astype(np.float32))
        arrays.append(rs.normal(0, 50, size=n).astype(np.float32))                 # sum wanders through zero
        arrays.append((-rs.gamma(2.0, 30.0, size=n)).astype(np.float32))            # all negative
    arrays.append(np.full(200000, 333.0, dtype=np.float32))                        # integer sum beyond 2^24: ties
    arrays.append(np.where(np.arange(100000) % 2 == 0, 0.5, 1.5).astype(np.float32))
    arrays.append(np.concatenate([np.zeros(3000), np.full(500, 7.25), np.zeros(3000), np.full(4000, -7.25)]).astype(np.float32))
    big = rs.normal(100, 10, size=50000).astype(np.float32); big[20000] = 3e7; big[30000] = -3e7; big[40000] = 1e-30
    arrays.append(big)
    arrays.append((rs.rand(5000) * 1e-38).astype(np.float32))                       # denormal sums
    arrays.append((rs.rand(5000) * 3e38).astype(np.float32))                        # overflows to inf
    nn = rs.normal(100, 10, size=5000).astype(np.float32); nn[2500] = np.nan
    arrays.append(nn)
    ii = rs.normal(100, 10, size=5000).astype(np.float32); ii[2500] = np.inf
    arrays.append(ii)
    ij = ii.copy(); ij[3000] = -np.inf
    arrays.append(ij)
    with np.errstate(all="ignore"):
        for k, x in enumerate(arrays):
            g = gpu.shim_stat_f32(x)
            e = oracle.statf(x)
            for name, a, b in (("mean", g[0], e[0]), ("std", g[1], e[1]), ("median", g[2], e[2])):
                a, b = np.float32(a), np.float32(b)
                assert (np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32), \
                    "array %d (n=%d) %s: gpu %r oracle %r" % (k, x.size, name, a, b)


def test_per_read_shims_on_a_long_read(gpu, oracle):
    """a read of 700 001 samples through the reference-signature per-read calls: they take the long-read path of the
    batch API (k_long_chains) and give the reference's segments / adaptor"""
    for kind in (0, 1):
        reads, dig, off, rng = _reads(gpu, kind, [700001], 53 + kind)
        raw = reads[0]
        clamped = np.clip(raw, 0, 1200).astype(np.float32)
        for kw in (dict(std_scale=0.75, corrector=50, seg_dist=50, window=150, stall_len=0.25, error=5),
                   dict(std_scale=0.75, corrector=50, seg_dist=50, window=1000, stall_len=1.0, error=5)):
            po = oracle.jnn_param(**kw)
            pg = gpu.JnnParam(po.std_scale, po.corrector, po.seg_dist, po.window, po.stall_len, po.error, po.top, po.bot)
            ex, ey = oracle.jnn_core(clamped, po)
            assert gpu.shim_jnn_raw(raw, pg) == list(zip(ex.tolist(), ey.tolist()))
        for pore in (0, 2):
            assert gpu.shim_find_adaptor(raw, pore) == oracle.find_adaptor(raw, pore)
