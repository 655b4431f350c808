"""CPU: the host-side plan of the event path (sgk_event_plan: no GPU work) -- how a batch's long reads are cut into
segments and how many lanes its short reads get.  The capacities must cover every batch with the given totals."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from sigtk_amd import api
    return api.load_library()


def opts(**kw):
    from sigtk_amd import api
    o = api.EventOptions()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def plan(L, lens, rna=0, opt=None):
    from sigtk_amd import api
    lens = np.asarray(lens, dtype=np.int64)
    n_samples = int(((lens + 7) // 8 * 8).sum())   # reads laid out on 8-sample boundaries
    p = api.EventPlan()
    assert L.sgk_event_plan_opt(len(lens), n_samples, int(lens.max()), rna, C.byref(opt if opt is not None else opts()),
                            C.byref(p)) == 0
    return p, n_samples


def test_defaults_and_no_long_reads(lib):
    p, _ = plan(lib, [100000] * 12288, opt=opts(tail_split=-1))
    assert (p.segment_len, p.long_min) == (131072, 262144)
    assert p.max_segments == 0 and p.max_long_reads == 0          # nothing reaches long_min: no lists, no extra kernels
    assert p.tail_segment_len == 0 and p.tail_split_from == 12288
    assert p.lanes_per_short_read == 0 and p.short_max == 16384   # 100 000-sample reads keep their wavefront
    assert plan(lib, [100000] * 10000, rna=1)[0].short_max == 65536
    # a null options pointer is the defaults
    from sigtk_amd import api
    q = api.EventPlan()
    assert lib.sgk_event_plan_opt(10, 10 * 5000, 5000, 0, None, C.byref(q)) == 0 and q.segment_len == 65536
    # the long reads' geometry follows the batch (api.hip: event_config_for): long_min = 0.9 x the batch's samples per
    # wavefront slot, between 131 072 and 262 144, segments of half of it; an explicit segment_len / long_min is kept
    assert [(plan(lib, [100000] * n)[0].long_min, plan(lib, [100000] * n)[0].segment_len) for n in (3000, 6000, 10000)] == \
        [(131072, 65536), (176128, 88064), (262144, 131072)]
    assert plan(lib, [100000] * 10000, rna=1)[0].long_min == 262144 and plan(lib, [100000] * 3000, rna=1)[0].long_min == 133120
    assert plan(lib, [100000] * 3000, opt=opts(long_min=200000))[0].long_min == 200000


def test_segment_capacities_cover_any_batch_with_these_totals(lib):
    rs = np.random.RandomState(1)
    for seg, lmin in ((0, 0), (1024, 1025), (4096, 10000), (131072, 131073)):
        o = opts(segment_len=seg, long_min=lmin, tail_split=-1)
        for _ in range(200):
            n = int(rs.randint(1, 60))
            lens = np.exp(rs.uniform(np.log(1), np.log(3e6), size=n)).astype(np.int64)
            p, n_samples = plan(lib, lens, opt=o)
            long_reads = lens[lens >= p.long_min]
            segs = int(sum(-(-int(x) // p.segment_len) for x in long_reads))
            if long_reads.size:
                assert p.max_segments >= segs and p.max_long_reads >= long_reads.size, (seg, lmin, lens.tolist())
                assert p.max_long_reads <= n
            else:
                assert p.max_segments == 0
            assert p.long_min > p.segment_len and p.segment_len % 1024 == 0
    # the workspace grows with the lists, and only then
    a = lib.sgk_event_workspace_bytes(100, 100 * 100000, 100000)
    b = lib.sgk_event_workspace_bytes(100, 100 * 100000, 3000000)
    assert b > a + (100 * 100000 // 131072) * 320


def test_lanes_for_short_reads(lib):
    # 200 000 x 5 000: chunks of >= 8 warm-ups (32 samples with DNA parameters, 128 with RNA parameters)
    assert plan(lib, [5000] * 200000, rna=0)[0].lanes_per_short_read == 16
    assert plan(lib, [5000] * 200000, rna=1)[0].lanes_per_short_read == 4
    # a small batch wants every lane it can get: >= 4 rounds of waves or no packing at all
    assert plan(lib, [5000] * 20000, rna=1)[0].lanes_per_short_read == 0
    assert plan(lib, [5000] * 50000, rna=1)[0].lanes_per_short_read == 16
    # reads at or above the threshold are not packed, whatever their number
    assert plan(lib, [20000] * 200000, rna=0)[0].lanes_per_short_read == 0
    assert plan(lib, [20000] * 200000, rna=1)[0].lanes_per_short_read in (8, 16)
    assert plan(lib, [70000] * 200000, rna=1)[0].lanes_per_short_read == 0
    # a ragged batch of >= 1024 reads has a dispatch order: its short reads are packed even if some reads are long
    lens = [4000] * 150000 + [300000, 50000]
    assert plan(lib, lens, rna=0)[0].lanes_per_short_read > 0
    # ... a ragged batch without one (< 1024 reads) is not
    assert plan(lib, [4000] * 500 + [50000], rna=0)[0].lanes_per_short_read == 0
    # forced / off
    assert plan(lib, [5000] * 100, rna=0, opt=opts(lanes_per_short_read=8))[0].lanes_per_short_read == 8
    assert plan(lib, [5000] * 200000, rna=0, opt=opts(lanes_per_short_read=-1))[0].lanes_per_short_read == 0


def test_tail_split(lib):
    """a batch of fewer than 8 rounds of wavefronts (256 CUs x 4 SIMDs x 3 waves with the DNA preset, x 2 with RNA
    parameters; without a GPU the plan assumes 256 CUs): the reads of its last, partial round are cut into segments where
    that was measured to pay (api.hip: event_tail_plan, profiles/r05_tail_split_sweep.txt) -- a batch of at most 0.55 of a
    round (RNA preset: 7 / 8), a last round of at most a quarter of a round behind one full round, a sixth behind more
    (RNA preset: never); a cut read costs 1.5 x a whole one"""
    p, _ = plan(lib, [100000] * 9300)                   # 9 300 = 3 x 3072 + 84: the last 84 reads, 16 384-sample segments
    assert (p.tail_split_from, p.tail_segment_len) == (9216, 16384)
    assert p.max_segments >= 84 * 7 and p.max_long_reads >= 84
    assert plan(lib, [100000] * 10000)[0].tail_segment_len == 0             # 784 reads over: a quarter of a round, not cut
    assert plan(lib, [100000] * 2000)[0].tail_segment_len == 0              # two thirds of a round: not cut
    p, _ = plan(lib, [100000] * 1000)                   # a third of a round: 3 segments each fill it
    assert p.tail_split_from == 0 and p.tail_segment_len == 33792
    p, _ = plan(lib, [100000] * 160)                    # a CLI-sized batch: at most 8 segments per read
    assert p.tail_split_from == 0 and p.tail_segment_len == 16384 + 1024 * 0 or p.tail_segment_len >= 12288
    assert plan(lib, [100000] * 12288)[0].tail_segment_len == 0             # whole rounds
    assert plan(lib, [100000] * 12000)[0].tail_segment_len == 0             # the last round is nearly full
    assert plan(lib, [100000] * 30000)[0].tail_segment_len == 0             # >= 8 rounds: the tail does not matter
    assert plan(lib, [100000] * 10000, rna=1)[0].tail_segment_len == 0      # 2 048 slots: 4.88 rounds, nearly full
    assert plan(lib, [100000] * 8400, rna=1)[0].tail_segment_len == 0       # 4 x 2048 + 208: RNA preset, more than a round
    assert plan(lib, [100000] * 1700, rna=1)[0].tail_segment_len > 0        # 0.83 of a round of the RNA preset
    assert plan(lib, [100000] * 1900, rna=1)[0].tail_segment_len == 0
    assert plan(lib, [100000] * 1600)[0].tail_segment_len > 0               # 0.52 of a round
    assert plan(lib, [100000] * 1800)[0].tail_segment_len == 0
    assert plan(lib, [100000] * 3800)[0].tail_segment_len > 0               # 728 reads behind ONE round: a quarter
    assert plan(lib, [100000] * 3900)[0].tail_segment_len == 0
    assert plan(lib, [100000] * 6900)[0].tail_segment_len == 0              # 756 behind two: a sixth
    assert plan(lib, [20000] * 10000)[0].tail_segment_len == 0              # short reads: not worth two kernels more
    assert plan(lib, [100000] * 9300, opt=opts(tail_split=-1))[0].tail_segment_len == 0
    p = plan(lib, [100000] * 30000, opt=opts(tail_split=2000))[0]      # the caller's number, whatever the batch
    assert (p.tail_split_from, p.tail_segment_len) == (28000, 50176)
    # the workspace has room for the lists
    from sigtk_amd import api
    lib.sgk_event_workspace_bytes_opt.restype = C.c_size_t
    a = lib.sgk_event_workspace_bytes_opt(9300, 93 * 10 ** 7, 100000, C.byref(opts(tail_split=-1)))
    b = lib.sgk_event_workspace_bytes_opt(9300, 93 * 10 ** 7, 100000, C.byref(opts()))
    assert b >= a + 84 * 7 * 320


def test_the_0_1_0_plan_call_keeps_its_signature(lib):
    """0.2.0 / 0.2.1 had put the six-argument plan under the 0.1.0 symbol; it is sgk_event_plan_opt now and
    sgk_event_plan is the five-argument call again: the defaults, and only the 32 bytes sgk_event_plan_t had then"""
    from sigtk_amd import api
    buf = (C.c_uint32 * 12)(*([0xdeadbeef] * 12))
    assert lib.sgk_event_plan(10, 10 * 5000, 5000, 0, C.cast(buf, C.c_void_p)) == 0
    q = api.EventPlan()
    assert lib.sgk_event_plan_opt(10, 10 * 5000, 5000, 0, None, C.byref(q)) == 0
    assert (buf[0], buf[1], buf[4], buf[5]) == (q.segment_len, q.long_min, q.short_max, q.lanes_per_short_read)
    assert all(buf[k] == 0xdeadbeef for k in range(8, 12))   # nothing behind the 32 bytes is touched
    assert lib.sgk_event_plan(10, 50000, 5000, 0, None) != 0
