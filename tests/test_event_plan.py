"""CPU: the host-side plan of the event path (sgk_event_plan: no GPU work) -- how a batch's long reads are cut into
segments and how many lanes its short reads get.  The capacities must cover every batch with the given totals."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from sigtk_amd import api
    L = C.CDLL(api.LIB_PATH)
    L.sgk_event_plan.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(api.EventPlan)]
    L.sgk_event_configure.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.sgk_event_configure_short.argtypes = [C.c_int]
    L.sgk_event_workspace_bytes.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
    L.sgk_event_workspace_bytes.restype = C.c_size_t
    yield L
    L.sgk_event_configure(0, 0, 0)
    L.sgk_event_configure_short(0)


def plan(L, lens, rna=0):
    from sigtk_amd import api
    lens = np.asarray(lens, dtype=np.int64)
    n_samples = int(((lens + 7) // 8 * 8).sum())   # reads laid out on 8-sample boundaries
    p = api.EventPlan()
    assert L.sgk_event_plan(len(lens), n_samples, int(lens.max()), rna, C.byref(p)) == 0
    return p, n_samples


def test_defaults_and_no_long_reads(lib):
    lib.sgk_event_configure(0, 0, 0)
    p, _ = plan(lib, [100000] * 10000)
    assert (p.segment_len, p.long_min) == (131072, 262144)
    assert p.max_segments == 0 and p.max_long_reads == 0          # nothing reaches long_min: no lists, no extra kernels
    assert p.lanes_per_short_read == 0 and p.short_max == 16384   # 100 000-sample reads keep their wavefront
    assert plan(lib, [100000] * 10000, rna=1)[0].short_max == 65536


def test_segment_capacities_cover_any_batch_with_these_totals(lib):
    rs = np.random.RandomState(1)
    for seg, lmin in ((0, 0), (1024, 1025), (4096, 10000), (131072, 131073)):
        lib.sgk_event_configure(seg, lmin, 0)
        for _ in range(200):
            n = int(rs.randint(1, 60))
            lens = np.exp(rs.uniform(np.log(1), np.log(3e6), size=n)).astype(np.int64)
            p, n_samples = plan(lib, lens)
            long_reads = lens[lens >= p.long_min]
            segs = int(sum(-(-int(x) // p.segment_len) for x in long_reads))
            if long_reads.size:
                assert p.max_segments >= segs and p.max_long_reads >= long_reads.size, (seg, lmin, lens.tolist())
                assert p.max_long_reads <= n
            else:
                assert p.max_segments == 0
            assert p.long_min > p.segment_len and p.segment_len % 1024 == 0
    lib.sgk_event_configure(0, 0, 0)
    # the workspace grows with the lists, and only then
    a = lib.sgk_event_workspace_bytes(100, 100 * 100000, 100000)
    b = lib.sgk_event_workspace_bytes(100, 100 * 100000, 3000000)
    assert b > a + (100 * 100000 // 131072) * 320


def test_lanes_for_short_reads(lib):
    lib.sgk_event_configure(0, 0, 0)
    lib.sgk_event_configure_short(0)
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
    lib.sgk_event_configure_short(8)
    assert plan(lib, [5000] * 100, rna=0)[0].lanes_per_short_read == 8
    lib.sgk_event_configure_short(-1)
    assert plan(lib, [5000] * 200000, rna=0)[0].lanes_per_short_read == 0
    lib.sgk_event_configure_short(0)
