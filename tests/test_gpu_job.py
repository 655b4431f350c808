"""GPU: the pipelined host jobs (sgk_job_*, include/sigtk_gpu.h) -- every subtool through a job, fed with
int16 samples and with svb-zd blobs (decoded on the device), against the oracle; jobs are recycled
across batches of different sizes and several jobs are in flight at once."""
import numpy as np
import pytest

from sigtk_amd import blow5

pytestmark = pytest.mark.gpu


def _same_events(got, exp, what):
    assert got.start.size == exp.start.size, "%s: %d events, oracle %d" % (what, got.start.size, exp.start.size)
    assert np.array_equal(got.start.astype(np.uint64), exp.start.astype(np.uint64)), what
    assert np.array_equal(got.length.astype(np.uint64), exp.length.astype(np.uint64)), what
    assert np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32)), what
    assert np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32)), what


def _batch(gpu, lens, seed, kind):
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=seed, kind=kind)
    return reads, dig, off, rng


@pytest.mark.parametrize("svb", [False, True])
def test_all_subtools_through_a_recycled_job(gpu, oracle, svb):
    job = gpu.Job(0)
    # three batches of different geometry through the SAME job: buffers grow, then are reused
    for b, lens in enumerate(([5000, 100000, 333, 70001], [100000] * 6 + [0, 1, 250], [64, 4096])):
        kind = b % 2
        reads, dig, off, rng = _batch(gpu, lens, 20 + b, kind)
        sig = [blow5.svb_zd_encode(r) for r in reads] if svb else reads
        counts = [r.size for r in reads] if svb else None

        job.submit(gpu.TOOL_EVENT, sig, dig, off, rng, rna=kind, counts=counts)
        res = job.wait()
        for r, raw in enumerate(reads):
            if raw.size == 0:
                assert res["events"][r].start.size == 0
                continue
            _same_events(res["events"][r], oracle.event_raw(raw, dig[r], off[r], rng[r], kind), "event read %d" % r)

        job.submit(gpu.TOOL_EVENT, sig, dig, off, rng, rna=kind, flags=gpu.JOB_EVENTS_COMPACT, counts=counts)
        res = job.wait()
        for r, raw in enumerate(reads):
            if raw.size:
                exp = oracle.event_raw(raw, dig[r], off[r], rng[r], kind)
                assert np.array_equal(res["events"][r].length.astype(np.uint64), exp.length.astype(np.uint64))
                assert res["events"][r].mean.size == 0   # not copied back in compact mode

        # the lengths alone (what `event -c` prints): the starts are their running sums -- events are contiguous from 0
        job.submit(gpu.TOOL_EVENT, sig, dig, off, rng, rna=kind, flags=gpu.JOB_EVENTS_LENGTHS, counts=counts)
        res = job.wait()
        for r, raw in enumerate(reads):
            if raw.size:
                exp = oracle.event_raw(raw, dig[r], off[r], rng[r], kind)
                assert np.array_equal(res["events"][r].length.astype(np.uint64), exp.length.astype(np.uint64))
                assert np.array_equal(res["events"][r].start.astype(np.uint64), exp.start.astype(np.uint64))
                assert res["events"][r].mean.size == 0

        job.submit(gpu.TOOL_STAT, sig, dig, off, rng, counts=counts)
        st = job.wait()["stat"]
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = oracle.stat(raw, dig[r], off[r], rng[r])
            assert int(st[r]["raw_median"]) == e[4]
            for name, ev in (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]), ("pa_median", e[5])):
                assert np.float32(st[r][name]).view(np.uint32) == np.float32(ev).view(np.uint32), (r, name)

        job.submit(gpu.TOOL_JNN, sig, dig, off, rng, rna=kind, counts=counts)
        segs = job.wait()["segs"]
        for r, raw in enumerate(reads):
            ex, ey = oracle.jnn_raw(raw, kind)
            assert np.array_equal(segs[r][0].astype(np.int64), ex) and np.array_equal(segs[r][1].astype(np.int64), ey)

        job.submit(gpu.TOOL_PA, sig, dig, off, rng, counts=counts)
        pa = job.wait()["pa"]
        for r, raw in enumerate(reads):
            assert np.array_equal(pa[r].view(np.uint32), oracle.pa(raw, dig[r], off[r], rng[r]).view(np.uint32))

        job.submit(gpu.TOOL_PREFIX, sig, dig, off, rng, rna=kind, pore=0, counts=counts)
        pf = job.wait()["prefix"]
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = oracle.prefix(raw, dig[r], off[r], rng[r], kind, 0)
            assert (int(pf[r]["adapt_x"]), int(pf[r]["adapt_y"])) == (e.adapt_x, e.adapt_y), r
            assert (int(pf[r]["polya_x"]), int(pf[r]["polya_y"])) == (e.polya_x, e.polya_y), r
            if e.adapt_y > 0:
                assert np.float32(pf[r]["adapt_mean"]).view(np.uint32) == np.float32(e.adapt_mean).view(np.uint32)
    job.close()


def test_jobs_in_flight_concurrently(gpu, oracle):
    jobs = [gpu.Job(0) for _ in range(3)]
    batches = [_batch(gpu, [30000 + 1000 * k] * 4, 40 + k, 0) for k in range(3)]
    for j, (reads, dig, off, rng) in zip(jobs, batches):
        j.submit(gpu.TOOL_EVENT, reads, dig, off, rng)        # all three enqueued before any wait
    for j, (reads, dig, off, rng) in zip(jobs, batches):
        res = j.wait()
        for r, raw in enumerate(reads):
            _same_events(res["events"][r], oracle.event_raw(raw, dig[r], off[r], rng[r], 0), "read %d" % r)
    for j in jobs:
        j.close()


def test_malformed_blob_is_reported(gpu):
    reads, dig, off, rng = _batch(gpu, [5000, 5000], 7, 0)
    blobs = [blow5.svb_zd_encode(r) for r in reads]
    blobs[1] = blobs[1][:-10]                                  # truncated data bytes
    job = gpu.Job(0)
    job.submit(gpu.TOOL_STAT, blobs, dig, off, rng, counts=[5000, 5000])
    with pytest.raises(Exception, match="malformed compressed signal"):
        job.wait()
    # the job stays usable
    job.submit(gpu.TOOL_STAT, reads, dig, off, rng)
    assert job.wait()["stat"].size == 2
    job.close()


def test_empty_batch_and_misuse(gpu):
    job = gpu.Job(0)
    job.submit(gpu.TOOL_STAT, [], [], [], [])
    assert job.wait()["n_reads"] == 0
    with pytest.raises(Exception):
        job.wait()                                             # nothing submitted
    job.close()


def test_many_tiny_reads_and_one_huge_read(gpu, oracle):
    """shape extremes through one job: 20 000 reads of 200..260 samples (slot arithmetic, per-read overheads) and a
    single 3 000 000-sample read (chunk length, 32-bit positions)"""
    rs = np.random.RandomState(12)
    lens = rs.randint(200, 261, size=20000).tolist()
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=77, kind=0)
    job = gpu.Job(0)
    job.submit(gpu.TOOL_EVENT, reads, dig, off, rng)
    ev = job.wait()["events"]
    job.launch(gpu.TOOL_STAT)
    st = job.wait()["stat"]
    for r in list(range(0, 20000, 997)) + [19999]:
        _same_events(ev[r], oracle.event_raw(reads[r], dig[r], off[r], rng[r], 0), "tiny read %d" % r)
        e = oracle.stat(reads[r], dig[r], off[r], rng[r])
        assert int(st[r]["raw_median"]) == e[4]
        assert np.float32(st[r]["pa_std"]).view(np.uint32) == np.float32(e[3]).view(np.uint32)
    big, dig, off, rng = gpu.synth_reads_host(1, [3000000], seed=78, kind=1)
    for rna in (0, 1):
        job.submit(gpu.TOOL_EVENT, big, dig, off, rng, rna=rna)
        _same_events(job.wait()["events"][0], oracle.event_raw(big[0], dig[0], off[0], rng[0], rna), "huge read rna=%d" % rna)
    job.launch(gpu.TOOL_JNN, rna=1)
    x, y = job.wait()["segs"][0]
    ex, ey = oracle.jnn_raw(big[0], 1)
    assert np.array_equal(x.astype(np.int64), ex) and np.array_equal(y.astype(np.int64), ey)
    job.launch(gpu.TOOL_PREFIX, rna=1, pore=0)
    p = job.wait()["prefix"][0]
    e = oracle.prefix(big[0], dig[0], off[0], rng[0], 1, 0)
    assert (int(p["adapt_x"]), int(p["adapt_y"]), int(p["polya_x"]), int(p["polya_y"])) == (e.adapt_x, e.adapt_y, e.polya_x, e.polya_y)
    job.close()


def test_two_threads_with_different_geometries_on_one_device(gpu, oracle):
    """VERDICT r03 task 7: the library keeps no process-wide configuration.  Two threads drive jobs on the same device at
    the same time, one with 1024-sample segments and a 16-sample warm-up (hundreds of seams, speculation failing at
    every few of them) and packed short reads, the other with the defaults; both get the oracle's events, and each
    job's status shows ITS geometry."""
    import threading
    lens = [40000, 5000, 3000, 70000, 12000, 900, 20000, 0, 1500]
    reads, dig, off, rng = _batch(gpu, lens, 91, 0)
    exp = [oracle.event_raw(r, dig[i], off[i], rng[i], 0) if r.size else None for i, r in enumerate(reads)]
    small = gpu.EventOptions()
    small.segment_len, small.long_min, small.warmup, small.lanes_per_short_read = 1024, 1025, 16, 4
    default = gpu.EventOptions()
    default.tail_split = -1
    errors, seen = [], {}

    def worker(name, opt):
        try:
            job = gpu.Job(0)
            job.set_options(opt)
            for _ in range(6):
                job.submit(gpu.TOOL_EVENT, reads, dig, off, rng, rna=0)
                res = job.wait()
                for i, e in enumerate(exp):
                    if e is None:
                        assert res["events"][i].start.size == 0
                    else:
                        _same_events(res["events"][i], e, "%s read %d" % (name, i))
                seen[name] = int(res["status"].n_split_reads)
            job.close()
        except Exception as ex:   # noqa: BLE001 -- reported below, in the main thread
            errors.append("%s: %r" % (name, ex))

    ts = [threading.Thread(target=worker, args=("small", small)), threading.Thread(target=worker, args=("default", default))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert seen["small"] == sum(1 for n in lens if n >= 1025) and seen["default"] == 0
