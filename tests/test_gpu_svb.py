"""GPU: svb-zd signal decode on the device against the host decoders (bit-exact; integer work)."""
import os
import struct

import numpy as np
import pytest

from sigtk_amd import blow5

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _decode(gpu, blobs, counts):
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    reads, status = device.svbzd_decode(blobs, counts, dev)
    torch.cuda.synchronize()
    out = []
    for r in range(len(blobs)):
        o = int(reads.offsets_host[r]); n = int(reads.lengths_host[r])
        out.append(reads.samples[o:o + n].cpu().numpy())
    return out, status[:len(blobs)].cpu().numpy()


def test_reference_written_blobs(gpu, sp1):
    """the signal blobs of the reference's bundled fixture (written by ONT/slow5 tooling, not by us)"""
    recs = blow5.read_signal_blobs(os.path.join(GOLDEN, "sp1_dna.blow5"))
    blobs = [b for _, b in recs]
    counts = [r.raw.size for r in sp1.reads]
    got, status = _decode(gpu, blobs, counts)
    assert (status == 0).all()
    for g, r in zip(got, sp1.reads):
        assert np.array_equal(g, r.raw)


def test_all_code_lengths_and_ragged_counts(gpu):
    rs = np.random.RandomState(5)
    arrays = []
    for n in (0, 1, 2, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4096, 100000):
        arrays.append((500 + rs.randint(-40, 40, size=n)).astype(np.int16))            # 1-byte deltas
    arrays.append(rs.randint(-32768, 32767, size=5000).astype(np.int16))               # 2- and 3-byte deltas
    arrays.append(np.where(rs.rand(3000) < 0.5, -32768, 32767).astype(np.int16))       # extreme jumps
    arrays.append(np.zeros(2048, dtype=np.int16))
    blobs = [blow5.svb_zd_encode(a) for a in arrays]
    got, status = _decode(gpu, blobs, [a.size for a in arrays])
    assert (status == 0).all()
    for g, a in zip(got, arrays):
        assert np.array_equal(g, a)
    # a blob containing 4-byte codes (values >= 2^24 never come from int16 data, but the format allows them)
    vals = np.array([1 << 25, 3, (1 << 31) + 7, 255, 256, 65535, 65536, 1 << 24], dtype=np.uint32)
    codes = [3, 0, 3, 0, 1, 1, 2, 3]
    keys = bytes([sum(c << (2 * j) for j, c in enumerate(codes[0:4])), sum(c << (2 * j) for j, c in enumerate(codes[4:8]))])
    data = b"".join(int(v).to_bytes(c + 1, "little") for v, c in zip(vals, codes))
    blob = struct.pack("<I", 8) + keys + data
    exp = blow5.svb_zd_decode(blob)
    got, status = _decode(gpu, [blob], [8])
    assert status[0] == 0 and np.array_equal(got[0], exp)


def test_corrupt_blobs_are_reported(gpu):
    a = (500 + np.arange(3000) % 50).astype(np.int16)
    good = blow5.svb_zd_encode(a)
    got, status = _decode(gpu, [good, good[:-5], good + b"\x00\x00", good], [3000, 3000, 3000, 2999])
    assert list(status) == [0, 2, 2, 1]
    assert np.array_equal(got[0], a)


def test_corrupted_blobs_are_rejected_not_trusted(gpu):
    """bit flips, truncations, trailing junk and wrong count words: every batch is either rejected (per-read status
    -> SGK_ERR_FORMAT from sgk_job_wait) or was still a well-formed stream; the decoder never reads or writes
    outside the blob / the read's sample range, and the job stays usable."""
    import random
    rnd = random.Random(3)
    reads, dig, off, rng = gpu.synth_reads_host(6, [5000, 1, 17, 4096, 30000, 1023], 5, 0)
    good = [blow5.svb_zd_encode(r) for r in reads]
    counts = [r.size for r in reads]
    job = gpu.Job(0)
    rejected = 0
    for it in range(200):
        blobs = [bytearray(b) for b in good]
        k = rnd.randrange(len(blobs))
        mode = it % 4
        if mode == 0 and len(blobs[k]) > 5:
            blobs[k][rnd.randrange(4, len(blobs[k]))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            blobs[k] = blobs[k][: rnd.randrange(0, len(blobs[k]))]
        elif mode == 2:
            blobs[k] = blobs[k] + bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 9)))
        else:
            blobs[k][0:4] = int(rnd.randrange(0, 70000)).to_bytes(4, "little")
        try:
            job.submit(gpu.TOOL_STAT, [bytes(b) for b in blobs], dig, off, rng, counts=counts)
            job.wait()
        except gpu.SigtkGpuError as e:
            assert "malformed compressed signal" in str(e)
            rejected += 1
    assert rejected > 100
    job.submit(gpu.TOOL_STAT, good, dig, off, rng, counts=counts)
    assert job.wait()["stat"].size == 6
    job.close()
