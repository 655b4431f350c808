"""GPU: `qts` pieces -- the per-sample quantisers of src/qts.c and the svb-zd encoder, whose output
must equal slow5lib's blobs byte for byte (streamvbyte's encoding is canonical)."""
import os

import numpy as np
import pytest

from sigtk_amd import blow5

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(arrays):
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    lens = np.array([a.size for a in arrays], dtype=np.int64)
    b = device.alloc_reads(lens, dev)
    host = np.full(b.n_samples, 12345, dtype=np.int16)   # gaps are arbitrary data
    for r, a in enumerate(arrays):
        o = int(b.offsets_host[r]); host[o:o + a.size] = a
    b.samples.copy_(torch.from_numpy(host).to(dev))
    return b


def test_encoder_reproduces_the_reference_files_blobs(gpu, sp1):
    """decode -> encode of the blobs in the reference's bundled BLOW5 (written by slow5lib) is the identity"""
    from sigtk_amd import device
    recs = blow5.read_signal_blobs(os.path.join(GOLDEN, "sp1_dna.blow5"))
    b = _load([r.raw for r in sp1.reads])
    got = device.svbzd_encode(b)
    for g, (_, blob) in zip(got, recs):
        assert g == blob


def test_encoder_all_code_lengths_ragged_and_empty(gpu):
    from sigtk_amd import device
    rs = np.random.RandomState(8)
    arrays = [np.zeros(0, dtype=np.int16)]
    for n in (1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1027, 4096, 100003):
        arrays.append((500 + rs.randint(-40, 40, size=n)).astype(np.int16))                  # 1-byte codes
    arrays.append(rs.randint(-32768, 32767, size=7001).astype(np.int16))                     # 2- and 3-byte codes
    arrays.append(np.where(rs.rand(3001) < 0.5, -32768, 32767).astype(np.int16))             # extreme jumps
    arrays.append(np.zeros(2050, dtype=np.int16))
    got = device.svbzd_encode(_load(arrays))
    for g, a in zip(got, arrays):
        assert g == blow5.svb_zd_encode(a), "n = %d" % a.size
        assert np.array_equal(blow5.svb_zd_decode(g), a)


@pytest.mark.parametrize("method,name", [(0, "floor"), (1, "round"), (2, "fill-ones")])
@pytest.mark.parametrize("bits", [1, 2, 5])
def test_quantisers_match_qts_c(gpu, method, name, bits):
    """src/qts.c:126-142 on int16 (int arithmetic, truncated back to int16 on assignment)"""
    import torch
    from sigtk_amd import device
    rs = np.random.RandomState(method * 10 + bits)
    arrays = [rs.randint(-32768, 32767, size=n).astype(np.int16) for n in (1, 100, 8191, 8192, 8193, 50000)]
    arrays.append(np.array([32767, 32766, -32768, -1, 0, 1], dtype=np.int16))               # wrap-around at the top
    b = _load(arrays)
    device.qts(b, bits, method)
    torch.cuda.synchronize()
    host = b.samples.cpu().numpy()
    for r, a in enumerate(arrays):
        x = a.astype(np.int64)
        if name == "floor":
            e = (x >> bits) << bits
        elif name == "fill-ones":
            e = x | ((1 << bits) - 1)
        else:
            mask = (1 << bits) - 1
            lsb = x & mask
            e = np.where(lsb < (1 << (bits - 1)), x & ~mask, (x & ~mask) + (1 << bits))
        e = e.astype(np.int16)      # C: int -> int16_t assignment wraps
        o = int(b.offsets_host[r])
        assert np.array_equal(host[o:o + a.size], e), (name, bits, r)
    assert host[0] == 12345        # nothing outside the reads was touched


@pytest.mark.parametrize("svb_in,svb_out", [(False, True), (True, True), (True, False)])
def test_qts_through_a_job(gpu, svb_in, svb_out):
    """sgk_job_submit_qts: (decode ->) quantise -> (encode) with the blobs laid out on the device"""
    lens = [0, 1, 5, 4096, 30001, 100000]
    reads, dig, off, rng = gpu.synth_reads_host(len(lens), lens, seed=41, kind=0)
    job = gpu.Job(0)
    job.stage([blow5.svb_zd_encode(r) for r in reads] if svb_in else reads, dig, off, rng,
              counts=[r.size for r in reads] if svb_in else None)
    job.launch_qts(3, 1, svb_out)
    res = job.wait()
    for r, raw in enumerate(reads):
        x = raw.astype(np.int64)
        e = np.where((x & 7) < 4, x & ~7, (x & ~7) + 8).astype(np.int16)
        if svb_out:
            assert res["blobs"][r] == blow5.svb_zd_encode(e), "read %d" % r
        else:
            assert np.array_equal(res["samples"][r], e)
    job.close()
