"""GPU: sgk_inflate (csrc/inflate_kernels.hip) -- zlib streams inflated one wavefront each -- against Python's zlib
(the library slow5lib itself calls, slow5lib/src/slow5_press.c:77-98): every block type, every level and window size,
sizes around the kernel's chunk / flush / window boundaries, and the malformed streams zlib rejects."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _payloads():
    rs = np.random.RandomState(5)
    text = (b"the quick brown fox jumps over the lazy dog, 0123456789; " * 6000)
    from sigtk_amd import api, blow5
    reads, *_ = api.synth_reads_host(2, 100000, 3, 0)
    svb = blow5.svb_zd_encode(reads[0])
    out = {"empty": b"", "one": b"x", "five": b"hello", "zeros": bytes(70000), "text": text,
           "random": rs.bytes(200000), "svb": svb, "svb2": blow5.svb_zd_encode(reads[1]) + rs.bytes(333),
           "runs": b"".join(bytes([i % 251]) * (1 + (i * 7) % 300) for i in range(3000)),
           "walk": np.cumsum(rs.randint(-2, 3, size=150000)).astype(np.int8).tobytes()}
    for n in (255, 256, 257, 1023, 1024, 1025, 32767, 32768, 32769, 65535, 65536, 65537):
        out["text%d" % n] = text[:n]
        out["rand%d" % n] = rs.bytes(n)
    return out


def test_every_block_type_level_and_window(gpu):
    from sigtk_amd import device
    pay = _payloads()
    streams, want = [], []
    for name, data in pay.items():
        for level in (0, 1, 6, 9):
            streams.append(zlib.compress(data, level)); want.append(data)
        for wbits in (9, 12, 15):
            c = zlib.compressobj(6, zlib.DEFLATED, wbits)
            streams.append(c.compress(data) + c.flush()); want.append(data)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)      # fixed Huffman blocks
        streams.append(c.compress(data) + c.flush()); want.append(data)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_HUFFMAN_ONLY)
        streams.append(c.compress(data) + c.flush()); want.append(data)
        c = zlib.compressobj(6)                                          # several blocks, sync flushes (empty stored blocks)
        parts = [c.compress(data[i:i + 7001]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 7001)]
        streams.append(b"".join(parts) + c.flush()); want.append(data)
    got, olen, st = device.inflate(streams, caps=[len(w) + 5 for w in want])
    for r, w in enumerate(want):
        assert st[r] == 0, (r, st[r])
        assert olen[r] == len(w), (r, olen[r], len(w))
        assert got[r] == w, r


def test_room(gpu):
    """out_caps[r] must hold the whole stream (far matches read the stream's own earlier bytes from the output): exactly
    enough is enough, one byte less is status 8"""
    from sigtk_amd import device
    pay = _payloads()
    data = pay["svb"] + pay["text"][:50000]
    s = zlib.compress(data)
    caps = [0, 1, 1024, len(pay["svb"]), len(data) - 1, len(data), len(data) + 100]
    got, olen, st = device.inflate([s] * len(caps), caps=caps)
    for r, c in enumerate(caps):
        if c >= len(data):
            assert st[r] == 0 and olen[r] == len(data) and got[r] == data
        else:
            assert st[r] == 8, (c, st[r])


def test_streams_zlib_rejects(gpu):
    from sigtk_amd import device
    pay = _payloads()
    good = zlib.compress(pay["svb"])
    bad = {}
    bad["truncated"] = good[:len(good) // 2]
    bad["no adler"] = good[:-4]
    bad["adler"] = good[:-1] + bytes([good[-1] ^ 1])
    bad["header"] = bytes([0x79]) + good[1:]
    bad["method"] = bytes([0x77, 0x9c]) + good[2:]          # CM != 8 (FCHECK still fine? 0x779c % 31 -> adjusted below)
    bad["dict"] = bytes([0x78, 0xbb]) + good[2:]            # FDICT set, 0x78bb % 31 == 0
    bad["short"] = good[:3]
    stored = zlib.compress(pay["random"][:1000], 0)
    bad["stored len"] = stored[:3] + bytes([stored[3] ^ 0xff]) + stored[4:]
    bad["block type 3"] = bytes([0x78, 0x9c, 0x07]) + good[3:]
    rs = np.random.RandomState(7)
    for k in range(24):                                     # a flipped bit somewhere in the middle
        i = int(rs.randint(2, len(good) - 4))
        bad["flip%d" % k] = good[:i] + bytes([good[i] ^ (1 << int(rs.randint(8)))]) + good[i + 1:]
    names = list(bad)
    got, olen, st = device.inflate([bad[k] for k in names], caps=[len(pay["svb"]) + 64] * len(names))
    for r, k in enumerate(names):
        try:
            zlib.decompress(bad[k])
            ok = True
        except zlib.error:
            ok = False
        assert not ok, k           # (zlib rejects every one of these)
        assert st[r] != 0, (k, st[r])
    # ... and the good one among them is still fine
    got, olen, st = device.inflate([good, bad["adler"], good], caps=[1 << 20] * 3)
    assert list(st) == [0, 7, 0] and got[0] == pay["svb"] and got[2] == pay["svb"]


def test_many_streams_at_once(gpu):
    """more streams than the GPU holds at once, of very different sizes"""
    from sigtk_amd import device
    rs = np.random.RandomState(11)
    base = _payloads()["walk"]
    want = [base[int(a):int(a) + int(n)] for a, n in zip(rs.randint(0, 50000, 3000), rs.randint(0, 90000, 3000))]
    got, olen, st = device.inflate([zlib.compress(w, int(rs.randint(1, 10))) for w in want], caps=[len(w) for w in want])
    assert (st == 0).all()
    assert all(g == w for g, w in zip(got, want))
