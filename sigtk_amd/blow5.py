"""Minimal BLOW5 reader/writer (pure Python + numpy + zlib).

Convenience for tests, golden generation and bench tooling; the product's host
reader is the C implementation in ``sigtk_amd/host/blow5.c``.  Both were written
from the on-disk layout (the reference reads it through
slow5lib: slow5lib/src/slow5.c:794-881 header, :2822-2927 record,
slow5lib/src/slow5_press.c:1091-1146 svb-zd signal blob).

Layout recap (little endian):
  magic "BLOW5\\1", version u8x3, record_press u8 (0 none, 1 zlib), num_read_groups u32,
  signal_press u8 (0 none, 1 svb-zd), zero padding to offset 64, u32 header_size,
  header text (@attr lines, '#types' line, '#names' line), then records
  (u64 size + bytes, the bytes being one zlib stream when record_press == 1) until "5WOLB".
  Record: u16 id_len, id, u32 read_group, f64 digitisation, f64 offset, f64 range,
  f64 sampling_rate, u64 len_raw_signal (byte length of the svb-zd blob when signal_press == 1,
  else the sample count), signal, auxiliary fields (ignored here).
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Sequence

import numpy as np

MAGIC = b"BLOW5\x01"
EOF_MARK = b"5WOLB"


@dataclass
class Read:
    read_id: str
    read_group: int
    digitisation: float
    offset: float
    range: float
    sampling_rate: float
    raw: np.ndarray  # int16


@dataclass
class Blow5:
    version: tuple
    record_press: int
    signal_press: int
    num_read_groups: int
    attrs: Dict[str, List[str]] = field(default_factory=dict)
    reads: List[Read] = field(default_factory=list)

    def attr(self, name: str, group: int = 0):
        v = self.attrs.get(name)
        return None if v is None else v[group]


# --------------------------------------------------------------------------- svb-zd

_KEY_SHIFTS = np.array([0, 2, 4, 6], dtype=np.uint8)


def svb_zd_decode(blob: bytes) -> np.ndarray:
    """u32 count, streamvbyte(count x u32 zigzag-deltas, prev=0) -> int16[count]."""
    (count,) = struct.unpack_from("<I", blob, 0)
    if count == 0:
        return np.zeros(0, dtype=np.int16)
    nkeys = (count + 3) // 4
    keys = np.frombuffer(blob, dtype=np.uint8, count=nkeys, offset=4)
    codes = ((keys[:, None] >> _KEY_SHIFTS[None, :]) & 3).reshape(-1)[:count].astype(np.int64)
    lens = codes + 1
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    data = np.frombuffer(blob, dtype=np.uint8, offset=4 + nkeys)
    if int(starts[-1] + lens[-1]) != len(data):
        raise ValueError("svb-zd: data length mismatch")
    padded = np.concatenate((data, np.zeros(4, dtype=np.uint8))).astype(np.uint32)
    vals = padded[starts].copy()
    for b in (1, 2, 3):
        m = lens > b
        vals[m] |= padded[starts[m] + b] << np.uint32(8 * b)
    zz = vals.astype(np.int64)
    delta = (zz >> 1) ^ -(zz & 1)
    return np.cumsum(delta).astype(np.int16)


def svb_zd_encode(raw: np.ndarray) -> bytes:
    raw = np.asarray(raw, dtype=np.int16).astype(np.int64)
    count = raw.size
    if count == 0:
        return struct.pack("<I", 0)
    delta = np.diff(raw, prepend=0)
    zz = ((delta << 1) ^ (delta >> 31)).astype(np.uint32)
    codes = np.zeros(count, dtype=np.uint8)
    codes[zz >= (1 << 8)] = 1
    codes[zz >= (1 << 16)] = 2
    codes[zz >= (1 << 24)] = 3
    nkeys = (count + 3) // 4
    padded = np.zeros(nkeys * 4, dtype=np.uint8)
    padded[:count] = codes
    keys = (padded.reshape(-1, 4) << _KEY_SHIFTS[None, :]).sum(axis=1).astype(np.uint8)
    lens = codes.astype(np.int64) + 1
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    data = np.zeros(int(lens.sum()), dtype=np.uint8)
    for b in range(4):
        m = lens > b
        data[starts[m] + b] = ((zz[m] >> np.uint32(8 * b)) & 0xFF).astype(np.uint8)
    return struct.pack("<I", count) + keys.tobytes() + data.tobytes()


# --------------------------------------------------------------------------- reader


def read_blow5(path: str) -> Blow5:
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:6] != MAGIC:
        raise ValueError("not a BLOW5 file")
    version = tuple(buf[6:9])
    record_press = buf[9]
    (nrg,) = struct.unpack_from("<I", buf, 10)
    signal_press = buf[14] if version >= (0, 2, 0) else 0
    (hsize,) = struct.unpack_from("<I", buf, 64)
    text = buf[68 : 68 + hsize].decode("ascii", errors="replace")
    out = Blow5(version, record_press, signal_press, nrg)
    for line in text.split("\n"):
        if line.startswith("@"):
            parts = line[1:].split("\t")
            out.attrs[parts[0]] = parts[1:]
    if record_press not in (0, 1):
        raise ValueError("record compression %d not supported (zstd?)" % record_press)
    pos = 68 + hsize
    end = len(buf)
    while True:
        if buf[pos : pos + 5] == EOF_MARK and pos + 5 == end:
            break
        if pos + 8 > end:
            raise ValueError("truncated BLOW5 (no EOF marker)")
        (size,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        rec = buf[pos : pos + size]
        pos += size
        if record_press == 1:
            rec = zlib.decompress(rec)
        (idl,) = struct.unpack_from("<H", rec, 0)
        rid = rec[2 : 2 + idl].decode("ascii")
        p = 2 + idl
        rg, dig, off, rng, sr, ln = struct.unpack_from("<IddddQ", rec, p)
        p += 44
        if signal_press == 1:
            raw = svb_zd_decode(rec[p : p + ln])
        else:
            raw = np.frombuffer(rec, dtype="<i2", count=ln, offset=p).copy()
        out.reads.append(Read(rid, rg, dig, off, rng, sr, raw))
    return out


# --------------------------------------------------------------------------- writer

_TYPES = "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
_NAMES = "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"


def write_blow5(path: str, reads: Sequence[Read], attrs: Dict[str, str] | None = None,
                record_press: int = 1, signal_press: int = 1) -> None:
    """Write a single-read-group BLOW5 (version 0.2.0) the reference CLI can open."""
    attrs = dict(attrs or {})
    text = "".join("@%s\t%s\n" % (k, attrs[k]) for k in sorted(attrs)) + _TYPES + _NAMES
    tb = text.encode("ascii")
    head = bytearray(64)
    head[0:6] = MAGIC
    head[6:9] = bytes((0, 2, 0))
    head[9] = record_press
    head[10:14] = struct.pack("<I", 1)
    head[14] = signal_press
    with open(path, "wb") as fh:
        fh.write(bytes(head))
        fh.write(struct.pack("<I", len(tb)))
        fh.write(tb)
        for r in reads:
            raw = np.ascontiguousarray(r.raw, dtype="<i2")
            if signal_press == 1:
                sig = svb_zd_encode(raw)
                ln = len(sig)
            else:
                sig = raw.tobytes()
                ln = raw.size
            rid = r.read_id.encode("ascii")
            rec = struct.pack("<H", len(rid)) + rid + struct.pack(
                "<IddddQ", r.read_group, r.digitisation, r.offset, r.range, r.sampling_rate, ln) + sig
            if record_press == 1:
                rec = zlib.compress(rec)
            fh.write(struct.pack("<Q", len(rec)))
            fh.write(rec)
        fh.write(EOF_MARK)


def read_signal_blobs(path: str):
    """-> list of (Read with an EMPTY raw array, svb-zd blob bytes): the signal exactly as it sits in
    the (inflated) record, for the GPU decoder (sgk_svbzd_decode).  Only for signal_press == 1 files."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:6] != MAGIC:
        raise ValueError("not a BLOW5 file")
    record_press = buf[9]
    signal_press = buf[14]
    if signal_press != 1:
        raise ValueError("file does not use svb-zd signal compression")
    (hsize,) = struct.unpack_from("<I", buf, 64)
    pos = 68 + hsize
    out = []
    while buf[pos : pos + 5] != EOF_MARK or pos + 5 != len(buf):
        (size,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        rec = buf[pos : pos + size]
        pos += size
        if record_press == 1:
            rec = zlib.decompress(rec)
        (idl,) = struct.unpack_from("<H", rec, 0)
        rid = rec[2 : 2 + idl].decode("ascii")
        p = 2 + idl
        rg, dig, off, rng, sr, ln = struct.unpack_from("<IddddQ", rec, p)
        p += 44
        out.append((Read(rid, rg, dig, off, rng, sr, np.zeros(0, dtype=np.int16)), bytes(rec[p : p + ln])))
    return out


def iter_reads(path: str) -> Iterator[Read]:
    yield from read_blow5(path).reads


def digest(path: str) -> str:
    """sha256 over what a reader sees in a file: (read id, read group, digitisation, offset, range, sampling rate,
    samples) of every record, in file order.  Used to compare `qts` outputs written by different writers."""
    import hashlib
    h = hashlib.sha256()
    for r in read_blow5(path).reads:
        h.update(r.read_id.encode())
        h.update(struct.pack("<Idddd", r.read_group, r.digitisation, r.offset, r.range, r.sampling_rate))
        h.update(r.raw.astype("<i2").tobytes())
    return h.hexdigest()


def raw_records(path: str):
    """-> list of the inflated record byte strings of a file (whole records, auxiliary fields included)"""
    with open(path, "rb") as fh:
        buf = fh.read()
    record_press = buf[9]
    (hsize,) = struct.unpack_from("<I", buf, 64)
    pos = 68 + hsize
    out = []
    while buf[pos : pos + 5] != EOF_MARK or pos + 5 != len(buf):
        (size,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        rec = buf[pos : pos + size]
        pos += size
        out.append(zlib.decompress(rec) if record_press == 1 else bytes(rec))
    return out
