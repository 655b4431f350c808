"""Device-resident batches on top of the C ABI.  torch is used only as the allocator / stream
provider (plumbing): every compute call goes through libsigtk_gpu.so with raw device pointers
and the current torch HIP stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import api


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


@dataclass
class DeviceReads:
    """A batch of reads resident in HBM (sgk_batch_t view + owning tensors)."""
    samples: torch.Tensor   # int16 [n_samples]
    offsets: torch.Tensor   # int64 (uint64 bits) [n_reads]
    lengths: torch.Tensor   # int32 (uint32 bits) [n_reads]
    dig: torch.Tensor       # float64 [n_reads]
    off: torch.Tensor
    rng: torch.Tensor
    n_reads: int
    max_read_len: int
    n_samples: int
    offsets_host: np.ndarray
    lengths_host: np.ndarray

    def view(self) -> api.Batch:
        return api.Batch(_ptr(self.samples), _ptr(self.offsets), _ptr(self.lengths), _ptr(self.dig),
                         _ptr(self.off), _ptr(self.rng), self.n_reads, self.max_read_len, self.n_samples)

    @property
    def total_samples(self) -> int:
        return int(self.lengths_host.astype(np.int64).sum())


def alloc_reads(lengths: np.ndarray, device: torch.device, align: int = 64) -> DeviceReads:
    """Lay out reads of the given lengths with every read starting on an `align`-sample boundary."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.size
    padded = (lengths + align - 1) // align * align
    # 64 samples of head room and tail room: the event fast path reads a little outside each read
    offsets = np.full(n, 256, dtype=np.int64)
    if n > 1:
        offsets[1:] += np.cumsum(padded[:-1])
    n_samples = ((int(padded.sum()) if n else 0) + 320 + 7) // 8 * 8   # the ABI wants a multiple of 8
    return DeviceReads(
        samples=torch.zeros(n_samples, dtype=torch.int16, device=device),
        offsets=torch.from_numpy(offsets).to(device),
        lengths=torch.from_numpy(lengths.astype(np.int32)).to(device),
        dig=torch.zeros(max(n, 1), dtype=torch.float64, device=device),
        off=torch.zeros(max(n, 1), dtype=torch.float64, device=device),
        rng=torch.zeros(max(n, 1), dtype=torch.float64, device=device),
        n_reads=n, max_read_len=int(lengths.max()) if n else 0, n_samples=n_samples,
        offsets_host=offsets.astype(np.uint64), lengths_host=lengths.astype(np.uint32))


def synth_reads(n_reads: int, read_len: int, seed: int, kind: int, device: torch.device,
                first_read: int = 0, lengths=None) -> DeviceReads:
    """Synthetic reads generated on the device (same generator as api.synth_reads_host); `lengths` (one per read)
    replaces the common `read_len`."""
    L = api.load_library()
    lens = np.full(n_reads, read_len, dtype=np.int64) if lengths is None else np.asarray(lengths, dtype=np.int64)
    b = alloc_reads(lens, device)
    api.check(L.sgk_synth_reads(_ptr(b.samples), _ptr(b.offsets), _ptr(b.lengths), _ptr(b.dig), _ptr(b.off),
                                _ptr(b.rng), b.n_reads, b.max_read_len, first_read, seed, kind, _stream_ptr()),
              "sgk_synth_reads")
    return b


def upload_reads(reads, dig, off, rng, device: torch.device) -> DeviceReads:
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    b = alloc_reads(lens, device)
    host = np.zeros(b.n_samples, dtype=np.int16)
    for r, raw in enumerate(reads):
        o = int(b.offsets_host[r])
        host[o:o + len(raw)] = raw
    b.samples.copy_(torch.from_numpy(host))
    b.dig[:b.n_reads] = torch.from_numpy(np.asarray(dig, dtype=np.float64))
    b.off[:b.n_reads] = torch.from_numpy(np.asarray(off, dtype=np.float64))
    b.rng[:b.n_reads] = torch.from_numpy(np.asarray(rng, dtype=np.float64))
    return b


class EventArena:
    """Output arena + workspace of sgk_event for one batch shape (allocated once, reused)."""

    def __init__(self, b: DeviceReads):
        L = api.load_library()
        dev = b.samples.device
        slots = np.zeros(b.n_reads + 1, dtype=np.int64)
        np.cumsum(api.event_slots_for(b.lengths_host), out=slots[1:])
        self.slots_host = slots
        self.n_slots = int(slots[-1])
        self.slots = torch.from_numpy(slots).to(dev)
        # sgk_event_rec_t[n_slots]: (start u32, length u32, mean f32, stdv f32) per slot
        self.events = torch.empty((max(self.n_slots, 1), 4), dtype=torch.int32, device=dev)
        assert self.events.data_ptr() % 16 == 0
        self.n_events = torch.zeros(max(b.n_reads, 1), dtype=torch.int32, device=dev)
        # (sized for the options of the call that follows: api.EVENT_OPTIONS as they are now)
        self.opt = api.EventOptions.from_buffer_copy(bytes(api.EVENT_OPTIONS))
        self.ws_bytes = int(L.sgk_event_workspace_bytes_opt(b.n_reads, b.n_samples, b.max_read_len, C.byref(self.opt)))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        assert self.ws.data_ptr() % 64 == 0

    def status(self) -> api.EventStatus:
        L = api.load_library()
        st = api.EventStatus()
        rc = L.sgk_event_status(_ptr(self.ws), C.byref(st), _stream_ptr())
        if rc != api.SGK_OK and rc != api.SGK_ERR_CAPACITY:
            api.check(rc, "sgk_event_status")
        return st

    def read_events(self, r: int) -> api.Events:
        k = int(self.n_events[r].item())
        s = int(self.slots_host[r])
        rec = self.events[s:s + k].cpu().numpy()
        return api.Events(rec[:, 0].copy().view(np.uint32), rec[:, 1].copy().view(np.uint32),
                          rec[:, 2].copy().view(np.float32), rec[:, 3].copy().view(np.float32))

    # column views of the record array (device tensors; mean / stdv as float32)
    @property
    def start(self) -> torch.Tensor:
        return self.events[:, 0]

    @property
    def length(self) -> torch.Tensor:
        return self.events[:, 1]

    @property
    def mean(self) -> torch.Tensor:
        return self.events.view(torch.float32)[:, 2]

    @property
    def stdv(self) -> torch.Tensor:
        return self.events.view(torch.float32)[:, 3]


def event(b: DeviceReads, arena: EventArena, rna: int) -> None:
    """Enqueue one pass of the event path over the batch on the current stream (async)."""
    L = api.load_library()
    view = b.view()
    api.check(L.sgk_event_opt(C.byref(view), int(rna), _ptr(arena.slots), _ptr(arena.events), _ptr(arena.n_events),
                              _ptr(arena.ws), arena.ws_bytes, _stream_ptr(), C.byref(arena.opt)), "sgk_event_opt")


# ---------------------------------------------------------------------- stat / jnn / prefix / pa (device API)

def _workspace(b: "DeviceReads", size_fn_name: str) -> torch.Tensor:
    """the workspace sgk_<tool>_workspace_bytes asks for (counters + the longest-first dispatch order of the
    wave-per-read kernels), allocated once per batch"""
    cache = b.__dict__.setdefault("_ws_cache", {})
    if size_fn_name not in cache:
        L = api.load_library()
        n = int(getattr(L, size_fn_name)(b.n_reads, b.n_samples, b.max_read_len))
        cache[size_fn_name] = torch.zeros(max(n, 64), dtype=torch.uint8, device=b.samples.device)
    return cache[size_fn_name]


def long_status(b: "DeviceReads", tool: str = "stat", ws: Optional[torch.Tensor] = None) -> "api.LongStatus":
    """sgk_stat_long_status of the workspace the last stat / prefix call on this batch used (jnn: pass arena.ws);
    synchronises"""
    if ws is None:
        ws = _workspace(b, "sgk_%s_workspace_bytes" % tool)
    torch.cuda.synchronize()
    st = api.LongStatus()
    api.check(api.load_library().sgk_stat_long_status(_ptr(ws), ws.numel(), b.n_reads, C.byref(st)), "sgk_stat_long_status")
    return st


def stat(b: DeviceReads) -> torch.Tensor:
    """sgk_stat -> uint8 tensor holding n_reads sgk_stat_rec_t (view it with api.STAT_DTYPE)."""
    L = api.load_library()
    out = torch.zeros(max(b.n_reads, 1) * api.STAT_DTYPE.itemsize, dtype=torch.uint8, device=b.samples.device)
    view = b.view()
    ws = _workspace(b, "sgk_stat_workspace_bytes")
    api.check(L.sgk_stat_opt(C.byref(view), _ptr(out), _ptr(ws), ws.numel(), _stream_ptr(), C.byref(api.STAT_OPTIONS)),
              "sgk_stat_opt")
    return out


def stat_pa(b: DeviceReads, pa_out: Optional[torch.Tensor] = None):
    """sgk_stat_pa (BASELINE config 4): stat records + pA of every sample in one launch sequence; the pA values
    are written by the median pass.  -> (stat record bytes, pa tensor laid out like b.samples)"""
    L = api.load_library()
    out = torch.zeros(max(b.n_reads, 1) * api.STAT_DTYPE.itemsize, dtype=torch.uint8, device=b.samples.device)
    if pa_out is None:
        pa_out = torch.empty(b.n_samples, dtype=torch.float32, device=b.samples.device)
    view = b.view()
    ws = _workspace(b, "sgk_stat_workspace_bytes")
    api.check(L.sgk_stat_pa_opt(C.byref(view), _ptr(out), _ptr(pa_out), _ptr(ws), ws.numel(), _stream_ptr(),
                                C.byref(api.STAT_OPTIONS)), "sgk_stat_pa_opt")
    return out, pa_out


def pipeline(b: DeviceReads, arena: "EventArena", rna: int, pa_out: Optional[torch.Tensor] = None,
             stat_out: Optional[torch.Tensor] = None):
    """sgk_pipeline (BASELINE config 5): pa -> event -> stat over one resident batch; the event builder writes the pA.
    -> (stat record bytes, pa tensor laid out like b.samples); events land in `arena`"""
    L = api.load_library()
    if stat_out is None:
        stat_out = torch.zeros(max(b.n_reads, 1) * api.STAT_DTYPE.itemsize, dtype=torch.uint8, device=b.samples.device)
    if pa_out is None:
        pa_out = torch.empty(b.n_samples, dtype=torch.float32, device=b.samples.device)
    view = b.view()
    ws = _workspace(b, "sgk_stat_workspace_bytes")
    api.check(L.sgk_pipeline(C.byref(view), int(rna), _ptr(arena.slots), _ptr(arena.events), _ptr(arena.n_events), _ptr(pa_out),
                             _ptr(stat_out), _ptr(arena.ws), arena.ws_bytes, _ptr(ws), ws.numel(), _stream_ptr(),
                             C.byref(arena.opt), C.byref(api.STAT_OPTIONS)), "sgk_pipeline")
    return stat_out, pa_out


def prefix(b: DeviceReads, rna: int, pore: int) -> torch.Tensor:
    L = api.load_library()
    out = torch.zeros(max(b.n_reads, 1) * api.PREFIX_DTYPE.itemsize, dtype=torch.uint8, device=b.samples.device)
    view = b.view()
    ws = _workspace(b, "sgk_prefix_workspace_bytes")
    api.check(L.sgk_prefix_opt(C.byref(view), int(rna), int(pore), _ptr(out), _ptr(ws), ws.numel(), _stream_ptr(),
                               C.byref(api.STAT_OPTIONS)), "sgk_prefix_opt")
    return out


class SegArena:
    def __init__(self, b: DeviceReads):
        dev = b.samples.device
        slots = np.zeros(b.n_reads + 1, dtype=np.int64)
        np.cumsum(b.lengths_host.astype(np.int64) // 32 + 2, out=slots[1:])
        self.slots_host = slots
        self.slots = torch.from_numpy(slots).to(dev)
        self.x = torch.empty(max(int(slots[-1]), 1), dtype=torch.int32, device=dev)
        self.y = torch.empty(max(int(slots[-1]), 1), dtype=torch.int32, device=dev)
        self.n_segs = torch.zeros(max(b.n_reads, 1), dtype=torch.int32, device=dev)
        n = int(api.load_library().sgk_jnn_workspace_bytes(b.n_reads, b.n_samples, b.max_read_len))
        self.ws = torch.zeros(max(n, 64), dtype=torch.uint8, device=dev)


def jnn(b: DeviceReads, arena: SegArena, rna: int) -> None:
    L = api.load_library()
    view = b.view()
    api.check(L.sgk_jnn_opt(C.byref(view), int(rna), _ptr(arena.slots), _ptr(arena.x), _ptr(arena.y),
                            _ptr(arena.n_segs), _ptr(arena.ws), arena.ws.numel(), _stream_ptr(), C.byref(api.STAT_OPTIONS)),
              "sgk_jnn_opt")


def pa(b: DeviceReads, out: torch.Tensor) -> None:
    L = api.load_library()
    view = b.view()
    api.check(L.sgk_pa(C.byref(view), _ptr(out), _stream_ptr()), "sgk_pa")


# ---------------------------------------------------------------------- svb-zd decode (device API)

def svbzd_decode(blobs, counts, device: torch.device):
    """Decode svb-zd signal blobs (bytes objects) on the device.
    -> (DeviceReads with the decoded samples and zeroed scaling, status tensor [n_reads] int32)"""
    L = api.load_library()
    n = len(blobs)
    blens = np.array([len(b) for b in blobs], dtype=np.uint32)
    boffs = np.zeros(n, dtype=np.int64)
    if n > 1:
        boffs[1:] = np.cumsum((blens[:-1].astype(np.int64) + 15) // 16 * 16)  # 16-byte aligned blob starts
    total = int(boffs[-1] + blens[-1]) if n else 0
    host = np.zeros(max(total + 16, 16), dtype=np.uint8)
    for i, b in enumerate(blobs):
        host[int(boffs[i]):int(boffs[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    d_blobs = torch.from_numpy(host).to(device)
    d_boffs = torch.from_numpy(boffs).to(device)
    d_blens = torch.from_numpy(blens.astype(np.int32)).to(device)
    reads = alloc_reads(np.asarray(counts, dtype=np.int64), device)
    status = torch.full((max(n, 1),), -1, dtype=torch.int32, device=device)
    api.check(L.sgk_svbzd_decode(_ptr(d_blobs), _ptr(d_boffs), _ptr(d_blens), n, _ptr(reads.samples),
                                 _ptr(reads.offsets), _ptr(reads.lengths), _ptr(status), _stream_ptr()),
              "sgk_svbzd_decode")
    return reads, status


# ---------------------------------------------------------------------- qts + svb-zd encode (device API)

def qts(b: DeviceReads, bits: int, method: int) -> None:
    """sgk_qts: quantise the samples of every read in place (method 0 floor, 1 round, 2 fill-ones)."""
    L = api.load_library()
    api.check(L.sgk_qts(_ptr(b.samples), _ptr(b.offsets), _ptr(b.lengths), b.n_reads, b.max_read_len, int(bits),
                        int(method), _stream_ptr()), "sgk_qts")


def svbzd_encode(b: DeviceReads):
    """sgk_svbzd_size + sgk_svbzd_encode -> list of blobs (bytes), one per read"""
    L = api.load_library()
    dev = b.samples.device
    n = b.n_reads
    blens = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    api.check(L.sgk_svbzd_size(_ptr(b.samples), _ptr(b.offsets), _ptr(b.lengths), n, _ptr(blens), _stream_ptr()),
              "sgk_svbzd_size")
    torch.cuda.synchronize()
    lens = blens[:n].cpu().numpy().astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum((lens + 7) // 8 * 8, out=offs[1:])
    blobs = torch.zeros(max(int(offs[-1]), 8), dtype=torch.uint8, device=dev)
    d_offs = torch.from_numpy(offs[:n].copy() if n else np.zeros(1, dtype=np.int64)).to(dev)
    api.check(L.sgk_svbzd_encode(_ptr(b.samples), _ptr(b.offsets), _ptr(b.lengths), n, _ptr(blobs), _ptr(d_offs),
                                 _ptr(blens), _stream_ptr()), "sgk_svbzd_encode")
    torch.cuda.synchronize()
    host = blobs.cpu().numpy()
    return [host[int(offs[r]):int(offs[r]) + int(lens[r])].tobytes() for r in range(n)]


def inflate(streams, caps=None, device: Optional[torch.device] = None):
    """sgk_inflate over a list of zlib streams (bytes) -> (list of inflated bytes as kept: the first caps[r] of each,
    out_lengths, status) -- the device-side replacement of slow5lib's per-record uncompress()"""
    L = api.load_library()
    dev = device or torch.device("cuda", 0)
    n = len(streams)
    in_off = np.zeros(n, dtype=np.uint64)
    in_len = np.asarray([len(s) for s in streams], dtype=np.uint32)
    pos = 0
    for r, s in enumerate(streams):
        in_off[r] = pos
        pos += len(s) + (r % 3)          # (odd offsets on purpose: the kernel aligns its dword reads itself)
    blob = np.zeros((pos + 7) // 4 * 4 + 4, dtype=np.uint8)
    for r, s in enumerate(streams):
        blob[int(in_off[r]):int(in_off[r]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
    caps_a = np.asarray(caps if caps is not None else [1 << 20] * n, dtype=np.uint32)
    out_off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        out_off[1:] = np.cumsum((caps_a[:-1].astype(np.uint64) + 15) // 16 * 16)
    total = int(out_off[-1] + (int(caps_a[-1]) + 15) // 16 * 16) if n else 16
    d_in = torch.from_numpy(blob).to(dev)
    d_ioff = torch.from_numpy(in_off.view(np.int64)).to(dev)
    d_ilen = torch.from_numpy(in_len.view(np.int32)).to(dev)
    d_out = torch.zeros(max(total, 16), dtype=torch.uint8, device=dev)
    d_ooff = torch.from_numpy(out_off.view(np.int64)).to(dev)
    d_caps = torch.from_numpy(caps_a.view(np.int32)).to(dev)
    d_olen = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    d_st = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    api.check(L.sgk_inflate(_ptr(d_in), _ptr(d_ioff), _ptr(d_ilen), n, _ptr(d_out), _ptr(d_ooff), _ptr(d_caps), _ptr(d_olen),
                            _ptr(d_st), _stream_ptr()), "sgk_inflate")
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    olen = d_olen.cpu().numpy().view(np.uint32)[:n]
    st = d_st.cpu().numpy()[:n]
    kept = [out[int(out_off[r]):int(out_off[r]) + min(int(olen[r]), int(caps_a[r]))].tobytes() for r in range(n)]
    return kept, olen, st
