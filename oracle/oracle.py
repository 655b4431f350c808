"""ctypes loaders for the test oracle.  TEST INFRASTRUCTURE ONLY.

``Oracle``   wraps oracle/liboracle.so (the repo's own C restatement, built by
             ``make -C oracle oracle``; gcc only).
``RefLib``   wraps oracle/_ref/libsigtk_ref.so (the REAL reference compiled from
             /root/reference by ``make -C oracle ref``) when it has been built.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsigtk_ref.so")
REF_BIN = os.path.join(HERE, "_ref", "sigtk_ref")


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, if /root/reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.exists("/root/reference/src/events.c"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Events(NamedTuple):
    start: np.ndarray   # uint64
    length: np.ndarray  # float32
    mean: np.ndarray    # float32
    stdv: np.ndarray    # float32


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class _PrefixT(C.Structure):
    _fields_ = [("adapt_x", C.c_int64), ("adapt_y", C.c_int64), ("polya_x", C.c_int64),
                ("polya_y", C.c_int64), ("adapt_mean", C.c_float), ("adapt_std", C.c_float),
                ("adapt_median", C.c_float), ("polya_mean", C.c_float), ("polya_std", C.c_float),
                ("polya_median", C.c_float)]


class _JnnParam(C.Structure):
    _fields_ = [("std_scale", C.c_float), ("corrector", C.c_int), ("seg_dist", C.c_int),
                ("window", C.c_int), ("stall_len", C.c_float), ("error", C.c_int),
                ("top", C.c_float), ("bot", C.c_float)]


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_event_raw.restype = C.c_int64
        L.orc_getevents.restype = C.c_int64
        L.orc_peaks.restype = C.c_int64
        L.orc_event_batch_count.restype = C.c_int64
        L.orc_jnn_raw.restype = C.c_int64
        L.orc_jnn_core.restype = C.c_int64
        L.orc_jnn_pa.restype = C.c_int64
        L.orc_jnn_preset.restype = _JnnParam
        for f in ("orc_meanf", "orc_stdvf", "orc_medianf", "orc_meani16", "orc_stdvi16"):
            getattr(L, f).restype = C.c_float
        L.orc_mediani16.restype = C.c_int16

    # -- pa
    def pa(self, raw, dig, off, rng) -> np.ndarray:
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.empty(raw.size, dtype=np.float32)
        self.lib.orc_pa(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_double(dig), C.c_double(off),
                        C.c_double(rng), _p(out, C.c_float))
        return out

    # -- event pieces
    def prefix_sums(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s = np.empty(x.size + 1, dtype=np.float64)
        q = np.empty(x.size + 1, dtype=np.float64)
        self.lib.orc_prefix_sums(_p(x, C.c_float), C.c_int64(x.size), _p(s, C.c_double), _p(q, C.c_double))
        return s, q

    def tstat(self, s, q, w):
        n = s.size - 1
        t = np.empty(n, dtype=np.float32)
        self.lib.orc_tstat(_p(s, C.c_double), _p(q, C.c_double), C.c_int64(n), C.c_int(w), _p(t, C.c_float))
        return t

    def peaks(self, t1, t2, rna):
        n = t1.size
        w1, w2 = (7, 14) if rna else (3, 6)
        thr1, thr2, ph = (2.5, 9.0, 1.0) if rna else (1.4, 9.0, 0.2)
        pk = np.empty(max(n, 1), dtype=np.int64)
        k = self.lib.orc_peaks(_p(t1, C.c_float), _p(t2, C.c_float), C.c_int64(n), C.c_int(w1), C.c_int(w2),
                               C.c_float(thr1), C.c_float(thr2), C.c_float(ph), _p(pk, C.c_int64))
        return pk[:k].copy()

    def getevents(self, pa, rna, faithful=0) -> Events:
        pa = np.ascontiguousarray(pa, dtype=np.float32)
        cap = pa.size // 2 + 2
        st = np.empty(cap, dtype=np.uint64)
        ln = np.empty(cap, dtype=np.float32)
        mn = np.empty(cap, dtype=np.float32)
        sd = np.empty(cap, dtype=np.float32)
        k = self.lib.orc_getevents(_p(pa, C.c_float), C.c_int64(pa.size), C.c_int(rna), C.c_int(faithful),
                                   _p(st, C.c_uint64), _p(ln, C.c_float), _p(mn, C.c_float), _p(sd, C.c_float),
                                   C.c_int64(cap))
        return Events(st[:k].copy(), ln[:k].copy(), mn[:k].copy(), sd[:k].copy())

    def event_raw(self, raw, dig, off, rng, rna, faithful=0) -> Events:
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        cap = raw.size // 2 + 2
        st = np.empty(cap, dtype=np.uint64)
        ln = np.empty(cap, dtype=np.float32)
        mn = np.empty(cap, dtype=np.float32)
        sd = np.empty(cap, dtype=np.float32)
        k = self.lib.orc_event_raw(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_double(dig), C.c_double(off),
                                   C.c_double(rng), C.c_int(rna), C.c_int(faithful), _p(st, C.c_uint64),
                                   _p(ln, C.c_float), _p(mn, C.c_float), _p(sd, C.c_float), C.c_int64(cap))
        return Events(st[:k].copy(), ln[:k].copy(), mn[:k].copy(), sd[:k].copy())

    def event_batch_count(self, samples, offsets, dig, off, rng, rna, faithful=1) -> int:
        samples = np.ascontiguousarray(samples, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        dig = np.ascontiguousarray(dig, dtype=np.float64)
        off = np.ascontiguousarray(off, dtype=np.float64)
        rng = np.ascontiguousarray(rng, dtype=np.float64)
        return int(self.lib.orc_event_batch_count(_p(samples, C.c_int16), _p(offsets, C.c_uint64),
                                                  C.c_uint32(offsets.size - 1), _p(dig, C.c_double),
                                                  _p(off, C.c_double), _p(rng, C.c_double), C.c_int(rna),
                                                  C.c_int(faithful)))

    # -- stat
    def stat(self, raw, dig, off, rng):
        """-> (raw_mean, pa_mean, raw_std, pa_std, raw_median, pa_median)"""
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out5 = np.empty(5, dtype=np.float32)
        med = C.c_int32(0)
        self.lib.orc_stat(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_double(dig), C.c_double(off),
                          C.c_double(rng), _p(out5, C.c_float), C.byref(med))
        return (out5[0], out5[1], out5[2], out5[3], int(med.value), out5[4])

    def statf(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = C.c_int64(x.size)
        return (np.float32(self.lib.orc_meanf(_p(x, C.c_float), n)),
                np.float32(self.lib.orc_stdvf(_p(x, C.c_float), n)),
                np.float32(self.lib.orc_medianf(_p(x, C.c_float), n)))

    # -- jnn
    def jnn_raw(self, raw, rna):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        cap = raw.size // 16 + 16
        x = np.empty(cap, dtype=np.int64)
        y = np.empty(cap, dtype=np.int64)
        k = self.lib.orc_jnn_raw(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_int(rna), _p(x, C.c_int64),
                                 _p(y, C.c_int64), C.c_int64(cap))
        return x[:k].copy(), y[:k].copy()

    def jnn_param(self, **kw) -> "_JnnParam":
        """jnn_param_t (src/jnn.h:18-27) from keywords"""
        return _JnnParam(kw.get("std_scale", 0.75), kw.get("corrector", 50), kw.get("seg_dist", 50),
                         kw.get("window", 150), kw.get("stall_len", 0.25), kw.get("error", 5),
                         kw.get("top", 0.0), kw.get("bot", 0.0))

    def jnn_core(self, sig, p: "_JnnParam"):
        """jnn_core (src/jnn.c:190-278) on an already-clamped float signal, any jnn_param_t"""
        sig = np.ascontiguousarray(sig, dtype=np.float32)
        cap = sig.size // 2 + 16
        x = np.empty(cap, dtype=np.int64)
        y = np.empty(cap, dtype=np.int64)
        k = self.lib.orc_jnn_core(_p(sig, C.c_float), C.c_int64(sig.size), p, _p(x, C.c_int64), _p(y, C.c_int64),
                                  C.c_int64(cap))
        return x[:k].copy(), y[:k].copy()

    def jnn_pa(self, pa, p: "_JnnParam"):
        """jnn_pa (src/jnn.c:295-306): rm_outlierf then jnn_core"""
        pa = np.ascontiguousarray(pa, dtype=np.float32)
        cap = pa.size // 2 + 16
        x = np.empty(cap, dtype=np.int64)
        y = np.empty(cap, dtype=np.int64)
        k = self.lib.orc_jnn_pa(_p(pa, C.c_float), C.c_int64(pa.size), p, _p(x, C.c_int64), _p(y, C.c_int64),
                                C.c_int64(cap))
        return x[:k].copy(), y[:k].copy()

    def find_adaptor(self, raw, pore):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        xy = np.zeros(2, dtype=np.int64)
        self.lib.orc_find_adaptor(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_int(pore), _p(xy, C.c_int64))
        return int(xy[0]), int(xy[1])

    def find_polya(self, pa, top, bot, pore):
        pa = np.ascontiguousarray(pa, dtype=np.float32)
        xy = np.zeros(2, dtype=np.int64)
        self.lib.orc_find_polya(_p(pa, C.c_float), C.c_int64(pa.size), C.c_float(top), C.c_float(bot),
                                C.c_int(pore), _p(xy, C.c_int64))
        return int(xy[0]), int(xy[1])

    def ent(self, raw):
        """-> (raw_ent, delta_ent, byte_ent) as float64, src/ent.c"""
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.zeros(3, dtype=np.float64)
        self.lib.orc_ent(_p(raw, C.c_int16), C.c_int64(raw.size), _p(out, C.c_double))
        return out

    def prefix(self, raw, dig, off, rng, rna, pore):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = _PrefixT()
        self.lib.orc_prefix(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_double(dig), C.c_double(off),
                            C.c_double(rng), C.c_int(rna), C.c_int(pore), C.byref(out))
        return out


class RefLib:
    """The real reference (oracle/_ref/libsigtk_ref.so), if built."""

    def __init__(self, path: str = REF_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_getevents.restype = C.c_int64
        L.ref_event_raw.restype = C.c_int64
        L.ref_event_batch_count.restype = C.c_int64
        L.ref_jnn_raw.restype = C.c_int

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def pa(self, raw, dig, off, rng):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.empty(raw.size, dtype=np.float32)
        self.lib.ref_pa(_p(raw, C.c_int16), C.c_uint64(raw.size), C.c_double(dig), C.c_double(off),
                        C.c_double(rng), _p(out, C.c_float))
        return out

    def event_raw(self, raw, dig, off, rng, rna) -> Events:
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        cap = raw.size // 2 + 2
        st = np.empty(cap, dtype=np.uint64)
        ln = np.empty(cap, dtype=np.float32)
        mn = np.empty(cap, dtype=np.float32)
        sd = np.empty(cap, dtype=np.float32)
        k = self.lib.ref_event_raw(_p(raw, C.c_int16), C.c_uint64(raw.size), C.c_double(dig), C.c_double(off),
                                   C.c_double(rng), C.c_int(rna), _p(st, C.c_uint64), _p(ln, C.c_float),
                                   _p(mn, C.c_float), _p(sd, C.c_float), C.c_int64(cap))
        return Events(st[:k].copy(), ln[:k].copy(), mn[:k].copy(), sd[:k].copy())

    def event_batch_count(self, samples, offsets, dig, off, rng, rna) -> int:
        samples = np.ascontiguousarray(samples, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        dig = np.ascontiguousarray(dig, dtype=np.float64)
        off = np.ascontiguousarray(off, dtype=np.float64)
        rng = np.ascontiguousarray(rng, dtype=np.float64)
        return int(self.lib.ref_event_batch_count(_p(samples, C.c_int16), _p(offsets, C.c_uint64),
                                                  C.c_uint32(offsets.size - 1), _p(dig, C.c_double),
                                                  _p(off, C.c_double), _p(rng, C.c_double), C.c_int(rna)))

    def stat(self, raw, dig, off, rng):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out5 = np.empty(5, dtype=np.float32)
        med = C.c_int32(0)
        self.lib.ref_stat(_p(raw, C.c_int16), C.c_uint64(raw.size), C.c_double(dig), C.c_double(off),
                          C.c_double(rng), _p(out5, C.c_float), C.byref(med))
        return (out5[0], out5[1], out5[2], out5[3], int(med.value), out5[4])

    def statf(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(3, dtype=np.float32)
        self.lib.ref_statf(_p(x, C.c_float), C.c_int(x.size), _p(out, C.c_float))
        return (out[0], out[1], out[2])

    def jnn_raw(self, raw, rna):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        cap = raw.size // 16 + 16
        x = np.empty(cap, dtype=np.int64)
        y = np.empty(cap, dtype=np.int64)
        k = self.lib.ref_jnn_raw(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_int(rna), _p(x, C.c_int64),
                                 _p(y, C.c_int64), C.c_int(cap))
        return x[:k].copy(), y[:k].copy()

    def find_adaptor(self, raw, pore):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        xy = np.zeros(2, dtype=np.int64)
        self.lib.ref_find_adaptor(_p(raw, C.c_int16), C.c_int64(raw.size), C.c_int(pore), _p(xy, C.c_int64))
        return int(xy[0]), int(xy[1])

    def find_polya(self, pa, top, bot, pore):
        pa = np.ascontiguousarray(pa, dtype=np.float32)
        xy = np.zeros(2, dtype=np.int64)
        self.lib.ref_find_polya(_p(pa, C.c_float), C.c_int64(pa.size), C.c_float(top), C.c_float(bot),
                                C.c_int(pore), _p(xy, C.c_int64))
        return int(xy[0]), int(xy[1])


_oracle: Optional[Oracle] = None


def get_oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle
