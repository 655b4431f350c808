"""ctypes binding of libsigtk_gpu.so (include/sigtk_gpu.h) and a numpy-level mirror of the
reference's per-read operators (src/sigtk.h:124-134, src/jnn.h:104-109).

There is NO CPU fallback here: if the HIP library is missing or no GPU is usable the calls
raise ``SigtkGpuError``.  (The test oracle lives under oracle/ and is never imported from
this package.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, NamedTuple, Optional, Sequence

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libsigtk_gpu.so")
# development only: A/B builds of the kernels (tools/build_variant.sh) are selected with SIGTK_AMD_LIB
LIB_PATH = os.environ.get("SIGTK_AMD_LIB", LIB_PATH)

SGK_OK = 0
SGK_ERR_CAPACITY = -5


class SigtkGpuError(RuntimeError):
    pass


class Batch(C.Structure):
    """sgk_batch_t (device pointers)."""
    _fields_ = [("samples", C.c_void_p), ("offsets", C.c_void_p), ("lengths", C.c_void_p),
                ("digitisation", C.c_void_p), ("offset", C.c_void_p), ("range", C.c_void_p),
                ("n_reads", C.c_uint32), ("max_read_len", C.c_uint32), ("n_samples", C.c_uint64)]


class HostBatch(C.Structure):
    """sgk_host_batch_t (host pointers, CSR offsets)."""
    _fields_ = [("samples", C.c_void_p), ("offsets", C.c_void_p), ("digitisation", C.c_void_p),
                ("offset", C.c_void_p), ("range", C.c_void_p), ("n_reads", C.c_uint32)]


class EventStatus(C.Structure):
    _fields_ = [("n_fallback_reads", C.c_uint32), ("n_rerun_passes", C.c_uint32),
                ("n_capacity_overflow", C.c_uint32), ("n_long_replays", C.c_uint32),
                ("n_events_total", C.c_uint64), ("n_split_reads", C.c_uint32), ("n_segments", C.c_uint32),
                ("n_seam_reruns", C.c_uint32), ("reserved", C.c_uint32), ("n_replay_indices", C.c_uint64)]


class EventPlan(C.Structure):   # sgk_event_plan_t
    _fields_ = [("segment_len", C.c_uint32), ("long_min", C.c_uint32), ("max_segments", C.c_uint32),
                ("max_long_reads", C.c_uint32), ("short_max", C.c_uint32), ("lanes_per_short_read", C.c_uint32),
                ("warmup_override", C.c_uint32), ("tail_split_from", C.c_uint32), ("tail_segment_len", C.c_uint32),
                ("reserved", C.c_uint32 * 3)]


class EventOptions(C.Structure):   # sgk_event_options_t (all zero = the defaults)
    _fields_ = [("segment_len", C.c_uint32), ("long_min", C.c_uint32), ("warmup", C.c_int32),
                ("lanes_per_short_read", C.c_int32), ("short_max", C.c_uint32), ("tail_split", C.c_int32),
                ("reserved", C.c_uint32 * 2)]


class StatOptions(C.Structure):   # sgk_stat_options_t
    _fields_ = [("kernels", C.c_int32), ("long_min", C.c_int32), ("debug_fault", C.c_uint32), ("reserved", C.c_uint32)]


class StatPlan(C.Structure):      # sgk_stat_plan_t
    _fields_ = [("kernels", C.c_uint32), ("long_min", C.c_uint32), ("long_max_reads", C.c_uint32),
                ("reserved", C.c_uint32), ("workspace_bytes", C.c_uint64)]


class StatLaneRule(C.Structure):  # sgk_stat_lane_rule_t
    _fields_ = [("tool", C.c_uint32), ("min_reads", C.c_uint32), ("slope_x1024", C.c_uint32), ("intercept", C.c_int32),
                ("cap", C.c_uint32), ("reserved", C.c_uint32)]


class LongStatus(C.Structure):    # sgk_long_status_t
    _fields_ = [("n_long_reads", C.c_uint32), ("n_tiles", C.c_uint32), ("n_true_tiles", C.c_uint32),
                ("n_timeouts", C.c_uint32)]


def _env_int(name, default=0):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


#: The options every wrapper of this module (and sigtk_amd.device) passes with its calls.  The library itself keeps no
#: configuration and reads no environment variable; the development / A-B switches live here, in the bindings:
#: SGK_EVENT_SEG, SGK_EVENT_LONG_MIN, SGK_EVENT_LEAD, SGK_EVENT_MULTI, SGK_EVENT_MULTI_MAX, SGK_EVENT_TAIL (0 = off),
#: SGK_LANE_PER_READ (1 / 0 = one read per lane / per wavefront), SGK_STAT_LONG_MIN (-1 = no long-read path).  Tests
#: set them with event_configure() & co.
EVENT_OPTIONS = EventOptions(_env_int("SGK_EVENT_SEG"), _env_int("SGK_EVENT_LONG_MIN"), _env_int("SGK_EVENT_LEAD"),
                             _env_int("SGK_EVENT_MULTI"), _env_int("SGK_EVENT_MULTI_MAX"),
                             {"0": -1, "1": 0}.get(os.environ.get("SGK_EVENT_TAIL", "1"), _env_int("SGK_EVENT_TAIL")))
STAT_OPTIONS = StatOptions({"1": 1, "0": 2}.get(os.environ.get("SGK_LANE_PER_READ", ""), 0), _env_int("SGK_STAT_LONG_MIN"))


def stat_lane_rules():
    """the table behind sgk_stat_options_t::kernels = 0 (sgk_stat_lane_rules) -> list of (tool, min_reads, max_len(n_reads))"""
    L = load_library()
    out = (StatLaneRule * 32)()
    n = L.sgk_stat_lane_rules(out, 32)
    rows = []
    for k in range(n):
        q = out[k]
        s_, i_, c_ = int(q.slope_x1024), int(q.intercept), int(q.cap)
        rows.append((int(q.tool), int(q.min_reads), (lambda nr, s_=s_, i_=i_, c_=c_: max(0, min(c_, s_ * nr // 1024 + i_)))))
    return rows


def event_configure(segment_len: int = 0, long_min: int = 0, warmup: int = 0) -> None:
    """segment geometry / warm-up of the calls that follow (0 = the library's defaults)"""
    EVENT_OPTIONS.segment_len, EVENT_OPTIONS.long_min, EVENT_OPTIONS.warmup = int(segment_len), int(long_min), int(warmup)


def event_configure_short(lanes_per_read: int = 0, short_max: int = 0) -> None:
    EVENT_OPTIONS.lanes_per_short_read, EVENT_OPTIONS.short_max = int(lanes_per_read), int(short_max)


def event_configure_tail(on: bool = True) -> None:
    EVENT_OPTIONS.tail_split = 0 if on else -1


def stat_configure(kernels: int = 0, long_min: int = 0) -> None:
    """kernels 0: chosen per batch, 1: one read per lane (round-1 kernels), 2: one read per wavefront; long_min: reads of
    at least that many samples take the 16-wavefront sums (0 = 262 144, -1 = never)"""
    STAT_OPTIONS.kernels, STAT_OPTIONS.long_min = int(kernels), int(long_min)


def stat_plan(tool: str, n_reads: int, n_samples: int, max_read_len: int, opt: "StatOptions" = None) -> StatPlan:
    """sgk_stat_plan for tool 'stat' / 'jnn' / 'prefix' / 'stat_pa' (host arithmetic only)"""
    p = StatPlan()
    check(load_library().sgk_stat_plan({"stat": 0, "jnn": 1, "prefix": 2, "stat_pa": 3}[tool], int(n_reads), int(n_samples),
                                       int(max_read_len), C.byref(opt if opt is not None else STAT_OPTIONS), C.byref(p)),
          "sgk_stat_plan")
    return p


def event_plan(n_reads: int, n_samples: int, max_read_len: int, rna: int, opt: "EventOptions" = None) -> EventPlan:
    p = EventPlan()
    check(load_library().sgk_event_plan_opt(int(n_reads), int(n_samples), int(max_read_len), int(rna),
                                        C.byref(opt if opt is not None else EVENT_OPTIONS), C.byref(p)), "sgk_event_plan_opt")
    return p


class EventsHost(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("ev_offsets", C.POINTER(C.c_uint64)),
                ("start", C.POINTER(C.c_uint32)), ("length", C.POINTER(C.c_uint32)),
                ("mean", C.POINTER(C.c_float)), ("stdv", C.POINTER(C.c_float)), ("status", EventStatus)]


class SegsHost(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("seg_offsets", C.POINTER(C.c_uint64)),
                ("x", C.POINTER(C.c_int32)), ("y", C.POINTER(C.c_int32))]


class JobInput(C.Structure):
    """sgk_job_input_t"""
    _fields_ = [("samples", C.c_void_p), ("blobs", C.c_void_p), ("offsets", C.POINTER(C.c_uint64)),
                ("blob_offsets", C.POINTER(C.c_uint64)), ("digitisation", C.POINTER(C.c_double)),
                ("offset", C.POINTER(C.c_double)), ("range", C.POINTER(C.c_double)), ("n_samples", C.c_uint64)]


class JobOutput(C.Structure):
    """sgk_job_output_t"""
    _fields_ = [("n_reads", C.c_uint32), ("offsets", C.POINTER(C.c_uint64)), ("lengths", C.POINTER(C.c_uint32)),
                ("decode_status", C.POINTER(C.c_uint32)), ("pa", C.POINTER(C.c_float)),
                ("slots", C.POINTER(C.c_uint64)), ("counts", C.POINTER(C.c_uint32)),
                ("ev_start", C.POINTER(C.c_uint32)), ("ev_length", C.POINTER(C.c_uint32)),
                ("ev_mean", C.POINTER(C.c_float)), ("ev_stdv", C.POINTER(C.c_float)),
                ("seg_x", C.POINTER(C.c_int32)), ("seg_y", C.POINTER(C.c_int32)),
                ("stat", C.c_void_p), ("prefix", C.c_void_p), ("event_status", EventStatus),
                ("qts_blobs", C.c_void_p), ("qts_blob_offsets", C.POINTER(C.c_uint64)),
                ("qts_blob_lengths", C.POINTER(C.c_uint32)), ("qts_samples", C.c_void_p),
                ("ent", C.c_void_p), ("ent_over_raw", C.c_void_p), ("ent_over_delta", C.c_void_p)]


TOOL_PA, TOOL_EVENT, TOOL_STAT, TOOL_JNN, TOOL_PREFIX, TOOL_ENT, TOOL_QTS = range(7)
ENT_HIST_BYTES = 4 * (4 + 8192 + 4096 + 512)   # sizeof(sgk_ent_hist_t)
SIGNAL_INT16, SIGNAL_SVBZD = 0, 1
JOB_EVENTS_COMPACT = 1
JOB_EVENTS_LENGTHS = 2   # only the lengths come back; the starts are their running sums (events are contiguous from 0)

STAT_DTYPE = np.dtype([("raw_mean", "<f4"), ("pa_mean", "<f4"), ("raw_std", "<f4"), ("pa_std", "<f4"),
                       ("raw_median", "<i4"), ("pa_median", "<f4"), ("n", "<u4"), ("reserved", "<u4")])
PREFIX_DTYPE = np.dtype([("adapt_x", "<i4"), ("adapt_y", "<i4"), ("polya_x", "<i4"), ("polya_y", "<i4"),
                         ("adapt_mean", "<f4"), ("adapt_std", "<f4"), ("adapt_median", "<f4"),
                         ("polya_mean", "<f4"), ("polya_std", "<f4"), ("polya_median", "<f4"),
                         ("n", "<u4"), ("reserved", "<u4")])

#: every symbol include/sigtk_gpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "sgk_strerror", "sgk_version", "sgk_last_hip_error", "sgk_device_count", "sgk_set_device",
    "sgk_pa", "sgk_event_workspace_bytes", "sgk_event", "sgk_event_pa", "sgk_event_status", "sgk_event_plan", "sgk_event_plan_opt", "sgk_job_long_declined", "sgk_inflate", "sgk_job_begin_zrec", "sgk_pipeline", "sgk_stat_lane_rules",
    "sgk_event_workspace_bytes_opt", "sgk_event_opt", "sgk_event_pa_opt", "sgk_event_host_opt",
    "sgk_stat_workspace_bytes", "sgk_stat", "sgk_stat_pa", "sgk_jnn_workspace_bytes", "sgk_jnn",
    "sgk_prefix_workspace_bytes", "sgk_prefix", "sgk_stat_opt", "sgk_stat_long_status", "sgk_stat_plan", "sgk_stat_pa_opt", "sgk_jnn_opt", "sgk_prefix_opt",
    "sgk_stat_host_opt", "sgk_jnn_host_opt", "sgk_prefix_host_opt", "sgk_job_set_options", "sgk_ent", "sgk_ent_finish", "sgk_svbzd_decode",
    "sgk_qts", "sgk_svbzd_size", "sgk_svbzd_encode", "sgk_synth_reads", "sgk_synth_reads_host",
    "sgk_profile_enable", "sgk_profile_reset", "sgk_profile_read",
    "sgk_event_host", "sgk_events_host_free", "sgk_pa_host", "sgk_stat_host", "sgk_jnn_host",
    "sgk_segs_host_free", "sgk_prefix_host", "sgk_signal_in_picoamps", "sgk_getevents",
    "sgk_job_create", "sgk_job_destroy", "sgk_job_device", "sgk_job_begin", "sgk_job_submit", "sgk_job_submit_qts",
    "sgk_job_wait",
    "sgk_job_output",
    # per-read shims with the reference's signatures (csrc/shims.hip)
    "sgk_jnn_raw", "sgk_jnn_pa", "sgk_jnnv2", "sgk_find_adaptor", "sgk_find_polya",
    "sgk_meanf", "sgk_meani16", "sgk_stdvf", "sgk_stdvi16", "sgk_medianf", "sgk_mediani16", "sgk_shim_status",
]


def event_slots_for(n):
    """sgk_event_slots_for() of include/sigtk_gpu.h (a static inline there): arena slots of a read of n samples."""
    return np.asarray(n, dtype=np.int64) // 3 + 2


class Events(NamedTuple):
    start: np.ndarray   # uint32
    length: np.ndarray  # uint32
    mean: np.ndarray    # float32
    stdv: np.ndarray    # float32


_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load libsigtk_gpu.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SigtkGpuError("%s not found: build it with `python -m sigtk_amd.build` "
                            "(__graft_entry__.build())" % path)
    L = C.CDLL(path)
    L.sgk_strerror.restype = C.c_char_p
    L.sgk_version.restype = C.c_char_p
    L.sgk_last_hip_error.restype = C.c_char_p
    for f in ("sgk_event_workspace_bytes", "sgk_stat_workspace_bytes", "sgk_jnn_workspace_bytes",
              "sgk_prefix_workspace_bytes"):
        fn = getattr(L, f)
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
    L.sgk_event.argtypes = [C.POINTER(Batch), C.c_int] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    L.sgk_event_pa.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int] + \
                              [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    L.sgk_event_status.argtypes = [C.c_void_p, C.POINTER(EventStatus), C.c_void_p]
    ver = L.sgk_version().decode()
    old_ok = os.environ.get("SIGTK_AMD_LIB_ANY") == "1"   # development: an A/B build of an earlier round under the event calls
    if tuple(int(x) for x in ver.split(".")[:3]) < (0, 2, 3) and not old_ok:
        raise SigtkGpuError("%s is version %s: these bindings need the per-call options of 0.2, the long-read status of "
                            "0.2.1 and sgk_event_plan_opt / sgk_job_long_declined of 0.2.2" % (path, ver))
    OE, OS = C.POINTER(EventOptions), C.POINTER(StatOptions)
    L.sgk_event_workspace_bytes_opt.restype = C.c_size_t
    L.sgk_event_workspace_bytes_opt.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, OE]
    L.sgk_event_opt.argtypes = L.sgk_event.argtypes + [OE]
    L.sgk_event_pa_opt.argtypes = L.sgk_event_pa.argtypes + [OE]
    if not (old_ok and not hasattr(L, "sgk_event_plan_opt")):
        L.sgk_event_plan_opt.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, OE, C.POINTER(EventPlan)]
        L.sgk_event_plan_opt.restype = C.c_int
        L.sgk_event_plan.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]  # (0.1.0 form, deprecated)
        L.sgk_event_plan.restype = C.c_int
        L.sgk_job_long_declined.argtypes = [C.c_void_p]
        L.sgk_job_long_declined.restype = C.c_uint32
    L.sgk_pa.argtypes = [C.POINTER(Batch), C.c_void_p, C.c_void_p]
    L.sgk_stat.argtypes = [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sgk_stat_pa.argtypes = [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sgk_jnn.argtypes = [C.POINTER(Batch), C.c_int] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    L.sgk_prefix.argtypes = [C.POINTER(Batch), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    for f in ("sgk_stat", "sgk_stat_pa", "sgk_jnn", "sgk_prefix"):
        getattr(L, f + "_opt").argtypes = getattr(L, f).argtypes + [OS]
    L.sgk_stat_long_status.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(LongStatus)]
    L.sgk_stat_plan.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, OS, C.POINTER(StatPlan)]
    L.sgk_pipeline.argtypes = [C.POINTER(Batch), C.c_int] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, OE, OS]
    L.sgk_inflate.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 6
    L.sgk_svbzd_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    L.sgk_synth_reads.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                                     C.c_void_p]
    L.sgk_synth_reads_host.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int]
    L.sgk_synth_reads_host.restype = None
    L.sgk_profile_read.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]
    L.sgk_ent.argtypes = [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sgk_ent_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.sgk_ent_finish.restype = None
    L.sgk_qts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    L.sgk_svbzd_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.sgk_svbzd_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    L.sgk_job_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sgk_job_destroy.argtypes = [C.c_void_p]
    L.sgk_job_destroy.restype = None
    L.sgk_job_device.argtypes = [C.c_void_p]
    L.sgk_job_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(JobInput)]
    L.sgk_job_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.sgk_job_submit_qts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.sgk_job_wait.argtypes = [C.c_void_p]
    L.sgk_job_output.argtypes = [C.c_void_p, C.POINTER(JobOutput)]
    L.sgk_job_set_options.argtypes = [C.c_void_p, OE, OS]
    L.sgk_event_host.argtypes = [C.POINTER(HostBatch), C.c_int, C.POINTER(EventsHost)]
    L.sgk_event_host_opt.argtypes = L.sgk_event_host.argtypes + [OE]
    L.sgk_events_host_free.argtypes = [C.POINTER(EventsHost)]
    L.sgk_events_host_free.restype = None
    L.sgk_pa_host.argtypes = [C.POINTER(HostBatch), C.c_void_p]
    L.sgk_stat_host.argtypes = [C.POINTER(HostBatch), C.c_void_p]
    L.sgk_jnn_host.argtypes = [C.POINTER(HostBatch), C.c_int, C.POINTER(SegsHost)]
    L.sgk_segs_host_free.argtypes = [C.POINTER(SegsHost)]
    L.sgk_segs_host_free.restype = None
    L.sgk_prefix_host.argtypes = [C.POINTER(HostBatch), C.c_int, C.c_int, C.c_void_p]
    L.sgk_stat_host_opt.argtypes = L.sgk_stat_host.argtypes + [OS]
    L.sgk_jnn_host_opt.argtypes = L.sgk_jnn_host.argtypes + [OS]
    L.sgk_prefix_host_opt.argtypes = L.sgk_prefix_host.argtypes + [OS]
    L.sgk_signal_in_picoamps.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double]
    L.sgk_signal_in_picoamps.restype = C.POINTER(C.c_float)
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != SGK_OK:
        L = load_library()
        msg = L.sgk_strerror(rc).decode()
        hip = L.sgk_last_hip_error().decode()
        raise SigtkGpuError("%s failed: %s%s" % (what or "sgk call", msg, (" [" + hip + "]") if hip else ""))


def device_count() -> int:
    return int(load_library().sgk_device_count())


# ---------------------------------------------------------------------- host batches (numpy)

class _HB:
    """Keeps the numpy arrays behind a sgk_host_batch_t alive."""

    def __init__(self, reads: Sequence[np.ndarray], dig, off, rng):
        self.n = len(reads)
        lens = np.array([len(r) for r in reads], dtype=np.uint64)
        self.offsets = np.zeros(self.n + 1, dtype=np.uint64)
        np.cumsum(lens, out=self.offsets[1:])
        self.samples = (np.concatenate([np.asarray(r, dtype=np.int16) for r in reads])
                        if self.n and int(self.offsets[-1]) else np.zeros(1, dtype=np.int16))
        self.samples = np.ascontiguousarray(self.samples, dtype=np.int16)
        self.dig = np.ascontiguousarray(np.broadcast_to(np.asarray(dig, dtype=np.float64), (self.n,)))
        self.off = np.ascontiguousarray(np.broadcast_to(np.asarray(off, dtype=np.float64), (self.n,)))
        self.rng = np.ascontiguousarray(np.broadcast_to(np.asarray(rng, dtype=np.float64), (self.n,)))
        self.c = HostBatch(self.samples.ctypes.data, self.offsets.ctypes.data, self.dig.ctypes.data,
                           self.off.ctypes.data, self.rng.ctypes.data, self.n)


def _np_from(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


# ---------------------------------------------------------------------- operator mirror (numpy in/out)

def pa(reads: Sequence[np.ndarray], dig, off, rng) -> List[np.ndarray]:
    """signal_in_picoamps (src/misc.c:15) for a batch of reads."""
    L = load_library()
    hb = _HB(reads, dig, off, rng)
    out = np.empty(max(int(hb.offsets[-1]), 1), dtype=np.float32)
    check(L.sgk_pa_host(C.byref(hb.c), out.ctypes.data), "sgk_pa_host")
    return [out[int(hb.offsets[r]):int(hb.offsets[r + 1])].copy() for r in range(hb.n)]


def event(reads: Sequence[np.ndarray], dig, off, rng, rna: int):
    """event_func's compute part (src/cfunc.c:72-78) for a batch.  -> (list of Events, EventStatus)"""
    L = load_library()
    hb = _HB(reads, dig, off, rng)
    ev = EventsHost()
    check(L.sgk_event_host_opt(C.byref(hb.c), int(rna), C.byref(ev), C.byref(EVENT_OPTIONS)), "sgk_event_host_opt")
    try:
        offs = _np_from(ev.ev_offsets, hb.n + 1, np.uint64)
        tot = int(offs[-1]) if hb.n else 0
        st = _np_from(ev.start, tot, np.uint32)
        ln = _np_from(ev.length, tot, np.uint32)
        mn = _np_from(ev.mean, tot, np.float32)
        sd = _np_from(ev.stdv, tot, np.float32)
        status = EventStatus.from_buffer_copy(bytes(ev.status))
    finally:
        L.sgk_events_host_free(C.byref(ev))
    out = []
    for r in range(hb.n):
        a, b = int(offs[r]), int(offs[r + 1])
        out.append(Events(st[a:b], ln[a:b], mn[a:b], sd[a:b]))
    return out, status


def stat(reads: Sequence[np.ndarray], dig, off, rng) -> np.ndarray:
    """stat_func's compute part (src/cfunc.c:132-139) -> structured array (STAT_DTYPE)."""
    L = load_library()
    hb = _HB(reads, dig, off, rng)
    out = np.zeros(max(hb.n, 1), dtype=STAT_DTYPE)
    check(L.sgk_stat_host_opt(C.byref(hb.c), out.ctypes.data, C.byref(STAT_OPTIONS)), "sgk_stat_host_opt")
    return out[:hb.n]


def jnn(reads: Sequence[np.ndarray], dig, off, rng, rna: int):
    """jnn_raw with jnn_print's preset (src/jnn.c:282,313-319) -> list of (x, y) int32 arrays."""
    L = load_library()
    hb = _HB(reads, dig, off, rng)
    sg = SegsHost()
    check(L.sgk_jnn_host_opt(C.byref(hb.c), int(rna), C.byref(sg), C.byref(STAT_OPTIONS)), "sgk_jnn_host_opt")
    try:
        offs = _np_from(sg.seg_offsets, hb.n + 1, np.uint64)
        tot = int(offs[-1]) if hb.n else 0
        x = _np_from(sg.x, tot, np.int32)
        y = _np_from(sg.y, tot, np.int32)
    finally:
        L.sgk_segs_host_free(C.byref(sg))
    return [(x[int(offs[r]):int(offs[r + 1])], y[int(offs[r]):int(offs[r + 1])]) for r in range(hb.n)]


def prefix(reads: Sequence[np.ndarray], dig, off, rng, rna: int, pore: int) -> np.ndarray:
    """prefix_func's compute part (src/cfunc.c:169-216) -> structured array (PREFIX_DTYPE)."""
    L = load_library()
    hb = _HB(reads, dig, off, rng)
    out = np.zeros(max(hb.n, 1), dtype=PREFIX_DTYPE)
    check(L.sgk_prefix_host_opt(C.byref(hb.c), int(rna), int(pore), out.ctypes.data, C.byref(STAT_OPTIONS)),
          "sgk_prefix_host_opt")
    return out[:hb.n]


def synth_reads_host(n_reads: int, read_len, seed: int, kind: int, first_read: int = 0):
    """Deterministic synthetic reads (host generator; identical to the device kernel).
    -> (list of int16 arrays, dig, off, rng)"""
    L = load_library()
    lens = np.ascontiguousarray(np.broadcast_to(np.asarray(read_len, dtype=np.uint32), (n_reads,)))
    offs = np.zeros(n_reads, dtype=np.uint64)
    if n_reads > 1:
        np.cumsum(lens[:-1].astype(np.uint64), out=offs[1:])
    total = int(lens.astype(np.uint64).sum())
    samples = np.zeros(max(total, 1), dtype=np.int16)
    dig = np.zeros(max(n_reads, 1)); off = np.zeros(max(n_reads, 1)); rng = np.zeros(max(n_reads, 1))
    L.sgk_synth_reads_host(samples.ctypes.data, offs.ctypes.data, lens.ctypes.data, dig.ctypes.data,
                           off.ctypes.data, rng.ctypes.data, n_reads, first_read, seed, kind)
    reads = [samples[int(offs[r]):int(offs[r]) + int(lens[r])].copy() for r in range(n_reads)]
    return reads, dig[:n_reads], off[:n_reads], rng[:n_reads]


# ---- per-read shims with the reference's own signatures (include/sigtk_gpu.h, csrc/shims.hip) -----------------
class JnnPair(C.Structure):
    _fields_ = [("x", C.c_int64), ("y", C.c_int64)]


class JnnParam(C.Structure):   # jnn_param_t, src/jnn.h:18-27
    _fields_ = [("std_scale", C.c_float), ("corrector", C.c_int), ("seg_dist", C.c_int), ("window", C.c_int),
                ("stall_len", C.c_float), ("error", C.c_int), ("top", C.c_float), ("bot", C.c_float)]


class Jnnv2Param(C.Structure):  # jnnv2_param_t, src/jnn.h:74-81
    _fields_ = [("std_scale", C.c_float), ("seg_dist", C.c_int), ("window", C.c_int), ("stall_len", C.c_float),
                ("hi_thresh", C.c_int), ("lo_thresh", C.c_int)]


def _shim_lib():
    L = load_library()
    if not getattr(L, "_shims_bound", False):
        L.sgk_jnn_raw.restype = C.POINTER(JnnPair)
        L.sgk_jnn_raw.argtypes = [C.c_void_p, C.c_int64, JnnParam, C.POINTER(C.c_int)]
        L.sgk_jnn_pa.restype = C.POINTER(JnnPair)
        L.sgk_jnn_pa.argtypes = [C.c_void_p, C.c_int64, JnnParam, C.POINTER(C.c_int)]
        L.sgk_jnnv2.restype = JnnPair
        L.sgk_jnnv2.argtypes = [C.c_void_p, C.c_int64, Jnnv2Param]
        L.sgk_find_adaptor.restype = JnnPair
        L.sgk_find_adaptor.argtypes = [C.c_void_p, C.c_int64, C.c_int8]
        L.sgk_find_polya.restype = JnnPair
        L.sgk_find_polya.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int8]
        for f in ("sgk_meanf", "sgk_stdvf", "sgk_medianf", "sgk_meani16", "sgk_stdvi16"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.sgk_mediani16.restype = C.c_int16
        L.sgk_mediani16.argtypes = [C.c_void_p, C.c_int]
        L._libc_free = C.CDLL(None).free
        L._libc_free.argtypes = [C.c_void_p]
        L._shims_bound = True
    return L


def _segs(L, ptr, n):
    out = [(int(ptr[k].x), int(ptr[k].y)) for k in range(n)] if ptr else []
    if ptr:
        L._libc_free(ptr)
    return out


def shim_jnn_raw(raw, param: JnnParam):
    """jnn_raw(raw, n, param, &n) -> list of (x, y); raises on a shim error"""
    L = _shim_lib()
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    n = C.c_int(0)
    ptr = L.sgk_jnn_raw(raw.ctypes.data, raw.size, param, C.byref(n))
    check(L.sgk_shim_status(), "sgk_jnn_raw")
    return _segs(L, ptr, n.value)


def shim_jnn_pa(pa, param: JnnParam):
    L = _shim_lib()
    pa = np.ascontiguousarray(pa, dtype=np.float32)
    n = C.c_int(0)
    ptr = L.sgk_jnn_pa(pa.ctypes.data, pa.size, param, C.byref(n))
    check(L.sgk_shim_status(), "sgk_jnn_pa")
    return _segs(L, ptr, n.value)


def shim_jnnv2(raw, param: Jnnv2Param):
    L = _shim_lib()
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    p = L.sgk_jnnv2(raw.ctypes.data, raw.size, param)
    return (int(p.x), int(p.y)), L.sgk_shim_status()


def shim_find_adaptor(raw, pore: int):
    L = _shim_lib()
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    p = L.sgk_find_adaptor(raw.ctypes.data, raw.size, pore)
    check(L.sgk_shim_status(), "sgk_find_adaptor")
    return int(p.x), int(p.y)


def shim_find_polya(pa, top: float, bot: float, pore: int):
    L = _shim_lib()
    pa = np.ascontiguousarray(pa, dtype=np.float32)
    p = L.sgk_find_polya(pa.ctypes.data, pa.size, top, bot, pore)
    check(L.sgk_shim_status(), "sgk_find_polya")
    return int(p.x), int(p.y)


def shim_stat_f32(x):
    """(meanf, stdvf, medianf) of a float array through the three shims"""
    L = _shim_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = tuple(np.float32(getattr(L, f)(x.ctypes.data, x.size)) for f in ("sgk_meanf", "sgk_stdvf", "sgk_medianf"))
    check(L.sgk_shim_status(), "sgk_*f")
    return out


def shim_stat_i16(x):
    L = _shim_lib()
    x = np.ascontiguousarray(x, dtype=np.int16)
    out = (np.float32(L.sgk_meani16(x.ctypes.data, x.size)), np.float32(L.sgk_stdvi16(x.ctypes.data, x.size)),
           int(L.sgk_mediani16(x.ctypes.data, x.size)))
    check(L.sgk_shim_status(), "sgk_*i16")
    return out


def profile_read():
    """-> {kernel name: (total ms, calls)} accumulated since the last profile_reset()."""
    L = load_library()
    cap = 32
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    calls = (C.c_uint32 * cap)()
    k = L.sgk_profile_read(names, ms, calls, cap)
    return {names[i].decode(): (ms[i], calls[i]) for i in range(k)}


class Job:
    """A pipelined host job (sgk_job_*): pinned staging + device buffers + stream for one batch at a time.

    `signals` is a list of int16 arrays (SIGNAL_INT16) or of svb-zd blobs as bytes (SIGNAL_SVBZD, with
    `counts` = samples per read); results come back as numpy copies."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        self.h = C.c_void_p()
        check(self.L.sgk_job_create(device, C.byref(self.h)), "sgk_job_create")
        self.set_options()

    def set_options(self, event_opt: "EventOptions" = None, stat_opt: "StatOptions" = None):
        """the options the job's submits use from now on (default: this module's EVENT_OPTIONS / STAT_OPTIONS as they
        are NOW: the job keeps a copy)"""
        check(self.L.sgk_job_set_options(self.h, C.byref(event_opt if event_opt is not None else EVENT_OPTIONS),
                                         C.byref(stat_opt if stat_opt is not None else STAT_OPTIONS)), "sgk_job_set_options")

    def close(self):
        if self.h:
            self.L.sgk_job_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may already be gone
            pass

    def stage(self, signals, dig, off, rng, counts=None):
        """sgk_job_begin + fill the pinned staging (int16 arrays, or svb-zd blobs when `counts` is given)"""
        n = len(signals)
        svb = counts is not None
        lengths = np.asarray(counts if svb else [len(x) for x in signals], dtype=np.uint32)
        blens = np.asarray([len(b) for b in signals], dtype=np.uint32) if svb else None
        jin = JobInput()
        check(self.L.sgk_job_begin(self.h, n, lengths.ctypes.data, SIGNAL_SVBZD if svb else SIGNAL_INT16,
                                   blens.ctypes.data if svb else None, C.byref(jin)), "sgk_job_begin")
        for r in range(n):
            jin.digitisation[r] = float(dig[r]); jin.offset[r] = float(off[r]); jin.range[r] = float(rng[r])
            if svb:
                C.memmove(jin.blobs + jin.blob_offsets[r], signals[r], len(signals[r]))
            elif lengths[r]:
                x = np.ascontiguousarray(signals[r], dtype=np.int16)
                C.memmove(jin.samples + 2 * jin.offsets[r], x.ctypes.data, x.nbytes)

    def launch(self, tool: int, rna: int = 0, pore: int = 0, flags: int = 0):
        """sgk_job_submit on the staged batch (asynchronous; may be repeated after wait())"""
        self._tool = tool
        check(self.L.sgk_job_submit(self.h, tool, rna, pore, flags), "sgk_job_submit")

    def launch_qts(self, bits: int, method: int, out_svb: bool):
        self._tool = TOOL_QTS
        self._qts_svb = out_svb
        check(self.L.sgk_job_submit_qts(self.h, bits, method, SIGNAL_SVBZD if out_svb else SIGNAL_INT16),
              "sgk_job_submit_qts")

    def submit(self, tool: int, signals, dig, off, rng, rna: int = 0, pore: int = 0, flags: int = 0, counts=None):
        self.stage(signals, dig, off, rng, counts)
        self.launch(tool, rna, pore, flags)

    def wait(self):
        """-> dict of numpy results (per read lists for the variable-length outputs)"""
        check(self.L.sgk_job_wait(self.h), "sgk_job_wait")
        o = JobOutput()
        check(self.L.sgk_job_output(self.h, C.byref(o)), "sgk_job_output")
        n = o.n_reads
        res = {"n_reads": n, "lengths": _np_from(o.lengths, n, np.uint32).copy()}
        if self._tool == TOOL_PA:
            res["pa"] = [_np_from(C.cast(C.addressof(o.pa.contents) + 4 * o.offsets[r], C.POINTER(C.c_float)),
                                  int(o.lengths[r]), np.float32).copy() if o.lengths[r] else np.zeros(0, np.float32)
                         for r in range(n)]
        elif self._tool in (TOOL_EVENT, TOOL_JNN):
            def seg(ptr, dtype, r):
                k = int(o.counts[r])
                if not k or not ptr:
                    return np.zeros(0, dtype)
                base = C.addressof(ptr.contents) + 4 * o.slots[r]
                return _np_from(C.cast(base, C.POINTER(C.c_uint32)), k, np.uint32).view(dtype).copy()
            if self._tool == TOOL_EVENT:
                def starts(r):
                    if o.ev_start:
                        return seg(o.ev_start, np.uint32, r)
                    ln = seg(o.ev_length, np.uint32, r).astype(np.uint64)
                    return (np.cumsum(ln) - ln).astype(np.uint32)
                res["events"] = [Events(starts(r), seg(o.ev_length, np.uint32, r),
                                        seg(o.ev_mean, np.float32, r), seg(o.ev_stdv, np.float32, r))
                                 for r in range(n)]
                res["status"] = o.event_status
            else:
                res["segs"] = [(seg(o.seg_x, np.int32, r), seg(o.seg_y, np.int32, r)) for r in range(n)]
        elif self._tool == TOOL_STAT:
            res["stat"] = np.frombuffer(C.string_at(o.stat, n * STAT_DTYPE.itemsize), dtype=STAT_DTYPE).copy()
        elif self._tool == TOOL_PREFIX:
            res["prefix"] = np.frombuffer(C.string_at(o.prefix, n * PREFIX_DTYPE.itemsize), dtype=PREFIX_DTYPE).copy()
        elif self._tool == TOOL_QTS:
            if self._qts_svb:
                res["blobs"] = [C.string_at(o.qts_blobs + int(o.qts_blob_offsets[r]), int(o.qts_blob_lengths[r]))
                                for r in range(n)]
            else:
                res["samples"] = [np.frombuffer(C.string_at(o.qts_samples + 2 * int(o.offsets[r]), 2 * int(o.lengths[r])),
                                                dtype=np.int16).copy() for r in range(n)]
        elif self._tool == TOOL_ENT:
            ent = np.zeros((n, 3), dtype=np.float64)
            for r in range(n):
                off = int(o.offsets[r]) * 2
                self.L.sgk_ent_finish(o.ent + r * ENT_HIST_BYTES,
                                      (o.ent_over_raw + off) if o.ent_over_raw else None,
                                      (o.ent_over_delta + off) if o.ent_over_delta else None,
                                      ent[r].ctypes.data_as(C.POINTER(C.c_double)))
            res["ent"] = ent
        return res
