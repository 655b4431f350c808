"""CPU: the C-ABI shared library loads and exports every function include/sigtk_gpu.h declares;
error behaviour without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest

from sigtk_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sigtk_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(sgk_[a-z0-9_]+)\s*\(", src))
    names -= {"sgk_event_slots_for", "sgk_jnn_slots_for"}  # static inline helpers
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    decl = declared_functions()
    assert len(decl) >= 25
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, "missing from libsigtk_gpu.so: %s" % missing
    assert sorted(api.ABI_SYMBOLS) == decl, "api.ABI_SYMBOLS out of sync with the header"


def test_version_and_strerror():
    lib = api.load_library()
    assert lib.sgk_version().decode() == "0.2.3"   # per-call options (0.2.0), long-read status (0.2.1), decline-and-redo + sgk_event_plan_opt (0.2.2), SGK_JOB_EVENTS_LENGTHS (0.2.3)
    assert lib.sgk_strerror(0).decode() == "ok"
    assert "GPU" in lib.sgk_strerror(-3).decode()


def test_library_is_reentrant_by_construction():
    """no getenv and no mutable process-wide configuration in libsigtk_gpu.so (VERDICT r03 task 7): options travel with
    every call (sgk_event_options_t, sgk_stat_options_t)"""
    csrc = os.path.join(ROOT, "sigtk_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r"//.*", "", src)
        assert "getenv" not in src, f
    lib = api.load_library()
    assert not hasattr(lib, "sgk_event_configure") and not hasattr(lib, "sgk_event_configure_short")
    # two option sets give two plans from the same library at the same time
    a, b = api.EventOptions(), api.EventOptions()
    a.segment_len, a.long_min = 4096, 10000
    pa_, pb_ = api.event_plan(10, 10 * 300000, 300000, 0, a), api.event_plan(10, 10 * 300000, 300000, 0, b)
    assert (pa_.segment_len, pb_.segment_len) == (4096, 65536)   # (the default geometry of a batch this small, api.hip: event_config_for)


def test_no_cpu_fallback_without_gpu():
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    reads = [np.arange(300, dtype=np.int16)]
    for call in (lambda: api.event(reads, 8192.0, 0.0, 1400.0, 0), lambda: api.pa(reads, 8192.0, 0.0, 1400.0),
                 lambda: api.stat(reads, 8192.0, 0.0, 1400.0), lambda: api.jnn(reads, 8192.0, 0.0, 1400.0, 0),
                 lambda: api.prefix(reads, 8192.0, 0.0, 1400.0, 0, 0)):
        with pytest.raises(api.SigtkGpuError, match="no usable GPU"):
            call()


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(api, "_lib", None)
    with pytest.raises(api.SigtkGpuError, match="not found"):
        api.load_library(str(tmp_path / "nope.so"))
    monkeypatch.setattr(api, "_lib", None)
    api.load_library()


def test_synth_generator_is_deterministic_and_plausible():
    a, dig, off, rng = api.synth_reads_host(3, 50000, 42, 0)
    b, _, _, _ = api.synth_reads_host(3, 50000, 42, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    c, _, _, _ = api.synth_reads_host(1, 50000, 42, 0, first_read=2)
    assert np.array_equal(a[2], c[0])  # read index, not batch position, seeds a read
    pa_mean = (a[0].astype(np.float64) + off[0]).mean() * rng[0] / dig[0]
    assert 80 < pa_mean < 100 and a[0].min() >= 0 and a[0].max() <= 4000
