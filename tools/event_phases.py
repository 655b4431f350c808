#!/usr/bin/env python3
"""development (VERDICT r04 task 1a / 1b): where k_event's time and instructions go, by phase, and what the first round
of waves does differently -- with the instrumented build of the library:

    tools/build_variant.sh dev -DSGK_DEV=1
    SIGTK_AMD_LIB=sigtk_amd/_variants/libsigtk_gpu_dev.so python tools/event_phases.py [--reads 10000] [--rna 0] \
            [--trace out.npz] [--steps 5]

Every mode (csrc/event_args.h: SGK_DEV_*) is launched `steps` times in the order of MODES below; under
`rocprofv3 --pmc ...` the k_event dispatches therefore come in groups of `steps` per mode (tools/event_phases_pmc.py
reads them back).  --trace: one extra launch with per-wave timestamps (wave start / detector end / builder end in
s_memrealtime ticks of 10 ns, HW_ID, XCC_ID), saved as an .npz."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sigtk_amd import api, device

NO_BUILD, NO_DETECT, NO_ROUNDS, RAW_EVENTS, TRACE = 1, 2, 4, 8, 16
MODES = [("full", 0), ("detector+bitmap", NO_BUILD), ("full-rounds (walk only)", NO_ROUNDS),
         ("full-arith (raw events)", RAW_EVENTS), ("builder only", NO_DETECT), ("builder walk only", NO_DETECT | NO_ROUNDS)]

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10000)
ap.add_argument("--read-len", type=int, default=100000)
ap.add_argument("--rna", type=int, default=0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--tail", type=int, default=-1, help="tail split: -1 off (default: every workgroup is a whole read), 0 the plan's")
ap.add_argument("--trace", default=None)
ap.add_argument("--modes", default=None, help="comma-separated subset of mode indices")
ap.add_argument("--prio", default=None, help="comma-separated issue-priority policies (0 none, 1 by progress, 2 ... and builder first, 3 builder first only): "
                "one extra full launch group each (dev bits 8..10 = policy + 1)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = api.load_library()
assert hasattr(L, "sgk_debug_scratch_offset"), "needs the SGK_DEV build (SIGTK_AMD_LIB=...)"
api.EVENT_OPTIONS.tail_split = a.tail
b = device.synth_reads(a.reads, a.read_len, seed=1, kind=a.rna, device=dev)
arena = device.EventArena(b)


def kms(prof):
    """the event kernels of one launch: k_event + k_event_seg (summed: they run one behind the other or side by side)"""
    if "path:event" in prof:   # the whole launch on the caller's stream (order + plan + kernels + joins + fallback)
        return round(prof["path:event"][0] / prof["path:event"][1], 4), {k: round(v[0] / v[1], 4) for k, v in prof.items()}
    return round(sum(prof[k][0] / prof[k][1] for k in ("k_event", "k_event_seg") if k in prof), 4), \
        {k: round(v[0] / v[1], 4) for k, v in prof.items()}


def run(mode, n):
    api.EVENT_OPTIONS.reserved[0] = mode
    arena.opt = api.EventOptions.from_buffer_copy(bytes(api.EVENT_OPTIONS))
    for _ in range(n):
        device.event(b, arena, a.rna)
    torch.cuda.synchronize()


run(0, 2)   # warm-up; leaves the bitmap the builder-only modes read
res = {"lib": os.environ.get("SIGTK_AMD_LIB", "default"), "reads": a.reads, "read_len": a.read_len, "rna": a.rna,
       "steps": a.steps, "tail_split": a.tail, "modes": []}
L.sgk_profile_enable(1)
sel = range(len(MODES)) if a.modes is None else [int(x) for x in a.modes.split(",")]
for i in sel:
    name, mode = MODES[i]
    L.sgk_profile_reset()
    run(mode, a.steps)
    prof = api.profile_read()
    res["modes"].append({"name": name, "dev": mode, "k_event_ms": kms(prof)[0], "kernels_ms": kms(prof)[1]})
if a.prio:
    for pol in [int(x) for x in a.prio.split(",")]:
        L.sgk_profile_reset()
        run((pol + 1) << 8, a.steps)
        prof = api.profile_read()
        res["modes"].append({"name": "full, priority policy %d" % pol, "dev": (pol + 1) << 8, "k_event_ms": kms(prof)[0], "kernels_ms": kms(prof)[1]})
L.sgk_profile_enable(0)
if a.trace:
    L.sgk_debug_scratch_offset.restype = C.c_size_t
    L.sgk_debug_scratch_offset.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t]
    off = int(L.sgk_debug_scratch_offset(b.n_reads, b.n_samples, b.max_read_len, C.byref(arena.opt), arena.ws_bytes))
    plan = api.event_plan(b.n_reads, b.n_samples, b.max_read_len, a.rna, arena.opt)
    nwg = a.reads + (plan.max_segments if plan.tail_segment_len or plan.max_segments else 0)
    arena.ws[off:off + 32 * nwg].zero_()
    run(TRACE | (((int(a.prio.split(",")[-1]) + 1) << 8) if a.prio else 0), 1)
    tr_all = arena.ws[off:off + 32 * nwg].cpu().numpy().view(np.uint64).reshape(nwg, 4).copy()
    np.savez_compressed(a.trace, trace=tr_all)
    is_seg = (tr_all[:, 3] >> 63) != 0
    segs = tr_all[is_seg & (tr_all[:, 0] != 0)]
    tr = tr_all[~is_seg & (tr_all[:, 0] != 0)]
    n = len(tr)
    if len(segs):
        t0s = tr_all[tr_all[:, 0] != 0][:, 0].min()
        det = (segs[:, 1] - segs[:, 0]) / 100.0
        seam = (((segs[:, 3] >> 16) & 0x7fffffffffff).astype(np.float64)) / 100.0 - det
        bld = (segs[:, 2] - segs[:, 0]) / 100.0 - det - seam
        res["segments"] = {"n": int(len(segs)), "start_us_min_med_max": [round(float(x), 1) for x in ((segs[:, 0].min() - t0s) / 100.0, (np.median(segs[:, 0]) - t0s) / 100.0, (segs[:, 0].max() - t0s) / 100.0)],
                           "detector_us_p10_med_p90": [round(float(np.percentile(det, q)), 1) for q in (10, 50, 90)],
                           "wait_seam_publish_us_p10_med_p90": [round(float(np.percentile(seam, q)), 1) for q in (10, 50, 90)],
                           "builder_us_p10_med_p90": [round(float(np.percentile(bld, q)), 1) for q in (10, 50, 90)],
                           "end_us_max": round(float((segs[:, 2].max() - t0s) / 100.0), 1),
                           "by_g_wait_median": {int(gg): round(float(np.median(seam[(segs[:, 3] & 0xffff) == gg])), 1) for gg in sorted(set(int(x) for x in segs[:, 3] & 0xffff))}}
    t0 = tr[:, 0].min()
    st, de, be = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0   # microseconds
    order = np.argsort(st)
    slots = 256 * 4 * (2 if a.rna else 3)
    rounds = []
    for k in range(0, n, slots):
        idx = order[k:k + slots]
        rounds.append({"waves": int(idx.size), "start_us": [round(float(st[idx].min()), 1), round(float(np.median(st[idx])), 1), round(float(st[idx].max()), 1)],
                       "detector_us_median": round(float(np.median(de[idx] - st[idx])), 1),
                       "builder_us_median": round(float(np.median(be[idx] - de[idx])), 1),
                       "detector_us_p10_p90": [round(float(np.percentile(de[idx] - st[idx], 10)), 1), round(float(np.percentile(de[idx] - st[idx], 90)), 1)],
                       "builder_us_p10_p90": [round(float(np.percentile(be[idx] - de[idx], 10)), 1), round(float(np.percentile(be[idx] - de[idx], 90)), 1)],
                       "end_us_max": round(float(be[idx].max()), 1)})
    res["trace"] = {"file": a.trace, "kernel_span_us": round(float(be.max()), 1), "rounds_by_start_order": rounds,
                    "xcc_ids": sorted(set(int(x) for x in (tr[:, 3] >> 32) & 0xf))}
print(json.dumps(res))
