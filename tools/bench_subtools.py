#!/usr/bin/env python3
"""Throughput of the non-headline subtools (pa, stat, stat+pa, jnn, prefix) on device-resident
synthetic reads; prints one JSON object per subtool with per-kernel HIP-event times and the
achieved fraction of the HBM roofline on ALGORITHMIC bytes.
    python tools/bench_subtools.py [--reads 20000] [--read-len 100000] [--rna 1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--rna", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--pipeline", type=int, default=0, help="1: also time the config-5 pa->event->stat pipeline")
    ap.add_argument("--ragged", type=float, default=0.0, help="sigma of log-normal read lengths with mean --read-len "
                    "(0: all reads --read-len samples)")
    ap.add_argument("--sorted", type=int, default=0, help="1: ragged reads laid out longest first")
    ap.add_argument("--one-long", type=int, default=0, help="N > 0: the batch's middle read has N samples (the others "
                    "keep their length): what a single very long read costs the batch")
    args = ap.parse_args()
    import torch
    from sigtk_amd import api, device
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = api.load_library()
    lens = None
    if args.ragged > 0:
        import numpy as np
        rs = np.random.RandomState(5)
        lens = args.read_len * np.exp(rs.normal(-0.5 * args.ragged ** 2, args.ragged, size=args.reads))
        lens = np.clip(lens, 2500, 16 * args.read_len).astype(np.int64)
        if args.sorted:
            lens = np.sort(lens)[::-1].copy()
    if args.one_long > 0:
        import numpy as np
        if lens is None:
            lens = np.full(args.reads, args.read_len, dtype=np.int64)
        lens[args.reads // 2] = args.one_long
    b = device.synth_reads(args.reads, args.read_len, seed=2, kind=args.rna, device=dev, lengths=lens)
    S, R = b.total_samples, b.n_reads
    pa_out = torch.empty(b.n_samples, dtype=torch.float32, device=dev)
    segs = device.SegArena(b)

    def run(name, fn, alg_bytes, long_ws=None):
        for _ in range(4):   # (the library's side streams are created on first use, four per device)
            fn()
        torch.cuda.synchronize()
        ls = device.long_status(b, ws=long_ws) if long_ws is not None else None
        L.sgk_profile_reset(); L.sgk_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        L.sgk_profile_enable(0)
        prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in api.profile_read().items()}
        L.sgk_profile_reset()
        print(json.dumps({"subtool": name, "reads": R, "samples": S, "ms": round(dt * 1e3, 3),
                          "samples_per_s": round(S / dt, 1), "reads_per_s": round(R / dt, 1),
                          "algorithmic_GBps": round(alg_bytes / dt / 1e9, 1),
                          "hbm_frac": round(alg_bytes / dt / 1e9 / 8000.0, 4), "kernels_ms": prof,
                          "longest_read": int(b.max_read_len), "stat_long_min": int(api.STAT_OPTIONS.long_min),
                          "long_reads": None if ls is None else {"reads": ls.n_long_reads, "tile_sums": ls.n_tiles,
                                                                 "evaluated_from_true_accumulator": ls.n_true_tiles,
                                                                 "barrier_timeouts": ls.n_timeouts}}),
              flush=True)

    run("pa", lambda: device.pa(b, pa_out), 6 * S)
    run("stat", lambda: device.stat(b), 2 * S + 32 * R, device._workspace(b, "sgk_stat_workspace_bytes"))
    run("stat+pa", lambda: device.stat_pa(b, pa_out), 6 * S + 32 * R,   # BASELINE config 4 (fused)
        device._workspace(b, "sgk_stat_workspace_bytes"))
    if args.pipeline:
        # BASELINE config 5: pa -> event -> stat over the same resident batch (pA is not materialised for the
        # event / stat kernels: they scale on the fly; the fused stat+pa pass writes it once)
        arena = device.EventArena(b)
        device.event(b, arena, args.rna); torch.cuda.synchronize()
        E = int(arena.status().n_events_total)

        def pipe():
            device.stat_pa(b, pa_out)
            device.event(b, arena, args.rna)
        run("pa->event->stat", pipe, 2 * S + 16 * E + 72 * R)
    run("jnn", lambda: device.jnn(b, segs, args.rna), 2 * S, segs.ws)
    run("prefix", lambda: device.prefix(b, args.rna, 0), 2 * S + 48 * R, device._workspace(b, "sgk_prefix_workspace_bytes"))


if __name__ == "__main__":
    main()
