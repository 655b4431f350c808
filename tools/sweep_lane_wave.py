#!/usr/bin/env python3
"""development: lane-per-read against wave-per-read kernels of stat / stat+pA / jnn / prefix over a grid of batch shapes
(uniform read lengths): the data behind LANE_RULES (csrc/stat_args.h).  python tools/sweep_lane_wave.py > profiles/...txt"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigtk_amd import api, device
dev = torch.device("cuda", 0)
pa_buf = torch.empty(int(1.4e10), dtype=torch.float32, device=dev)


def timed(fn, kernels):
    api.stat_configure(kernels)
    try:
        fn(); fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    finally:
        api.stat_configure(0)


for length in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 100000, 131072, 200000):
    for n in (4096, 8192, 16384, 24576, 32768, 49152, 65536, 81920, 98304, 131072, 196608):
        if n * length > 1.3e10 or n * length < 3e7:
            continue
        row = {"reads": n, "len": length}
        for kind, tools in ((0, ("stat", "stat_pa", "jnn")), (1, ("prefix",))):
            b = device.synth_reads(n, length, seed=5, kind=kind, device=dev)
            for tool in tools:
                if tool == "stat": fn = lambda: device.stat(b)
                elif tool == "stat_pa": fn = lambda: device.stat_pa(b, pa_buf[:b.n_samples])
                elif tool == "jnn":
                    ar = device.SegArena(b); fn = lambda: device.jnn(b, ar, 0)
                else: fn = lambda: device.prefix(b, 1, 0)
                tl, tw = timed(fn, 1), timed(fn, 2)
                row[tool] = [round(tl, 3), round(tw, 3), round(tl / tw, 2)]
            del b
            torch.cuda.empty_cache()
        print(json.dumps(row), flush=True)
