#!/usr/bin/env python3
"""development (round 5): BASELINE config 5's two halves -- stat+pa (HBM-bound) and event (issue-bound) -- one after the
other on one stream against the two on two streams at once.
    python tools/r05_concurrent.py [--reads 50000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from sigtk_amd import device
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    b = device.synth_reads(a.reads, 100000, seed=4, kind=0, device=dev)
    arena = device.EventArena(b)
    pa = torch.empty(b.n_samples, dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def serial():
        device.stat_pa(b, pa)
        device.event(b, arena, 0)

    def conc(event_first):
        def f():
            if event_first:
                with torch.cuda.stream(s1):
                    device.event(b, arena, 0)
                with torch.cuda.stream(s2):
                    device.stat_pa(b, pa)
            else:
                with torch.cuda.stream(s2):
                    device.stat_pa(b, pa)
                with torch.cuda.stream(s1):
                    device.event(b, arena, 0)
        return f

    def only_event():
        device.event(b, arena, 0)

    def only_stat():
        device.stat_pa(b, pa)

    for name, fn in (("event", only_event), ("stat+pa", only_stat), ("serial", serial), ("concurrent, event first", conc(True)),
                     ("concurrent, stat+pa first", conc(False))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
            torch.cuda.synchronize()
        print("%-28s %.2f ms" % (name, (time.perf_counter() - t0) / a.steps * 1e3))


if __name__ == "__main__":
    main()
