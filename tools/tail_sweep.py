#!/usr/bin/env python3
"""Tail split on / off over batch sizes, whole sgk_event calls of the shipped library (HIP events around the call on the
caller's stream, best of 5): what api.hip: event_tail_plan's rule is fitted to, and what
tests/test_gpu_event_long.py::test_tail_split_rule_is_no_cliff guards.
    python tools/tail_sweep.py [--rna 0] > profiles/<round>_tail_split_sweep.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rna", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=100000)
    a = ap.parse_args()
    import torch
    from sigtk_amd import api, device
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    api.load_library()
    slots = 2048 if a.rna else 3072

    def timed(b, tail):
        old = api.EVENT_OPTIONS.tail_split
        api.EVENT_OPTIONS.tail_split = tail
        try:
            arena = device.EventArena(b)
        finally:
            api.EVENT_OPTIONS.tail_split = old
        for _ in range(3):
            device.event(b, arena, a.rna)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); device.event(b, arena, a.rna); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    sizes = [slots * f // 12 for f in range(1, 12)]
    for k in (1, 2, 3, 5):
        sizes += [k * slots + slots * f // 24 for f in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16)]
    print("# %s preset, %d-sample reads, %d wave slots: reads, last round / slots, ms whole, ms split, the rule's choice"
          % ("RNA" if a.rna else "DNA", a.read_len, slots))
    for n in sizes:
        b = device.synth_reads(n, a.read_len, seed=9, kind=a.rna, device=dev)
        rem = n % slots
        t_off, t_on = timed(b, -1), timed(b, rem)
        rule = api.event_plan(n, b.total_samples, a.read_len, a.rna).tail_segment_len != 0
        print("%6d  %.3f  %.3f  %.3f  %s%s" % (n, rem / slots, t_off, t_on, "split" if rule else "whole",
                                            "" if (t_on < t_off) == rule or abs(t_on - t_off) < 0.03 * t_off else "   <-- the other is faster"))
        sys.stdout.flush()
        del b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
