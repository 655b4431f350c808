#!/usr/bin/env python3
"""Randomised differential test of the two implementations of stat / jnn / prefix at batch scale: the wave-per-read
kernels (default; seqsum.h chains, chunked automata, longest-first dispatch) against the lane-per-read kernels of round 1
(sgk_stat_options_t::kernels = 1) through the job API, on batches of 1 000 - 6 000 reads with log-normal lengths and the hostile
scalings of tests/soak_parity.py.  Every record must be identical bit for bit.
    python tests/soak_wave_vs_lane.py [--minutes 5] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--batches", type=int, default=0,
                    help="run exactly this many batches (the same work on every box) instead of a time box")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    from sigtk_amd import api
    api.load_library()
    rs = np.random.RandomState(a.seed)
    t_end = time.time() + a.minutes * 60
    stats = {"batches": 0, "reads": 0, "samples": 0, "long_min": {}, "reads_at_or_above_long_min": 0, "mismatches": []}

    def fail(msg):
        stats["mismatches"].append(msg)
        print("MISMATCH", msg, flush=True)

    while (stats["batches"] < a.batches if a.batches else time.time() < t_end) and len(stats["mismatches"]) < 5:
        kind = int(rs.randint(0, 2))
        nr = int(rs.randint(1000, 6000))
        mu = rs.uniform(7.0, 9.5)
        lens = [int(x) for x in np.clip(np.exp(rs.normal(mu, rs.uniform(0.2, 1.2), size=nr)), 0, 400000)]
        seed = int(rs.randint(0, 1 << 30))
        reads, dig, off, rng = api.synth_reads_host(nr, lens, seed, kind)
        dig = np.asarray(dig, dtype=np.float64).copy(); off = np.asarray(off, dtype=np.float64).copy()
        rng = np.asarray(rng, dtype=np.float64).copy()
        reads = list(reads)
        for r in rs.randint(0, nr, size=nr // 8):
            u = rs.rand()
            n = lens[r]
            if u < 0.2: rng[r] = -rng[r]
            elif u < 0.35: off[r] = off[r] + rs.uniform(-3, 3)
            elif u < 0.45 and n > 0: off[r] = -float(reads[r][int(rs.randint(0, n))]) + float(rs.choice([0.0, 1e-3, -1e-4]))
            elif u < 0.55: reads[r] = np.full(n, int(rs.randint(-100, 2000)), dtype=np.int16)
            elif u < 0.65: reads[r] = np.where(rs.rand(n) < 0.5, -32768, 32767).astype(np.int16)
            elif u < 0.75: reads[r] = np.clip(reads[r].astype(np.int32) * 40 - 9000, -32768, 32767).astype(np.int16)
            elif u < 0.85: reads[r] = rs.randint(-2000, 2000, size=n).astype(np.int16)
        rna = int(rs.randint(0, 2)); pore = int(rs.choice([0, 2]))
        tag = "batch %d (seed %d kind %d rna %d pore %d, %d reads)" % (stats["batches"], seed, kind, rna, pore, nr)
        if rs.rand() < 0.3:   # a few much longer reads: several rounds of the chunk merge, thousands of tiles
            for r in rs.randint(0, nr, size=3):
                lens[r] = int(rs.randint(300000, 1500000))
                reads[r] = api.synth_reads_host(1, [lens[r]], int(rs.randint(0, 1 << 30)), kind)[0][0]
        out = {}
        # the wave kernels' long-read path (k_long_chains: sums from tile summaries, histogram, automaton on 64 waves):
        # a low threshold sends hundreds of reads per batch through it, the default only the longest
        long_min = int(rs.choice([0, 8192, 8192, 20000, 60000, -1]))
        stats["long_min"][str(long_min)] = stats["long_min"].get(str(long_min), 0) + 1
        stats["reads_at_or_above_long_min"] += sum(1 for n in lens if long_min >= 0 and n >= (long_min or 262144))
        for mode in ("wave", "lane"):
            api.stat_configure(1 if mode == "lane" else 2, long_min)
            job = api.Job(0)
            job.stage(reads, dig, off, rng, None)
            job.launch(api.TOOL_STAT); st = job.wait()["stat"].copy()
            job.launch(api.TOOL_JNN, rna=rna); sg = [(x.copy(), y.copy()) for x, y in job.wait()["segs"]]
            job.launch(api.TOOL_PREFIX, rna=rna, pore=pore); pf = job.wait()["prefix"].copy()
            out[mode] = (st, sg, pf)
            job.close()
        api.stat_configure(0, 0)
        w, l = out["wave"], out["lane"]
        for r in range(nr):
            if w[0][r].tobytes() != l[0][r].tobytes():
                fail("%s stat read %d len %d: wave %r lane %r" % (tag, r, lens[r], w[0][r], l[0][r]))
            if not (np.array_equal(w[1][r][0], l[1][r][0]) and np.array_equal(w[1][r][1], l[1][r][1])):
                fail("%s jnn read %d len %d" % (tag, r, lens[r]))
            a_, b_ = w[2][r], l[2][r]
            same = all(int(a_[k]) == int(b_[k]) for k in ("adapt_x", "adapt_y", "polya_x", "polya_y"))
            if same and int(a_["adapt_y"]) > 0:
                same = all(np.float32(a_[k]).view(np.uint32) == np.float32(b_[k]).view(np.uint32) or
                           (np.isnan(a_[k]) and np.isnan(b_[k])) for k in ("adapt_mean", "adapt_std", "adapt_median"))
            if same and int(a_["polya_y"]) > 0:
                same = all(np.float32(a_[k]).view(np.uint32) == np.float32(b_[k]).view(np.uint32) or
                           (np.isnan(a_[k]) and np.isnan(b_[k])) for k in ("polya_mean", "polya_std", "polya_median"))
            if not same:
                fail("%s prefix read %d len %d: wave %r lane %r" % (tag, r, lens[r], a_, b_))
        stats["batches"] += 1
        stats["reads"] += nr
        stats["samples"] += int(sum(lens))
    print(json.dumps(stats))
    sys.exit(1 if stats["mismatches"] else 0)


if __name__ == "__main__":
    main()
