#!/usr/bin/env python3
"""Randomised differential test: every subtool through the job API (svb-zd and int16 input alternating) against
the oracle, on batches of reads with random lengths, kinds, seeds and scalings (incl. negative range, fractional
offsets, tiny digitisation), with a few adversarial reads mixed in (constant, saturated, alternating extremes).
    python tests/soak_parity.py [--minutes 3] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--batches", type=int, default=0,
                    help="run exactly this many batches (the same work on every box) instead of a time box")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--trace", default="", help="file that receives every batch's tag BEFORE its event launch (flushed): "
                    "after a GPU fault the last line names the batch")
    ap.add_argument("--segments", type=float, default=0.5,
                    help="share of the batches whose event pass runs with short segments (reads shared by several "
                         "wavefronts: the segment options) and, for a third of those, a warm-up short enough for "
                         "speculation to fail at seams")
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    from sigtk_amd import api, blow5
    from oracle.oracle import Oracle
    api.load_library()
    orc = Oracle()
    rs = np.random.RandomState(a.seed)
    rs_cfg = np.random.RandomState(a.seed ^ 0x5E6)
    job = api.Job(0)
    t_end = time.time() + a.minutes * 60
    stats = {"batches": 0, "reads": 0, "samples": 0, "fallback_reads": 0, "rerun_chunks": 0, "split_reads": 0,
             "segments": 0, "seam_reruns": 0, "mismatches": []}
    L = api.load_library()

    def fail(msg):
        stats["mismatches"].append(msg)
        print("MISMATCH", msg, flush=True)

    while (stats["batches"] < a.batches if a.batches else time.time() < t_end) and len(stats["mismatches"]) < 5:
        if stats["batches"] % 500 == 0:
            print("progress", json.dumps({k: v for k, v in stats.items() if k != "mismatches"}), flush=True)
        kind = int(rs.randint(0, 2))
        if rs.rand() < 0.1:   # many tiny reads: slot / offset arithmetic, reads shorter than every window
            nr = int(rs.randint(200, 2000))
            lens = [int(x) for x in rs.randint(0, 500, size=nr)]
        else:
            nr = int(rs.randint(1, 24))
            lens = [int(x) for x in np.exp(rs.uniform(np.log(1), np.log(200000), size=nr)).astype(np.int64)]
        seed = int(rs.randint(0, 1 << 30))
        reads, dig, off, rng = api.synth_reads_host(nr, lens, seed, kind)
        dig = np.asarray(dig, dtype=np.float64).copy(); off = np.asarray(off, dtype=np.float64).copy()
        rng = np.asarray(rng, dtype=np.float64).copy()
        for r in range(nr):
            u = rs.rand()
            if u < 0.15: rng[r] = -rng[r]
            if u > 0.85: off[r] = off[r] + rs.uniform(-3, 3)          # fractional offsets
            if 0.4 < u < 0.45: dig[r] = 2048.0
            if 0.5 < u < 0.6 and lens[r] > 0:
                # a sample whose pA is (almost) zero: the exactness guard fails -> the sequential-prefix fallback path
                off[r] = -float(reads[r][int(rs.randint(0, lens[r]))]) + float(rs.choice([0.0, 1e-3, 1e-5, -1e-4]))
            v = rs.rand()
            n = lens[r]
            if v < 0.03: reads[r] = np.full(n, int(rs.randint(-100, 2000)), dtype=np.int16)          # constant
            elif v < 0.06: reads[r] = np.where(rs.rand(n) < 0.5, -32768, 32767).astype(np.int16)     # extremes
            elif v < 0.09: reads[r] = np.clip(reads[r].astype(np.int32) * 40 - 9000, -32768, 32767).astype(np.int16)
            elif v < 0.12: reads[r] = (reads[r] - np.int16(-int(off[r]))).astype(np.int16) if abs(off[r]) < 100 else reads[r]
        svb = bool(stats["batches"] & 1)
        sig = [blow5.svb_zd_encode(x) for x in reads] if svb else reads
        counts = [x.size for x in reads] if svb else None
        rna = kind if rs.rand() < 0.8 else 1 - kind
        pore = int(rs.choice([0, 2]))
        cfg = (0, 0, 0)   # (drawn from a stream of its own: tests/soak_replay.py regenerates the main one)
        if rs_cfg.rand() < a.segments:
            seg = int(rs_cfg.choice([1024, 2048, 3072, 8192, 32768]))
            cfg = (seg, int(seg + 1 + rs_cfg.randint(0, 3 * seg)), int(rs_cfg.choice([0, 0, 16, 32])))
        lanes_cfg = int(rs_cfg.choice([0, 0, -1, 1, 2, 4, 8, 16, 32]))   # lanes per short read
        tag = "batch %d (seed %d kind %d rna %d svb %d segments %s lanes %d)" % (stats["batches"], seed, kind, rna, svb, cfg,
                                                                             lanes_cfg)

        t_batch = time.time()
        if a.trace:
            with open(a.trace, "a") as tf:
                tf.write(tag + " lens " + ",".join(str(x) for x in lens[:64]) + "\n")
        job.stage(sig, dig, off, rng, counts)
        api.event_configure(*cfg)
        api.event_configure_short(lanes_cfg)
        job.set_options()
        job.launch(api.TOOL_EVENT, rna=rna)
        res = job.wait()
        t_event = time.time() - t_batch
        api.event_configure(0, 0, 0)
        api.event_configure_short(0)
        job.set_options()
        stats["split_reads"] += int(res["status"].n_split_reads)
        stats["segments"] += int(res["status"].n_segments)
        stats["seam_reruns"] += int(res["status"].n_seam_reruns)
        stats["fallback_reads"] += int(res["status"].n_fallback_reads)
        stats["rerun_chunks"] += int(res["status"].n_rerun_passes)
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = orc.event_raw(raw, dig[r], off[r], rng[r], rna)
            g = res["events"][r]
            if not (g.start.size == e.start.size and np.array_equal(g.start.astype(np.uint64), e.start.astype(np.uint64))
                    and np.array_equal(g.length.astype(np.uint64), e.length.astype(np.uint64))
                    and np.array_equal(g.mean.view(np.uint32), e.mean.view(np.uint32))
                    and np.array_equal(g.stdv.view(np.uint32), e.stdv.view(np.uint32))):
                fail("%s event read %d len %d" % (tag, r, raw.size))
        job.launch(api.TOOL_STAT)
        st = job.wait()["stat"]
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = orc.stat(raw, dig[r], off[r], rng[r])
            ok = int(st[r]["raw_median"]) == e[4]
            for name, ev in (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]), ("pa_median", e[5])):
                a_, b_ = np.float32(st[r][name]), np.float32(ev)
                ok &= bool(a_.view(np.uint32) == b_.view(np.uint32)) or (np.isnan(a_) and np.isnan(b_))
            if not ok:
                fail("%s stat read %d len %d" % (tag, r, raw.size))
        job.launch(api.TOOL_JNN, rna=rna)
        segs = job.wait()["segs"]
        for r, raw in enumerate(reads):
            ex, ey = orc.jnn_raw(raw, rna)
            if not (np.array_equal(segs[r][0].astype(np.int64), ex) and np.array_equal(segs[r][1].astype(np.int64), ey)):
                fail("%s jnn read %d len %d" % (tag, r, raw.size))
        job.launch(api.TOOL_PREFIX, rna=rna, pore=pore)
        pf = job.wait()["prefix"]
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = orc.prefix(raw, dig[r], off[r], rng[r], rna, pore)
            ok = (int(pf[r]["adapt_x"]), int(pf[r]["adapt_y"]), int(pf[r]["polya_x"]), int(pf[r]["polya_y"])) == \
                 (e.adapt_x, e.adapt_y, e.polya_x, e.polya_y)
            if ok and e.adapt_y > 0:
                for name in ("adapt_mean", "adapt_std", "adapt_median"):
                    ok &= bool(np.float32(pf[r][name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32))
            if ok and e.polya_y > 0:
                for name in ("polya_mean", "polya_std", "polya_median"):
                    ok &= bool(np.float32(pf[r][name]).view(np.uint32) == np.float32(getattr(e, name)).view(np.uint32))
            if not ok:
                fail("%s prefix read %d len %d pore %d" % (tag, r, raw.size, pore))
        job.launch(api.TOOL_PA)
        pa = job.wait()["pa"]
        job.launch(api.TOOL_ENT)
        ent = job.wait()["ent"]
        for r, raw in enumerate(reads):
            if not np.array_equal(pa[r].view(np.uint32), orc.pa(raw, dig[r], off[r], rng[r]).view(np.uint32)):
                fail("%s pa read %d" % (tag, r))
            if raw.size and not np.array_equal(ent[r].view(np.uint64), orc.ent(raw).view(np.uint64)):
                fail("%s ent read %d" % (tag, r))
        bits = int(rs.randint(1, 9)); method = int(rs.randint(0, 3)); svb_out = bool(rs.randint(0, 2))
        job.launch_qts(bits, method, svb_out)
        q = job.wait()
        for r, raw in enumerate(reads):
            x = raw.astype(np.int64)
            mask = (1 << bits) - 1
            if method == 0: e = (x >> bits) << bits
            elif method == 2: e = x | mask
            else: e = np.where((x & mask) < (1 << (bits - 1)), x & ~mask, (x & ~mask) + (1 << bits))
            e = e.astype(np.int16)
            good = (q["blobs"][r] == blow5.svb_zd_encode(e)) if svb_out else np.array_equal(q["samples"][r], e)
            if not good:
                fail("%s qts read %d bits %d method %d svb_out %d" % (tag, r, bits, method, svb_out))
        if time.time() - t_batch > 20.0:   # (mostly the oracle on a batch of long reads; the GPU part is printed beside it)
            print("SLOW %s: %.1f s (event job %.2f s), %d reads, %d samples" % (tag, time.time() - t_batch, t_event, nr,
                                                                               int(sum(lens))), flush=True)
        stats["batches"] += 1
        stats["reads"] += nr
        stats["samples"] += int(sum(lens))
    job.close()
    print(json.dumps(stats))
    sys.exit(1 if stats["mismatches"] else 0)


if __name__ == "__main__":
    main()
