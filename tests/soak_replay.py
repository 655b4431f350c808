#!/usr/bin/env python3
"""Regenerate one batch of tests/soak_parity.py (same --seed, batch index) without running the others: the random
stream of the soak does not depend on any result, so it can be replayed on the CPU.  Writes an .npz with the reads.
    python tests/soak_replay.py --seed 77 --batch 6919 --out /tmp/b.npz        (CPU)
    python tests/soak_replay.py --check /tmp/b.npz                            (GPU: event parity of that batch)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen(seed0, target, last=None):
    """yield the batches target..last (one batch if last is None)"""
    last = target if last is None else last
    from sigtk_amd import api
    rs = np.random.RandomState(seed0)
    b = 0
    while True:
        kind = int(rs.randint(0, 2))
        if rs.rand() < 0.1:
            nr = int(rs.randint(200, 2000))
            lens = [int(x) for x in rs.randint(0, 500, size=nr)]
        else:
            nr = int(rs.randint(1, 24))
            lens = [int(x) for x in np.exp(rs.uniform(np.log(1), np.log(200000), size=nr)).astype(np.int64)]
        seed = int(rs.randint(0, 1 << 30))
        real = target <= b <= last
        if real:
            reads, dig, off, rng = api.synth_reads_host(nr, lens, seed, kind)
            dig = np.asarray(dig, dtype=np.float64).copy(); off = np.asarray(off, dtype=np.float64).copy()
            rng = np.asarray(rng, dtype=np.float64).copy()
        else:
            reads = [np.zeros(n, dtype=np.int16) for n in lens]
            dig = np.full(nr, 8192.0); off = np.zeros(nr); rng = np.full(nr, 1400.0)
        for r in range(nr):
            u = rs.rand()
            if u < 0.15: rng[r] = -rng[r]
            if u > 0.85: off[r] = off[r] + rs.uniform(-3, 3)
            if 0.4 < u < 0.45: dig[r] = 2048.0
            if 0.5 < u < 0.6 and lens[r] > 0:
                off[r] = -float(reads[r][int(rs.randint(0, lens[r]))]) + float(rs.choice([0.0, 1e-3, 1e-5, -1e-4]))
            v = rs.rand()
            n = lens[r]
            if v < 0.03: reads[r] = np.full(n, int(rs.randint(-100, 2000)), dtype=np.int16)
            elif v < 0.06: reads[r] = np.where(rs.rand(n) < 0.5, -32768, 32767).astype(np.int16)
            elif v < 0.09: reads[r] = np.clip(reads[r].astype(np.int32) * 40 - 9000, -32768, 32767).astype(np.int16)
            elif v < 0.12: reads[r] = (reads[r] - np.int16(-int(off[r]))).astype(np.int16) if abs(off[r]) < 100 else reads[r]
        svb = bool(b & 1)
        rna = kind if rs.rand() < 0.8 else 1 - kind
        pore = int(rs.choice([0, 2]))
        if real:
            yield dict(reads=reads, dig=dig, off=off, rng=rng, kind=kind, rna=rna, svb=svb, pore=pore, seed=seed, index=b)
        bits = int(rs.randint(1, 9)); method = int(rs.randint(0, 3)); svb_out = bool(rs.randint(0, 2))
        b += 1
        if b > last:
            return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--check", default="")
    ap.add_argument("--seq", type=int, default=0, help="GPU: run the N batches ending at --batch through ONE recycled job "
                    "(event only), as the soak does")
    a = ap.parse_args()
    if a.out:
        d = next(gen(a.seed, a.batch))
        print("batch %d: seed %d kind %d rna %d svb %d, %d reads" % (a.batch, d["seed"], d["kind"], d["rna"], d["svb"], len(d["reads"])))
        lens = np.array([len(x) for x in d["reads"]], dtype=np.int64)
        np.savez_compressed(a.out, samples=np.concatenate(d["reads"]) if len(lens) else np.zeros(0, np.int16), lens=lens,
                            dig=d["dig"], off=d["off"], rng=d["rng"], rna=d["rna"], svb=int(d["svb"]))
        return
    if a.seq:
        import torch
        torch.cuda.init()
        from sigtk_amd import api, blow5
        from oracle.oracle import Oracle
        api.load_library()
        orc = Oracle()
        job = api.Job(0)
        bad = 0
        for d in gen(a.seed, max(a.batch - a.seq + 1, 0), a.batch):
            reads = d["reads"]
            sig = [blow5.svb_zd_encode(x) for x in reads] if d["svb"] else reads
            counts = [x.size for x in reads] if d["svb"] else None
            job.stage(sig, d["dig"], d["off"], d["rng"], counts)
            job.launch(api.TOOL_EVENT, rna=d["rna"])
            res = job.wait()
            for r, raw in enumerate(reads):
                if raw.size == 0:
                    continue
                e = orc.event_raw(raw, d["dig"][r], d["off"][r], d["rng"][r], d["rna"])
                g = res["events"][r]
                if not (g.start.size == e.start.size and np.array_equal(g.start.astype(np.uint64), e.start.astype(np.uint64))
                        and np.array_equal(g.mean.view(np.uint32), e.mean.view(np.uint32))
                        and np.array_equal(g.stdv.view(np.uint32), e.stdv.view(np.uint32))):
                    bad += 1
                    k = 0
                    while k < min(g.start.size, e.start.size) and g.start[k] == e.start[k]:
                        k += 1
                    print("MISMATCH batch", d["index"], "read", r, "len", raw.size, "rna", d["rna"], "events", g.start.size, e.start.size,
                          "first differing event", k, "gpu", g.start[k:k + 4], g.length[k:k + 4], g.mean[k:k + 2],
                          "oracle", e.start[k:k + 4], e.length[k:k + 4], e.mean[k:k + 2], flush=True)
            for tool, kw in ((api.TOOL_STAT, {}), (api.TOOL_JNN, {"rna": d["rna"]}), (api.TOOL_PREFIX, {"rna": d["rna"], "pore": d["pore"]}),
                             (api.TOOL_PA, {}), (api.TOOL_ENT, {})):
                job.launch(tool, **kw)
                job.wait()
        print("bad", bad)
        return
    z = np.load(a.check)
    lens = z["lens"]; o = np.concatenate([[0], np.cumsum(lens)])
    reads = [z["samples"][o[i]:o[i + 1]].copy() for i in range(len(lens))]
    dig, off, rng, rna, svb = z["dig"], z["off"], z["rng"], int(z["rna"]), bool(z["svb"])
    import torch
    torch.cuda.init()
    from sigtk_amd import api, blow5
    from oracle.oracle import Oracle
    api.load_library()
    orc = Oracle()
    bad = 0
    for use_svb in (svb, not svb):
        job = api.Job(0)
        sig = [blow5.svb_zd_encode(x) for x in reads] if use_svb else reads
        counts = [x.size for x in reads] if use_svb else None
        job.stage(sig, dig, off, rng, counts)
        job.launch(api.TOOL_EVENT, rna=rna)
        res = job.wait()
        print("svb", use_svb, "fallback", res["status"].n_fallback_reads, "rerun", res["status"].n_rerun_passes)
        for r, raw in enumerate(reads):
            if raw.size == 0:
                continue
            e = orc.event_raw(raw, dig[r], off[r], rng[r], rna)
            g = res["events"][r]
            same_b = g.start.size == e.start.size and np.array_equal(g.start.astype(np.uint64), e.start.astype(np.uint64))
            same_v = same_b and np.array_equal(g.mean.view(np.uint32), e.mean.view(np.uint32)) and \
                np.array_equal(g.stdv.view(np.uint32), e.stdv.view(np.uint32))
            if not same_v:
                bad += 1
                print("MISMATCH read", r, "len", raw.size, "dig/off/rng", dig[r], off[r], rng[r], "events", g.start.size, e.start.size)
                if same_b:
                    w = np.nonzero((g.mean.view(np.uint32) != e.mean.view(np.uint32)) | (g.stdv.view(np.uint32) != e.stdv.view(np.uint32)))[0]
                    print("  values differ at events", w[:10], "gpu", g.mean[w[:4]], g.stdv[w[:4]], "oracle", e.mean[w[:4]], e.stdv[w[:4]],
                          "start/len", e.start[w[:4]], e.length[w[:4]])
                else:
                    k = 0
                    while k < min(g.start.size, e.start.size) and g.start[k] == e.start[k]:
                        k += 1
                    print("  boundaries differ from event", k, "gpu", g.start[k:k + 6], "oracle", e.start[k:k + 6])
        job.close()
    print("bad", bad)


if __name__ == "__main__":
    main()
