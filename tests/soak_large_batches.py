#!/usr/bin/env python3
"""Randomised differential test of what sgk_stat_options_t::kernels = 0 picks for LARGE batches (stat: one read per lane
with the median out of k_moments' second pass; prefix: the wave finders + the lane kernels for the region statistics)
against the wave-per-read kernels alone, on device-resident batches of 82 000 - 130 000 reads with the hostile reads
of tests/test_gpu_stat.py::test_large_batch_choices_match_the_wave_kernels.  Every record must be identical.
    python tests/soak_large_batches.py [--minutes 3] [--seed 1]"""
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
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    from sigtk_amd import api, device
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(a.seed)
    t_end = time.time() + a.minutes * 60
    stats = {"batches": 0, "reads": 0, "samples": 0, "flag_prone_reads": 0, "mismatches": []}
    while time.time() < t_end and len(stats["mismatches"]) < 5:
        kind = int(rs.randint(0, 2))
        n = int(rs.randint(82000, 130000))
        mean = int(rs.choice([600, 2500, 4000, 9000]))
        lens = np.clip(rs.normal(mean, mean * rs.uniform(0.02, 0.08), size=n), 0, None).astype(np.int64)
        lens[rs.randint(0, n, size=200)] = rs.randint(0, 12, size=200)
        seed = int(rs.randint(0, 1 << 30))
        b = device.synth_reads(n, 0, seed=seed, kind=kind, device=dev, lengths=lens)
        rng = b.rng.cpu().numpy().copy()
        rng[rs.randint(0, n, size=n // 40)] *= -1.0
        b.rng.copy_(torch.from_numpy(rng).to(dev))
        host = b.samples.cpu().numpy().copy()
        hostile = rs.randint(0, n, size=n // 100)
        for r in hostile:
            o, m = int(b.offsets_host[r]), int(lens[r])
            if m == 0:
                continue
            u = rs.rand()
            if u < 0.2: host[o:o + m] = rs.randint(-100, 2000)
            elif u < 0.4: host[o:o + m] = np.where(rs.rand(m) < 0.5, 300, 900)
            elif u < 0.6: host[o:o + m] = rs.randint(-2000, 2000, size=m)
            elif u < 0.8: host[o:o + m] = np.where(rs.rand(m) < 0.5, 500, 500 + rs.randint(1, 40))
            else: host[o:o + m] = np.clip(host[o:o + m].astype(np.int32) * 40 - 9000, -32768, 32767).astype(np.int16)
        b.samples.copy_(torch.from_numpy(host).to(dev))
        stats["flag_prone_reads"] += int(hostile.size)
        pore = int(rs.choice([0, 2]))
        out = {}
        for kernels in (0, 2):
            api.stat_configure(kernels)
            try:
                st = device.stat(b).cpu().numpy().copy()
                pf = device.prefix(b, kind, pore).cpu().numpy().copy()
            finally:
                api.stat_configure(0)
            out[kernels] = (st, pf)
        tag = "batch %d (seed %d kind %d pore %d, %d reads of ~%d)" % (stats["batches"], seed, kind, pore, n, mean)
        plan = api.stat_plan("stat", n, int(lens.sum()), int(lens.max()))
        stats["batches_on_lane_kernels"] = stats.get("batches_on_lane_kernels", 0) + (1 if plan.kernels == 1 else 0)
        for what, i, dt in (("stat", 0, api.STAT_DTYPE), ("prefix", 1, api.PREFIX_DTYPE)):
            x = np.frombuffer(out[0][i].tobytes(), dtype=dt)[:n]
            y = np.frombuffer(out[2][i].tobytes(), dtype=dt)[:n]
            bad = [r for r in range(n) if x[r].tobytes() != y[r].tobytes()] if x.tobytes() != y.tobytes() else []
            for r in bad[:3]:
                stats["mismatches"].append("%s %s read %d len %d: default %r wave %r" % (tag, what, r, lens[r], x[r], y[r]))
                print("MISMATCH", stats["mismatches"][-1], flush=True)
        # the ORACLE on 50 reads of the batch, the hostile ones first (the default's records at the size the choice is made at)
        from oracle.oracle import Oracle
        orc = stats.setdefault("_oracle", Oracle())
        got = np.frombuffer(out[0][0].tobytes(), dtype=api.STAT_DTYPE)[:n]
        dig_h, off_h = b.dig.cpu().numpy(), b.off.cpu().numpy()
        for r in [int(x) for x in hostile[:30]] + [int(x) for x in rs.randint(0, n, size=20)]:
            m = int(lens[r])
            if m == 0:
                continue
            o = int(b.offsets_host[r])
            e = orc.stat(host[o:o + m], dig_h[r], off_h[r], rng[r])
            g = got[r]
            ok = int(g["raw_median"]) == e[4] and all(np.float32(g[k]).view(np.uint32) == np.float32(v).view(np.uint32) for k, v in
                                                      (("raw_mean", e[0]), ("pa_mean", e[1]), ("raw_std", e[2]), ("pa_std", e[3]), ("pa_median", e[5])))
            stats["oracle_checked"] = stats.get("oracle_checked", 0) + 1
            if not ok:
                stats["mismatches"].append("%s stat read %d len %d against the oracle: %r vs %r" % (tag, r, m, g, e))
                print("MISMATCH", stats["mismatches"][-1], flush=True)
        stats["batches"] += 1
        stats["reads"] += n
        stats["samples"] += int(lens.sum())
    stats.pop("_oracle", None)
    print(json.dumps(stats))
    sys.exit(1 if stats["mismatches"] else 0)


if __name__ == "__main__":
    main()
