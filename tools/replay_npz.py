#!/usr/bin/env python3
"""development: run the event path on the reads of an .npz written by tests/soak_replay.py --out, with a segment
configuration, and print the per-kernel times (sgk_profile_*) and the status block
    python tools/replay_npz.py --npz b.npz [--seg 2048 --lmin 3287 --lead 0] [--only 8,21]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npz", required=True)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--lmin", type=int, default=0)
    ap.add_argument("--lead", type=int, default=0)
    ap.add_argument("--rna", type=int, default=-1)
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    from sigtk_amd import api
    L = api.load_library()
    z = np.load(a.npz)
    lens = z["lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    reads = [z["samples"][offs[i]:offs[i + 1]].astype(np.int16) for i in range(len(lens))]
    dig, off, rng = z["dig"], z["off"], z["rng"]
    rna = int(z["rna"]) if a.rna < 0 else a.rna
    sel = [int(x) for x in a.only.split(",")] if a.only else list(range(len(reads)))
    reads = [reads[i] for i in sel]; dig = dig[sel]; off = off[sel]; rng = rng[sel]
    api.event_configure(a.seg, a.lmin, a.lead)
    L.sgk_profile_enable(1)
    for rep in range(a.reps):
        L.sgk_profile_reset()
        t0 = time.time()
        got, st = api.event(reads, dig, off, rng, rna)
        wall = time.time() - t0
        prof = api.profile_read()
        print(json.dumps({"reads": sel, "lens": [int(r.size) for r in reads], "wall_s": round(wall, 3), "kernels_ms": prof,
                          "status": {k: int(getattr(st, k)) for k, _ in st._fields_}}), flush=True)


if __name__ == "__main__":
    main()
