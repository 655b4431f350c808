#!/usr/bin/env python3
"""Model (CPU, numpy) of k_long_chains (sigtk_amd/csrc/stat_kernels.hip): a sequential float32 sum over a LONG read whose
1024-term tiles are SUMMARISED independently by many wavefronts and COMPOSED by one.

seqsum_proto.py models what a wave does with the 16 terms of a lane (surrogate starts inside the accumulator's binade,
parity maps, a scan); the same is done here once more with the 1024 terms of a tile:

  level 1 (any order, no dependency between tiles): the accumulator in front of a tile is PREDICTED -- the exact sum of
    the terms in front of the wave's run of tiles, then tile by tile from the wave's own summaries -- and the tile is
    summarised for the binade E of the prediction: T0 / T1, the increment of the accumulator's significand over the
    tile's terms when it enters the tile even / odd (the lanes' parity maps composed for either entering parity).
    Tiles the argument does not cover (a surrogate left the binade, a negative term, the read's first tile with its
    natively added head) are marked.
  level 2 (in order, 64 tiles per step): the tiles' maps are composed like lanes'.  A tile whose predicted binade is not
    the true accumulator's, in which the sum leaves its binade (S + increments > 2^24), or that is marked, is evaluated
    from the TRUE accumulator by seqsum_proto.tile_chain.

compose_read() returns the float32 result and counts tiles composed / evaluated; tests/test_seqsum_tiles_model.py checks
it against the plain loop bit for bit."""
import numpy as np

import seqsum_proto as sq

f32 = np.float32
W, SPL, TILE = sq.W, sq.SPL, sq.TILE
MASK = (1 << 64) - 1


def _parity_in(f0, f1, p0):
    """mask of the lanes entered with an odd significand (ss_parity_in of seqsum.h)"""
    O0, O1 = f0 & 1, (f1 + 1) & 1
    Vm = sum(int(v) << i for i, v in enumerate(O0))
    Fm = sum(int(a == b) << i for i, (a, b) in enumerate(zip(O0, O1)))
    d = 1
    while d < 64:
        Vm ^= ((Vm << d) & MASK) & ~Fm
        Fm |= (Fm << d) & MASK
        d *= 2
    out = Vm ^ ((~Fm & MASK) if p0 else 0)
    return ((out << 1) | p0) & MASK


def _pick(f0, f1, p0):
    inn = _parity_in(f0, f1, p0)
    par = np.array([(inn >> i) & 1 for i in range(len(f0))])
    return np.where(par == 1, f1, f0)


def summary(x, mt, force_mark=False):
    """lc_summary: x (64, 16) oriented terms of a tile, mt the predicted accumulator.
    -> ((T0, T1, E) or None when marked, about the sum of the tile's terms)"""
    mb = int(sq.bits(f32(mt)))
    E = (mb >> 23) & 0xff
    ok = not (mb >> 31) and 27 <= E <= 227 and not force_mark
    if ok:
        b0 = (E << 23) | 0x400000
        a0 = np.full(W, np.array([b0], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
        a1 = np.full(W, np.array([b0 + 1], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
        with np.errstate(over="ignore", invalid="ignore"):
            for e in range(SPL):
                a0 = (a0 + x[:, e]).astype(np.float32)
                a1 = (a1 + x[:, e]).astype(np.float32)
        c0, c1 = sq.bits(a0).astype(np.int64), sq.bits(a1).astype(np.int64)
        bad = (((c0 ^ b0) | (c1 ^ (b0 + 1))) >> 23) != 0
        bad |= np.any(np.signbit(x), axis=1)
        if not np.any(bad):
            f0, f1 = c0 - b0, c1 - (b0 + 1)
            if np.any(f0 != f1):
                T0, T1 = int(_pick(f0, f1, 0).sum()), int(_pick(f0, f1, 1).sum())
            else:
                T0 = T1 = int(f0.sum())
            u = float(np.array([(E - 23) << 23], dtype=np.uint32).view(np.float32)[0])
            return (T0, T1, E), T0 * u
    with np.errstate(over="ignore", invalid="ignore"):
        return None, float(np.sum(x.astype(np.float64)))


def compose(m, recs, evaluate, stats):
    """lc_compose: the accumulator m through the tiles of recs (64 per step); evaluate(t, m) runs tile t from the true m"""
    nt = len(recs)
    for g0 in range(0, nt, 64):
        gn = min(64, nt - g0)
        skip = 0
        while skip < gn:
            mb = int(sq.bits(m))
            ex = (mb >> 23) & 0xff
            okl = [recs[g0 + l] is not None and not (mb >> 31) and recs[g0 + l][2] == ex for l in range(gn)]
            fb = next((l for l in range(skip, gn) if not okl[l]), gn)
            fail = fb
            if fb > skip:
                S = (mb & 0x7fffff) | 0x800000
                f0 = np.array([recs[g0 + l][0] for l in range(skip, fb)], dtype=np.int64)
                f1 = np.array([recs[g0 + l][1] for l in range(skip, fb)], dtype=np.int64)
                f = _pick(f0, f1, S & 1) if np.any(f0 != f1) else f0
                incl = np.cumsum(f)
                cross = np.nonzero(S + incl > (1 << 24))[0]
                if cross.size:
                    fail = skip + int(cross[0])
                if fail > skip:
                    u = np.array([(ex - 23) << 23], dtype=np.uint32).view(np.float32)[0]
                    m = f32(f32(S + int(incl[fail - skip - 1])) * u)
                    stats["composed"] = stats.get("composed", 0) + (fail - skip)
            if fail >= gn:
                break
            m = evaluate(g0 + fail, m)
            stats["evaluated"] = stats.get("evaluated", 0) + 1
            skip = fail + 1
    return m


def compose_read(x, waves=64, head=256, stats=None):
    """the sequential float32 sum of x as k_long_chains evaluates it; returns the float32 result"""
    stats = {} if stats is None else stats
    x = np.asarray(x, dtype=np.float32)
    n = x.size
    nt = (n + TILE - 1) // TILE
    pad = np.zeros(nt * TILE, dtype=np.float32)
    pad[:n] = x
    tiles = pad.reshape(nt, W, SPL)
    # pass A: the sum of the terms of each wave's run of tiles; the read is oriented by the sign of its total
    per = (nt + waves - 1) // waves
    with np.errstate(over="ignore", invalid="ignore"):
        seg = [float(np.sum(tiles[w * per:(w + 1) * per].astype(np.float64))) for w in range(waves)]
    sgn = f32(-1) if sum(seg) < 0 else f32(1)
    tiles = (tiles * sgn).astype(np.float32)
    # pass B: the summaries, every wave predicting from the terms in front of its run
    recs = [None] * nt
    for w in range(waves):
        mt = float(sgn) * sum(seg[:w])
        for t in range(w * per, min(nt, (w + 1) * per)):
            recs[t], ts = summary(tiles[t], mt, force_mark=(t == 0))
            mt += ts

    def evaluate(t, m):  # a tile from the true accumulator: the wave kernels' own routine (head natively in tile 0)
        xt = tiles[t].copy()
        if t == 0:
            flat = xt.reshape(-1)
            for v in flat[:min(head, n)]:
                m = f32(m + v)
            flat[:min(head, n)] = 0
        neg = np.any(np.signbit(xt), axis=1)
        return sq.tile_chain(m, xt, sq.Stats(), neg)

    m = compose(f32(0), recs, evaluate, stats)
    return f32(0) if m == 0 else f32(m * sgn)


def main():
    rs = np.random.RandomState(1)
    unit = f32(f32(1402.882324) / f32(8192.0))
    for n in (300000, 1000001):
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.float32)
        pa = ((raw + f32(7)).astype(np.float32) * unit).astype(np.float32)
        for name, v in (("raw", raw), ("pa", pa)):
            st = {}
            got = compose_read(v, stats=st)
            ref = sq.seq_ref(v)
            print(n, name, "equal" if int(sq.bits(got)) == int(sq.bits(ref)) else "MISMATCH", st)


if __name__ == "__main__":
    main()
