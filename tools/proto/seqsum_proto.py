#!/usr/bin/env python3
"""Prototype of the wave-parallel EXACT evaluation of a sequential float32 sum (s = fl(s + x_i), in order),
as used by sigtk_amd/csrc/seqsum.h: numpy emulation of the 64 lanes, checked against the plain loop.

Idea: while the running sum s stays inside one binade [2^E, 2^(E+1)], fl(s + x) depends on s only
through the parity of its significand S (ties to even).  A lane therefore runs its SPL consecutive terms from two
SURROGATE starts inside the same binade, 1.5*2^E (S even) and 1.5*2^E + ulp (S odd), with native float additions; the
difference of the float bit patterns is the lane's increment f_p for either incoming parity p.  The lanes' parity
maps (constant / identity / negation) are composed with a segmented xor scan on 64-bit masks, the chosen increments
are summed, and S + sum <= 2^24 certifies that the sum stayed inside the binade.  Otherwise the first crossing lane
is located by a scan and runs its terms natively from its true start; the remaining lanes repeat with the new binade
(in the HIP code the first of these steps happens in ss_fast, the repeats in ss_finish).
"""
import numpy as np

SPL = 16
W = 64
TILE = SPL * W
f32 = np.float32


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def seq_ref(x, s0=0.0):
    s = f32(s0)
    for v in x:
        s = f32(s + v)
    return s


def scalable(m):
    e = (int(bits(m)) >> 23) & 0xff
    return 27 <= e <= 227 and m > 0


class Stats:
    def __init__(self):
        self.walks = self.crossings = self.serial_tiles = self.composes = self.tiles = 0


def tile_chain(m, x, st, neg_lane=None):
    """m: oriented accumulator (np.float32, >= 0); x: (64, SPL) float32 oriented terms (masked = 0).
    returns the new accumulator."""
    st.tiles += 1
    skip = 0
    while True:
        ok = scalable(m)
        if not ok:
            if m == 0 and not np.any(x[skip:] != 0):
                return m
            break
        b = int(bits(m))
        E = (b >> 23) & 0xff
        S = (b & 0x7fffff) | 0x800000
        u = np.array([(E - 23) << 23], dtype=np.uint32).view(np.float32)[0]
        B0 = np.array([(E << 23) | 0x400000], dtype=np.uint32).view(np.float32)[0]
        B1 = f32(B0 + u)
        assert int(bits(B1)) == int(bits(B0)) + 1
        acc0 = np.full(W, B0, dtype=np.float32)
        acc1 = np.full(W, B1, dtype=np.float32)
        live = np.arange(W) >= skip
        st.walks += 1
        for e in range(SPL):
            t = np.where(live, x[:, e], f32(0))
            acc0 = (acc0 + t).astype(np.float32)
            acc1 = (acc1 + t).astype(np.float32)
        b0, b1 = bits(acc0).astype(np.int64), bits(acc1).astype(np.int64)
        f0 = b0 - int(bits(B0))
        f1 = b1 - int(bits(B1))
        inr = ((b0 ^ int(bits(B0))) | (b1 ^ int(bits(B1)))) < 0x800000
        bad = ~inr
        if neg_lane is not None:
            bad |= neg_lane & live
        if np.any(bad):
            break
        if np.any(f0 != f1):
            st.composes += 1
            O0 = f0 & 1
            O1 = (f1 + 1) & 1
            F = (O0 == O1)
            V = O0.copy()
            # inclusive prefix of the parity maps (doubling), then the incoming parity of each lane
            Vm = sum(int(v) << i for i, v in enumerate(V))
            Fm = sum(int(v) << i for i, v in enumerate(F))
            mask = (1 << 64) - 1
            d = 1
            while d < 64:
                Vm ^= ((Vm << d) & mask) & ~Fm
                Fm |= (Fm << d) & mask
                d *= 2
            p0 = S & 1
            out = Vm ^ ((~Fm & mask) if p0 else 0)
            inn = ((out << 1) | p0) & mask
            par = np.array([(inn >> i) & 1 for i in range(W)])
            f = np.where(par == 1, f1, f0)
        else:
            f = f0
        tot = int(f.sum())
        if S + tot <= (1 << 24):
            Snew = S + tot
            return f32(f32(Snew) * u)
        # crossing
        st.crossings += 1
        excl = np.concatenate([[0], np.cumsum(f)[:-1]])
        Sl = S + excl
        cross = (Sl + f) > (1 << 24)
        ls = int(np.argmax(cross))
        assert cross[ls] and Sl[ls] <= (1 << 24)
        v = f32(f32(Sl[ls]) * u)
        for e in range(SPL):
            v = f32(v + x[ls, e])
        m = v
        skip = ls + 1
        if skip >= W:
            return m
    # serial fallback from lane `skip`
    st.serial_tiles += 1
    for l in range(skip, W):
        for e in range(SPL):
            m = f32(m + x[l, e])
    return m


def seq_sum_wave(x, st=None, head=256, sign_aware=True, neg_possible=True):
    """Exact emulation of `s = 0; for v in x: s = fl(s + v)` for float32 x."""
    st = st or Stats()
    x = np.asarray(x, dtype=np.float32)
    n = x.size
    s = f32(0)
    h = min(head, n)
    for v in x[:h]:
        s = f32(s + v)
    pos = h
    sg = f32(1)
    while pos < n:
        # tiles are aligned to TILE in the kernel; here: start at a TILE boundary relative to 0
        t0 = (pos // TILE) * TILE
        blk = np.zeros(TILE, dtype=np.float32)
        lo, hi = pos, min(n, t0 + TILE)
        blk[lo - t0:hi - t0] = x[lo:hi]
        if s < 0 and sign_aware:
            sg = f32(-1)
        elif s > 0:
            sg = f32(1)
        xo = (blk * sg).astype(np.float32).reshape(W, SPL)
        neg_lane = np.any(xo < 0, axis=1) if neg_possible else None
        m = f32(s * sg)
        m = tile_chain(m, xo, st, neg_lane)
        s = f32(m * sg)
        pos = hi
    return s, st


def main():
    rs = np.random.RandomState(1)
    tot = Stats()
    cases = []
    for n in [0, 1, 5, 63, 64, 65, 100, 1023, 1024, 1025, 5000, 30000, 100000, 250000]:
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.int16)
        off = f32(rs.randint(0, 20)); unit = f32(f32(1402.882324) / f32(8192.0))
        pa = ((raw.astype(np.float32) + off).astype(np.float32) * unit).astype(np.float32)
        cases.append(("raw n=%d" % n, raw.astype(np.float32)))
        cases.append(("pa n=%d" % n, pa))
        cases.append(("-pa n=%d" % n, (-pa).astype(np.float32)))
        if n:
            mraw = f32(seq_ref(raw.astype(np.float32)) / f32(n))
            d = (raw.astype(np.float32) - mraw).astype(np.float32)
            cases.append(("devraw n=%d" % n, (d * d).astype(np.float32)))
            mpa = f32(seq_ref(pa) / f32(n))
            d = (pa - mpa).astype(np.float32)
            cases.append(("devpa n=%d" % n, (d * d).astype(np.float32)))
    # adversarial
    cases.append(("randint16", rs.randint(-32768, 32767, size=20000).astype(np.float32)))
    cases.append(("pm2000", rs.randint(-2000, 2000, size=5000).astype(np.float32)))
    cases.append(("zeros", np.zeros(5000, dtype=np.float32)))
    z = np.zeros(5000, dtype=np.float32); z[3000] = 7; z[4000:] = 1
    cases.append(("zeros then ones", z))
    cases.append(("ones 40M-ish", np.full(300000, 255, dtype=np.float32)))
    cases.append(("odd const", np.full(200000, 333, dtype=np.float32)))
    big = rs.normal(100, 10, size=50000).astype(np.float32); big[20000] = 3e7; big[30000] = 1e-30
    cases.append(("outliers", big))
    cases.append(("tiny", (rs.rand(5000) * 1e-35).astype(np.float32)))
    cases.append(("huge", (rs.rand(5000) * 1e35).astype(np.float32)))
    nn = rs.normal(100, 10, size=5000).astype(np.float32); nn[2500] = np.nan
    cases.append(("nan", nn))
    ii = rs.normal(100, 10, size=5000).astype(np.float32); ii[2500] = np.inf; ii[3000] = -np.inf
    cases.append(("inf", ii))
    half = np.full(100000, 0.5, dtype=np.float32); half[::3] = 1.5
    cases.append(("halves", half))
    p2 = np.full(70000, 256.0, dtype=np.float32)
    cases.append(("pow2", p2))
    bad = 0
    for name, x in cases:
        st = Stats()
        got, st = seq_sum_wave(x, st)
        exp = seq_ref(x)
        okk = (int(bits(got)) == int(bits(exp))) or (np.isnan(got) and np.isnan(exp))
        print("%-22s %s got %r exp %r tiles %d walks %d cross %d compose %d serial %d" % (
            name, "ok " if okk else "BAD", got, exp, st.tiles, st.walks, st.crossings, st.composes, st.serial_tiles))
        bad += not okk
    print("FAILED" if bad else "ALL OK")


if __name__ == "__main__":
    main()
