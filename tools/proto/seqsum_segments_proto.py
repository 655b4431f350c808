#!/usr/bin/env python3
"""Prototype (CPU, numpy): a sequential float32 sum over a LONG read cut into segments that are summed independently
and composed afterwards -- the piece `stat` / `jnn` / `prefix` lack to put a long read on several wavefronts
(the event path does it with speculative detector states, 3.1 "Long reads").

s = fl(s + x_i), i in order, all x_i >= 0.  While s stays inside one binade [2^E, 2^(E+1)) with unit u = 2^(E-23), the
increment of every addition depends on s only through the PARITY of its significand S = s/u (round to nearest even
consults it on ties, nothing else) -- sigtk_amd/csrc/seqsum.h uses that per 16 terms of a lane; here it is used per
segment of a read:

  summary(E, x[seg])  ->  (d0, d1): the segment's total increment in units of u when S enters even / odd,
                          obtained by running the segment from two surrogate starts inside the binade
                          (1.5 * 2^E and 1.5 * 2^E + u); valid while both surrogates stay inside the binade,
                          i.e. while the segment's sum is below a quarter of the accumulator.

  compose: s_out = (S_in + d[S_in & 1]) * u, valid if S_in + d <= 2^24 (the true sum did not leave the binade either).

A segment's summary needs the binade E of the accumulator at its start.  The accumulator differs from the exact sum
of the terms in front by at most n * u / 2, so E is known from a (parallel) prefix of exact sums except in a narrow
band under a power of two; there, and for segments in which the accumulator crosses a binade (at most one per segment
once the accumulator is >= 4 x a segment's sum), and for the first segments of a read (accumulator comparable to a
segment's sum), the segment is summed from its true start: the serial part of a long read is 1 + ~log2(G) of its G
segments.

compose_read() below does exactly that on the CPU and counts how many segments took which route; tests/
test_seqsum_segments_model.py compares it with the plain loop bit for bit."""
import numpy as np

f32 = np.float32


def bits(x):
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def from_bits(b):
    return np.array([b], dtype=np.uint32).view(np.float32)[0]


def seq(x, s0=f32(0)):
    s = f32(s0)
    for v in x:
        s = f32(s + v)
    return s


def summary(E, x):
    """increments (in units of 2^(E-150)... of the binade's ulp) of the segment for an even / odd entering significand,
    or None if a surrogate left the binade"""
    b0 = (E << 23) | 0x400000
    out = []
    for p in (0, 1):
        start = from_bits(b0 + p)
        end = seq(x, start)
        eb = bits(end)
        if (eb >> 23) != E:        # left the binade (or inf / nan): no summary
            return None
        out.append(eb - (b0 + p))
    return tuple(out)


def compose_read(x, seg_len, stats=None):
    """the sequential float32 sum of x (>= 0), segment by segment; returns the float32 result"""
    x = np.asarray(x, dtype=np.float32)
    n = x.size
    G = (n + seg_len - 1) // seg_len
    # what the segments can do without knowing the accumulator: exact sums of the terms in front (float64 of float32
    # terms: exact for these sizes), hence the binade the accumulator will be in -- up to its rounding error
    exact_before = np.concatenate([[0.0], np.cumsum(x.astype(np.float64))])[::seg_len][:G]
    s = f32(0)
    for g in range(G):
        xs = x[g * seg_len:(g + 1) * seg_len]
        route = "serial"
        sb = bits(s)
        E = (sb >> 23) & 0xff
        if 27 <= E <= 227 and s > 0:
            # the segment's own guess of E (from the exact prefix) must agree with the true one to have been useful
            guess = (bits(f32(exact_before[g])) >> 23) & 0xff if exact_before[g] > 0 else -1
            d = summary(E, xs) if guess == E else None
            if d is not None:
                S = (sb & 0x7fffff) | 0x800000
                tot = S + d[S & 1]
                if tot <= (1 << 24):
                    u = from_bits((E - 23) << 23)
                    s = f32(f32(tot) * u) if tot < (1 << 24) else from_bits((E + 1) << 23)
                    route = "composed"
        if route == "serial":
            s = seq(xs, s)
        if stats is not None:
            stats[route] = stats.get(route, 0) + 1
    return s


def main():
    rs = np.random.RandomState(1)
    unit = f32(f32(1402.882324) / f32(8192.0))
    for n, seg in ((1600000, 131072), (3000001, 131072), (500000, 16384)):
        raw = np.clip(np.rint(rs.normal(520, 75, size=n)), 0, 4000).astype(np.float32)
        pa = ((raw + f32(7)).astype(np.float32) * unit).astype(np.float32)
        for name, v in (("raw", raw), ("pa", pa)):
            st = {}
            got = compose_read(v, seg, st)
            ref = seq(v)
            print(n, seg, name, "composed" if bits(got) == bits(ref) else "MISMATCH", st)


if __name__ == "__main__":
    main()
