#!/usr/bin/env python3
"""Prototype (CPU, plain Python): the speculative chunk scheme of the event detector as an executable specification,
checked against the sequential automaton (oracle.peaks = src/events.c:371-443) on the statistics of real and synthetic
reads.  It pins the RULES the kernels follow (sigtk_amd/csrc/event_kernels.hip), not their arithmetic:

  * a read is cut into spans (segments), a span into chunks; chunk c > 0 of a span starts `lead` indices early from the
    FRESH state and is accepted iff the state it reached at its first index equals the state chunk c-1 ended with;
    otherwise it is run again from that state (which may change its own end state: a fixed-point loop);
  * the first chunk of a span g > 0 is speculative too; the seam is checked the same way once all spans are done, a
    failed span is run again from the true state, in order;
  * a chunk STOPS at the end of its range.  A peak still pending there is emitted by whoever continues from the same
    state -- the next chunk, which sees it as a peak in front of its own range ("inherited"); it records the position
    and the bit is set once the chunk's start state is known to be the true one;
  * nothing is emitted behind the read's last index: peaks pending there are dropped, as in the reference.

Both detectors are stepped exactly here (the kernels evaluate the long one lazily and replay it; that is a separate
mechanism with its own checks).  tests/test_detector_chunks_model.py runs this against the oracle."""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def fresh(i0):
    # short: peak_pos, peak_value, valid; long: masked_to, peak_pos, peak_value, valid
    return [-1, FLT_MAX, 0, (0 if i0 <= 0 else -1), -1, FLT_MAX, 0]


def norm(st, i):
    s = list(st)
    if s[3] < i:
        s[3] = -1          # a mask that no longer masks is no state
    return (s[0], np.float32(s[1]).tobytes(), s[2], s[3], s[4], np.float32(s[5]).tobytes(), s[6])


def step(st, i, v1, v2, w1, w2, thr1, thr2, ph):
    """one index of short_long_peak_detector; returns the emitted positions (short, long) or -1"""
    es = el = -1
    sp, sv, sval, lm, lp, lv, lval = st
    if i > 0:
        if sp < 0:
            if v1 < sv:
                sv = v1
            elif v1 - sv > ph:
                sv = v1; sp = i
        else:
            if v1 > sv:
                sv = v1; sp = i
            if sv > thr1:
                lm = sp + w1; lp = -1; lv = FLT_MAX; lval = 0
            if sv - v1 > ph and sv > thr1:
                sval = 1
            if sval and (i - sp) > w1 // 2:
                es = sp; sp = -1; sv = v1; sval = 0
    if not (lm >= i):
        if lp < 0:
            if v2 < lv:
                lv = v2
            elif v2 - lv > ph:
                lv = v2; lp = i
        else:
            if v2 > lv:
                lv = v2; lp = i
            if lv - v2 > ph and lv > thr2:
                lval = 1
            if lval and (i - lp) > w2 // 2:
                el = lp; lp = -1; lv = v2; lval = 0
    st[:] = [sp, sv, sval, lm, lp, lv, lval]
    return es, el


def run_chunk(t1, t2, P, i_begin, s, e, start_state):
    """run [i_begin, e) from start_state (given at i_begin); returns (state at s, state at e, own peaks, inherited)"""
    st = list(start_state)
    at_s = None
    own, inherited = [], []
    for i in range(i_begin, e):
        if i == s:
            at_s = norm(st, i)
        for p in step(st, i, t1[i], t2[i], *P):
            if p < 0:
                continue
            if p >= s:
                own.append(p)
            elif i >= s:
                inherited.append(p)   # emitted inside the range, lies in front of it: the state at s carried it
            # else: emitted during the warm-up: its owner emits it
    if at_s is None:
        at_s = norm(st, s)
    return at_s, norm(st, e), list(st), own, inherited


def detect_span(t1, t2, P, a, b, lanes, lead, true_start, stats):
    """chunks of one span [a, b); true_start: the state at a, or None (speculative first chunk).
    returns (state the first chunk reached at a, state at b as a list, peaks incl. inherited ones of chunks > 0,
             inherited peaks of the first chunk)"""
    K = max(16, 16 * (-(-(b - a) // (16 * lanes))))
    bounds = [(a + c * K, min(a + (c + 1) * K, b)) for c in range(lanes) if a + c * K < b]
    res = []
    for c, (s, e) in enumerate(bounds):
        if c == 0 and true_start is not None:
            r = run_chunk(t1, t2, P, s, s, e, true_start)
        else:
            ib = max(s - lead, 0)
            r = run_chunk(t1, t2, P, ib, s, e, fresh(ib))
        res.append(list(r))
    while True:   # verification / re-run loop
        bad = [c for c in range(1, len(bounds)) if res[c][0] != res[c - 1][1]]
        if not bad:
            break
        stats["rerun"] = stats.get("rerun", 0) + len(bad)
        new = {}
        for c in bad:
            s, e = bounds[c]
            new[c] = list(run_chunk(t1, t2, P, s, s, e, res[c - 1][2]))
            new[c][0] = res[c - 1][1]
        for c in bad:
            res[c] = new[c]
    peaks = []
    for c, r in enumerate(res):
        peaks += r[3]
        if c > 0:
            peaks += r[4]
    return res[0][0], res[-1][1], res[-1][2], peaks, res[0][4]


def detect_read(t1, t2, rna, seg_len, lanes=8, lead=32, stats=None):
    stats = {} if stats is None else stats
    n = len(t1)
    P = (7, 14, np.float32(2.5), np.float32(9.0), np.float32(1.0)) if rna else (3, 6, np.float32(1.4), np.float32(9.0), np.float32(0.2))
    spans = [(a, min(a + seg_len, n)) for a in range(0, max(n, 1), seg_len)]
    out = []
    for g, (a, b) in enumerate(spans):   # phase 1: every span on its own
        out.append(list(detect_span(t1, t2, P, a, b, lanes, lead, fresh(0) if g == 0 else None, stats)))
    for g in range(1, len(spans)):        # phase 2: the seams, in order
        if out[g][0] != out[g - 1][1]:
            stats["seam_rerun"] = stats.get("seam_rerun", 0) + 1
            a, b = spans[g]
            out[g] = list(detect_span(t1, t2, P, a, b, lanes, lead, out[g - 1][2], stats))
    peaks = []
    for g, r in enumerate(out):
        peaks += r[3]
        if g > 0:
            peaks += r[4]            # the first chunk's inherited peaks, now that its start state is the true one
    return sorted(peaks)
