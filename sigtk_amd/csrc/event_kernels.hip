// event_kernels.hip -- the `event` hot path (reference: src/events.c:293-573) for gfx950.
//
// The unit of work is a SPAN of a read on one wavefront (64 lanes):
//
//   detector         (detect_span, LazyPass)  window sums -> t-statistics -> short/long peak detector.
//                    Lane c owns chunk c of the span (K samples, K a multiple of 16).  Each lane slides a
//                    running double prefix sum through a register ring, evaluates the reference's mixed
//                    float/double t-statistic expression tree (events.c:338-361) for the short window in
//                    certified fast arithmetic, steps the short detector automaton (events.c:383-440) as
//                    lane-mask algebra, and runs the long detector lazily (exact only where a rigorous bound
//                    cannot exclude a peak).  The automaton is serial in the reference; here every chunk
//                    starts SPECULATIVELY from the fresh state `lead` samples before its chunk, and the
//                    speculation is verified: chunk c is accepted iff its state at its chunk start equals
//                    chunk c-1's state at that position; mismatching chunks are re-run from the true state
//                    until a fixed point (exact in the general case; re-runs are counted in the status
//                    block).  Output: one bit per sample (peak positions) in a workspace bitmap.
//
//   builder          (build_read)  bitmap + samples -> event table (events.c:457-504).  Lane-local double prefix
//                    sums, wave scan across lanes, boundary records compacted in LDS, then one event per lane per
//                    round with one 16-byte store of (start, length, mean, stdv).
//
// and the kernels differ in what a wave's span is:
//
//   k_event          a whole read: detector and builder back to back in the same wave (most reads);
//                    its first workgroups run the detector over the SEGMENTS of reads too long for one wave
//   chain_segment    (inside k_event) reads that several waves share -- long reads, the tail split: the wave of a
//                    segment checks the seam against the segment in front and builds its own events (round 4)
//   k_event_multi    several short reads per wave, `lanes` lanes each, on a side stream beside k_event (round 3)
//   k_event_fallback persistent kernel over the reads that fail the exactness guard: lane 0 reproduces
//                    compute_sum_sumsq's sequential double prefix scan (events.c:293-303) into workspace
//                    scratch, then the same detector and builder run with window/event sums taken as
//                    differences of those arrays, exactly as the reference does.
//
// Exactness guard: the reference accumulates double prefix sums sequentially and uses their differences; the
// fast path forms window sums and event sums directly.  Both give the real-number sums (hence identical bits)
// whenever no prefix sum can round: every sample is a multiple of 2^g (g = lowest bit of the smallest non-zero
// |x|) and all partial sums are below 2^(g+53).  Per read we check  ilogb(n*max|x|) - ilogb(min|x|!=0) <= 29
// for x and for the float squares, and that every non-zero |x| lies in [2^-20, 2^20] (the range in which the
// certified fast arithmetic of tstat_math.h has no subnormal intermediate); reads failing the check take the
// fallback kernel.
#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>

#include "event_args.h"
#include "side_stream.h"
#include "stat_args.h"
#include "sgk_common.h"
#include "tstat_math.h"

namespace sgk {

#ifndef SGK_LEAD_RNA
#define SGK_LEAD_RNA 256
#endif
constexpr int LEAD = 64;   // speculative warm-up (samples) of the generic pass; multiple of 64
#ifndef SGK_LEAD_DNA
#define SGK_LEAD_DNA 64
#endif
#ifndef SGK_LEAD_DNA_SHORT
#define SGK_LEAD_DNA_SHORT 32
#endif
#ifndef SGK_LEAD_RNA_SHORT
#define SGK_LEAD_RNA_SHORT 128
#endif

// chunk length of the generic pass: 64 lanes x K samples cover the read, K a multiple of 64 (its bitmap words are
// 64-bit)
__device__ inline uint32_t chunk_len(int64_t n) {
    const int64_t k = (n + 4095) / 4096;
    return (uint32_t)(k < 1 ? 64 : 64 * k);
}
// Chunk layout of the fast pass.  Every lane runs T = lead + K indices: lane 0 runs [0, T) from the true initial
// state and owns all of it; lane c >= 1 warms up over [cK, cK + lead) and owns [cK + lead, cK + lead + K).  No lane
// ever runs in front of the read.  K is a multiple of 16 (a lane owns whole 16-bit units of the bitmap), so short
// reads stay on most of the 64 lanes: 5 000 samples with lead 32 are 62 chunks of 80.
__device__ inline int chunk_len_fast(int n, int lead) {
    const int m = n > lead ? n - lead : 1;
    const int k = (m + 1023) / 1024;
    return 16 * (k < 1 ? 1 : k);
}
// ... when `lanes` lanes (a power of two) share the read instead of 64
__device__ inline int chunk_len_lanes(int n, int lead, int lanes) {
    const int m = n > lead ? n - lead : 1;
    const int k = (m + 16 * lanes - 1) / (16 * lanes);
    return 16 * (k < 1 ? 1 : k);
}

// ---------------------------------------------------------------- detector state
struct DetState {
    int sp;      // short peak_pos (-1 none)
    float sv;    // short peak_value
    int svalid;
    int lp;      // long peak_pos
    float lv;
    int lvalid;
    int lmask;   // long masked_to, normalised to -1 when it no longer masks
};
__device__ inline DetState det_fresh(int masked_to) {
    DetState d;
    d.sp = -1; d.sv = FLT_MAX; d.svalid = 0;
    d.lp = -1; d.lv = FLT_MAX; d.lvalid = 0;
    d.lmask = masked_to;
    return d;
}
__device__ inline DetState det_norm(DetState d, int i) {
    if (d.lmask < i) d.lmask = -1;
    return d;
}
__device__ inline bool det_equal(const DetState &a, const DetState &b) {
    return a.sp == b.sp && __float_as_int(a.sv) == __float_as_int(b.sv) && a.svalid == b.svalid &&
           a.lp == b.lp && __float_as_int(a.lv) == __float_as_int(b.lv) && a.lvalid == b.lvalid &&
           a.lmask == b.lmask;
}
__device__ inline DetState det_shfl_up(const DetState &a) {
    DetState r;
    r.sp = __shfl_up(a.sp, 1, 64);
    r.sv = __shfl_up(a.sv, 1, 64);
    r.svalid = __shfl_up(a.svalid, 1, 64);
    r.lp = __shfl_up(a.lp, 1, 64);
    r.lv = __shfl_up(a.lv, 1, 64);
    r.lvalid = __shfl_up(a.lvalid, 1, 64);
    r.lmask = __shfl_up(a.lmask, 1, 64);
    return r;
}

template <int W1>
struct DetParam;
template <>
struct DetParam<3> {  // event_detection_defaults, src/events.c:43-47
    static constexpr float thr1 = 1.4f, thr2 = 9.0f, ph = 0.2f;
};
template <>
struct DetParam<7> {  // event_detection_rna, src/events.c:50-54
    static constexpr float thr1 = 2.5f, thr2 = 9.0f, ph = 1.0f;
};

// One index of short_long_peak_detector (src/events.c:383-440): short first, then long.
// emit_s / emit_l receive the emitted peak position of each detector, or -1.
template <int W1>
__device__ inline void det_step(DetState &d, int i, float v1, float v2, int &emit_s, int &emit_l) {
    constexpr int W2 = 2 * W1;
    constexpr float ph = DetParam<W1>::ph;
    emit_s = -1;
    emit_l = -1;
    // ---- short detector: its masked_to stays 0, so only index 0 is skipped (events.c:387)
    if (i > 0) {
        if (d.sp < 0) {
            if (v1 < d.sv) {
                d.sv = v1;
            } else if (v1 - d.sv > ph) {
                d.sv = v1;
                d.sp = i;
            }
        } else {
            if (v1 > d.sv) {
                d.sv = v1;
                d.sp = i;
            }
            if (d.sv > DetParam<W1>::thr1) {  // dominate the long detector (events.c:414-422)
                d.lmask = d.sp + W1;
                d.lp = -1;
                d.lv = FLT_MAX;
                d.lvalid = 0;
            }
            if (d.sv - v1 > ph && d.sv > DetParam<W1>::thr1) d.svalid = 1;
            if (d.svalid && (i - d.sp) > W1 / 2) {
                emit_s = d.sp;
                d.sp = -1;
                d.sv = v1;
                d.svalid = 0;
            }
        }
    }
    // ---- long detector
    if (!(d.lmask >= i)) {
        if (d.lp < 0) {
            if (v2 < d.lv) {
                d.lv = v2;
            } else if (v2 - d.lv > ph) {
                d.lv = v2;
                d.lp = i;
            }
        } else {
            if (v2 > d.lv) {
                d.lv = v2;
                d.lp = i;
            }
            if (d.lv - v2 > ph && d.lv > DetParam<W1>::thr2) d.lvalid = 1;
            if (d.lvalid && (i - d.lp) > W2 / 2) {
                emit_l = d.lp;
                d.lp = -1;
                d.lv = v2;
                d.lvalid = 0;
            }
        }
    }
}

// ---------------------------------------------------------------- per-read context
template <typename T>
struct ReadCtx {
    const T *base;            // read's first sample
    int64_t n;                // samples in the read
    int64_t lo, hi;           // legal read-relative load range
    Scale sc;
    bool vec_ok;
    unsigned long long *bm;   // bitmap words of this read
    const double *P, *P2;     // fallback prefix arrays (n+1 entries) or null
    uint32_t dev;             // EvArgs::dev (development builds; 0 otherwise)
};

template <typename T>
__device__ inline ReadCtx<T> make_ctx(const EvArgs &a, uint32_t r) {
    ReadCtx<T> rc;
    const uint64_t o0 = a.offsets[r];
    rc.base = reinterpret_cast<const T *>(a.samples) + o0;
    rc.n = (int64_t)a.lengths[r];
    rc.lo = -(int64_t)o0;
    rc.hi = (int64_t)(a.n_alloc - o0);
    if (a.dig) rc.sc = make_scale(a.dig[r], a.off[r], a.rng[r]);
    else { rc.sc.offf = 0.0f; rc.sc.unit = 1.0f; }
    rc.vec_ok = ((reinterpret_cast<uintptr_t>(rc.base) & 15u) == 0);
    rc.bm = a.bitmap + (o0 >> 6) + r;
    rc.P = nullptr;
    rc.P2 = nullptr;
    rc.dev = a.dev;
    return rc;
}

// One pass of the GENERIC detector over the wave's chunks: every window sum is a difference of the reference's
// prefix arrays (rc.P / rc.P2), both detectors step on every index.  Slow (uncoalesced loads of the prefix arrays,
// library division and sqrt); only the fallback kernel uses it, for reads the fast pass cannot take.
//   lead   : samples each lane starts before its chunk start (LEAD: speculative pass, 0: re-run)
//   active : whether this lane runs in this pass
//   st     : state at the pass start (lead == 0 only; the speculative pass starts fresh)
//   at_s   : out, normalised state when the lane reaches its chunk start s (speculative pass)
//   at_e   : out, normalised state when the lane reaches its chunk end e (written only when reached)
template <int W1, typename T>
__device__ __attribute__((noinline)) void detect_pass(const ReadCtx<T> &rc, int lead, bool active, int64_t s,
                                                      int64_t e, uint32_t K, DetState st, DetState &at_s,
                                                      DetState &at_e) {
    constexpr int W2 = 2 * W1;
    const int64_t n = rc.n;
    const int64_t i_begin = s - lead;
    if (!__any(active)) return;
    DetState d = (lead > 0) ? det_fresh(i_begin <= 0 ? 0 : -1) : st;
    // bitmap register window: wcur = word of the current index, wprev = the word before it
    unsigned long long wcur = 0ull, wprev = 0ull;
    const int64_t wlo = s >> 6, whi = (e + 63) >> 6;
    bool done = !active;
    const int main_steps = lead + (int)K;
    const bool t1_ok = n >= 2 * W1, t2_ok = n >= 2 * W2;
    int j = 0;
    for (;; ++j) {
        if (j >= main_steps && !__any(!done)) break;
        const int64_t i = i_begin + j;
        if ((j & 63) == 0 && j > 0 && active) {
            // entering bitmap word (i>>6): retire the word two back
            const int64_t wr = (i >> 6) - 2;
            if (wr >= wlo && wr < whi) rc.bm[wr] = wprev;
            wprev = wcur;
            wcur = 0ull;
        }
        if (active && i >= 0) {
            if (i == s && lead > 0) at_s = det_norm(d, (int)i);
            if (i == e) at_e = det_norm(d, (int)i);
            if (i >= n) done = true;
            if (i >= e) {
                const bool pend = (d.sp >= 0 && d.sp < e) || (d.lp >= 0 && d.lp < e);
                if (!pend) done = true;
            }
            if (!done) {
                float v1 = 0.0f, v2 = 0.0f;
                if (t1_ok && i >= W1 && i <= n - W1) {
                    const double p0 = rc.P[i], q0 = rc.P2[i];
                    v1 = sgk_tstat_ref<W1>(p0 - rc.P[i - W1], q0 - rc.P2[i - W1], rc.P[i + W1] - p0,
                                           rc.P2[i + W1] - q0);
                }
                if (t2_ok && i >= W2 && i <= n - W2) {
                    const double p0 = rc.P[i], q0 = rc.P2[i];
                    v2 = sgk_tstat_ref<W2>(p0 - rc.P[i - W2], q0 - rc.P2[i - W2], rc.P[i + W2] - p0,
                                           rc.P2[i + W2] - q0);
                }
                int es, el;
                det_step<W1>(d, (int)i, v1, v2, es, el);
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const int p = z ? el : es;
                    if (p >= s && p < e) {
                        const int64_t wi = (int64_t)p >> 6, wb = i >> 6;
                        const unsigned long long bit = 1ull << (p & 63);
                        if (wi == wb) wcur |= bit;
                        else if (wi == wb - 1) wprev |= bit;
                        else rc.bm[wi] |= bit;  // older word: already retired, owned by this lane only
                    }
                }
            }
        }
    }
    if (active) {
        // the last processed index is i_begin + j - 1; the register window holds its word and
        // the one before it
        const int64_t wb = (i_begin + (int64_t)j - 1) >> 6;
        if (wb - 1 >= wlo && wb - 1 < whi) rc.bm[wb - 1] = wprev;
        if (wb >= wlo && wb < whi) rc.bm[wb] = wcur;
    }
}


// ================================================================ fast detector pass (round 2: "LazyPass")
// Same semantics as detect_pass, restructured around what the instruction stream costs on gfx950
// (tools/valu_rate.hip, profiles/archive/r02_valu_rate.txt: plain f32 add/mul/fma, logic and int add issue in 2.3 cycles per
// wave64 instruction; everything f64, conversions, v_cmp, v_cndmask, v_max/min and packed f32 take 4.45):
//  * window sums are differences of a RUNNING double prefix sum kept in a register ring (P(i) .. P(i+W2+1)): one
//    conversion and one addition per sample for the sums and for the float squares, one subtraction per window;
//    exact under the read-level guard, like every sum of this path;
//  * the A side of a t-statistic (mean1, sumsq1/w - mean1^2) is what the B side's window sum yields W indices later:
//    it is evaluated once per window position and ringed (SgkARole), not re-derived from sums;
//  * the tail |delta| / sqrt(cv/w) is evaluated in f32 with error-free transformations and certified
//    (sgk_tail_f32); uncertified evaluations (2^-12) are redone with the reference expression;
//  * the SHORT detector (events.c:383-440, k = 0) runs on every index, written as lane-mask algebra: the
//    comparisons produce wave masks (scalar registers), the boolean state (in a peak / valid / strong) lives in
//    masks, only peak_value and peak_pos are selected in vector registers;
//  * the LONG detector (k = 1) is LAZY.  It is reset whenever the short detector sits in a strong peak
//    (events.c:414-422) and can only emit if, since that reset, some t-statistic it saw exceeded thr2.  Per index
//    the kernel proves from cheap f32 estimates that the long window's statistic cannot exceed thr2
//    (sgk_long_cold); a run (reset .. next reset) in which the proof fails is recorded (2e-4 of the indices on
//    nanopore data) and re-played with exact arithmetic after the pass (replay_long_runs);
//  * emitted peaks go to a per-lane 512-position bitmap ring in LDS and leave as whole words.
// Positions inside a pass are BLOCK-relative (the 16-step unrolled block's first index = 0), so every position the
// automaton writes is an inline constant; they are rebased once per block.
template <int W1>
struct LzCfg {
    static constexpr int W2 = 2 * W1;
    static constexpr int R = 16;                    // unroll (multiple of every ring length)
    static constexpr int NP = (W1 == 3) ? 8 : 16;   // prefix ring >= W2 + 2
    static constexpr int NA = (W1 == 3) ? 4 : 8;    // short A-side ring >= W1
    static constexpr int NL = (W1 == 3) ? 8 : 16;   // long side ring >= W2
    static constexpr int H1 = W1 / 2;
};
static_assert(LzCfg<3>::NP >= 8 && LzCfg<7>::NP >= 16, "prefix ring holds P(i) .. P(i+W2+1)");

constexpr int LZ_NONE = -(1 << 29);   // "no mask" / far in the past (block-relative positions drift by -16 per block)
constexpr int LZ_NREC = 8;            // hot long-detector runs a lane can record per pass (more: read -> exact fallback)
constexpr int LZ_RING_WORDS = 16;     // per-lane bitmap ring: 512 positions

// detector state at a block boundary: LzSnapState (event_args.h), what chunks hand over / compare
struct LzSnap {
    LzSnapState init[64];  // state a chunk's accepted run started from (at its chunk start)
    LzSnapState at_e[64];  // state at the chunk end
    LzSnapState st0[64];   // start state handed to a re-run
};
__device__ inline bool lz_equal(const LzSnapState &a, const LzSnapState &b) {
    return a.sp == b.sp && __float_as_int(a.sv) == __float_as_int(b.sv) && a.lm == b.lm && a.r0 == b.r0 &&
           a.bits == b.bits;
}
struct LzLds {
    uint32_t ring[64][LZ_RING_WORDS + 1];  // + one word: the lane's inherited emission (LZ_PRE, below)
    LzSnap snap;
    LzRun runs[64][LZ_NREC];
    int nrec[64];
};

// Exact (reference-expression) t-statistic at index i of a read, window sums formed directly from
// the samples in global memory.  Out of line: only reached when a fast evaluation's certificate
// fails (about 2^-12 of the evaluations) and in the long detector's replay.
template <typename T>
__device__ __attribute__((noinline)) float tstat_exact_at(const T *base, Scale sc, int i, int w) {
    double A = 0.0, A2 = 0.0, B = 0.0, B2 = 0.0;
    for (int k = 0; k < w; ++k) {
        const float xa = to_pa(base[i - w + k], sc);
        const float xb = to_pa(base[i + k], sc);
        A = A + (double)xa;
        A2 = A2 + (double)(xa * xa);
        B = B + (double)xb;
        B2 = B2 + (double)(xb * xb);
    }
    if (w == 3) return sgk_tstat_ref<3>(A, A2, B, B2);
    if (w == 6) return sgk_tstat_ref<6>(A, A2, B, B2);
    if (w == 7) return sgk_tstat_ref<7>(A, A2, B, B2);
    return sgk_tstat_ref<14>(A, A2, B, B2);
}

// 16 consecutive samples starting at an even sample offset, as they sit in memory.
template <typename T>
struct Lead16;
template <>
struct Lead16<int16_t> {
    uint32_t w[8];
    template <int U>
    __device__ __forceinline__ float get(const Scale &sc) const {
        const int v = (U & 1) ? ((int)w[U / 2] >> 16) : (int)(short)(w[U / 2] & 0xffffu);
        const float shifted = (float)v + sc.offf;
        return shifted * sc.unit;
    }
};
template <>
struct Lead16<float> {
    float w[16];
    template <int U>
    __device__ __forceinline__ float get(const Scale &) const { return w[U]; }
};
typedef uint32_t sgk_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Samples in front of a read (the speculative warm-up of its first chunks reaches there) are whatever the caller's
// buffer holds.  No t-statistic that sees them is used, but they pass through the RUNNING prefix sums, and a value
// far larger than the read's own samples (a neighbour scaled with this read's offset/range) would leave a rounding
// residue in those double sums for the rest of the chunk.  Every position before the read therefore takes the
// value of the read's first sample: inside the magnitude range the exactness guard checks.  Rare (first lanes of a
// read, first blocks only), kept out of line.
template <typename T>
__device__ __attribute__((noinline)) Lead16<T> lead_fix_head(Lead16<T> g, int pos, T first) {
    T tmp[16];
    __builtin_memcpy(tmp, g.w, sizeof(tmp));
#pragma unroll
    for (int k = 0; k < 16; ++k) tmp[k] = (pos + k < 0) ? first : tmp[k];
    __builtin_memcpy(g.w, tmp, sizeof(tmp));
    return g;
}

// ---- repair context of a read that failed the exactness guard (fallback kernel only) ----------
// The reference's window sums are differences of its sequentially rounded prefix arrays.  They equal
// the exact sums the fast pass forms EXCEPT where an inexact addition of the sequential scan ("event"
// at sample t: prefix[t+1] != prefix[t] + y_t exactly) lies inside the window, i.e. for the indices
// i in [t-w+1, t+w].  The fast pass therefore runs unchanged on such reads and only those indices
// (plus uncertified evaluations) are re-evaluated from the scratch prefix arrays; for the long window they
// count as "hot" (the run is re-played from the prefix arrays).
constexpr int REP_MAX_EVENTS = 32;
struct RepairCtx {
    const double *P, *P2;   // reference prefix arrays (n+1 entries each)
    const int *ev;          // sorted event positions (LDS)
    int nev;
    bool all_dirty;         // more events than REP_MAX_EVENTS: every index is evaluated from the prefix arrays
};

__device__ __attribute__((noinline)) float tstat_prefix_at(const double *P, const double *P2, int i, int w) {
    const double p0 = P[i], q0 = P2[i];
    const double A = p0 - P[i - w], A2 = q0 - P2[i - w], B = P[i + w] - p0, B2 = P2[i + w] - q0;
    if (w == 3) return sgk_tstat_ref<3>(A, A2, B, B2);
    if (w == 6) return sgk_tstat_ref<6>(A, A2, B, B2);
    if (w == 7) return sgk_tstat_ref<7>(A, A2, B, B2);
    return sgk_tstat_ref<14>(A, A2, B, B2);
}

// marks (as "redo exactly" / "hot") the indices q0..q0+3 that lie within a window length of an event
template <int W1>
__device__ __forceinline__ void repair_mark(const RepairCtx &rep, int &next_t, int q0, unsigned cnt1, unsigned cnt2,
                                            unsigned &bad1, unsigned &bad2) {
    constexpr int W2 = 2 * W1;
    if (rep.all_dirty) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if ((unsigned)(q0 + u - W1) < cnt1) bad1 |= 1u << u;
            if ((unsigned)(q0 + u - W2) < cnt2) bad2 |= 1u << u;
        }
        return;
    }
    if (next_t > q0 + 3 + W2 - 1) return;  // no event can reach this quad (the usual case)
    int nt = 0x7fffffff;
    for (int k = 0; k < rep.nev; ++k) {
        const int t = rep.ev[k];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = q0 + u;
            if ((unsigned)(i - t + W1 - 1) < (unsigned)(2 * W1) && (unsigned)(i - W1) < cnt1) bad1 |= 1u << u;
            if ((unsigned)(i - t + W2 - 1) < (unsigned)(2 * W2) && (unsigned)(i - W2) < cnt2) bad2 |= 1u << u;
        }
        if (t + W2 >= q0 + 4 && t < nt) nt = t;  // may still reach a later quad
    }
    next_t = nt;
}

// out of line (rare): a peak whose bitmap word may have left the lane's ring, or that lies in front of the lane's own
// range [s, e).  A lane stops at e even if a peak is still pending there: the lane behind starts (verified) from the
// same state and emits it -- a peak that lies in front of its own range.  If that happens inside its range (cur >=
// own_lo), the lane INHERITED the peak with its start state and leaves the position in its ring's extra word (LZ_PRE):
// detect_span sets the bit once the lane's start state is known to be the true one.  (At most one per pass: after the
// emission every peak position lies inside the range.)  If it happens during the warm-up, the owner has emitted it.
// (Round 2 let the OWNER run on until its pending peak was emitted; on a flat signal that is never, and every pass of
// every lane walked to the end of the read: 5 s for a constant read of 75 000 samples.)
constexpr int LZ_PRE = LZ_RING_WORDS;
__device__ __attribute__((noinline)) void lz_emit_slow(uint32_t *ring, unsigned long long *bm, int flushed,
                                                       int i_begin, int s, int e, int p, int cur) {
    const int own_lo = s - i_begin;
    if (p < own_lo) {
        if (cur >= own_lo) ring[LZ_PRE] = (uint32_t)(i_begin + p);
        return;
    }
    if (p >= flushed) {
        atomicOr(&ring[(p >> 5) & (LZ_RING_WORDS - 1)], 1u << (p & 31));
    } else {
        const int pa = i_begin + p;
        if (pa >= s && pa < e) atomicOr(reinterpret_cast<uint32_t *>(bm) + (pa >> 5), 1u << (pa & 31));
    }
}
// out of line (rare): a hot long-detector run [a, b) ended at the reset of index b (or at the read's end, b = n); the
// lane whose chunk holds index b (for b = n: index n-1) replays it.  Lane c+1 meets the reset at its first index
// with the state it shares with lane c, so exactly one lane records every run.
__device__ __attribute__((noinline)) int lz_record(LzRun *runs, int nrec, int a, int b, int s, int e, int n) {
    if ((b >= s && b < e) || (b == n && e == n && s < n)) {
        if (nrec < LZ_NREC) {
            runs[nrec].a = a;
            runs[nrec].b = b;
        }
        ++nrec;
    }
    return nrec;
}

// out of line (rare): the uncertified t-statistics of a quad, redone with the reference expression
struct Redo4 {
    float v[4];
};
template <int W1, typename T, bool FLAGGED>
__device__ __attribute__((noinline)) Redo4 redo_quad(const T *base, Scale sc, const double *P, const double *P2, int i0,
                                                     unsigned cnt1, unsigned bits, float t0, float t1, float t2,
                                                     float t3) {
    Redo4 r;
    r.v[0] = t0; r.v[1] = t1; r.v[2] = t2; r.v[3] = t3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((bits >> k) & 1u) {
            const int i = i0 + k;
            float v = 0.0f;
            if ((unsigned)(i - W1) < cnt1) {
                if constexpr (FLAGGED) v = tstat_prefix_at(P, P2, i, W1);
                else v = tstat_exact_at<T>(base, sc, i, W1);
            }
            r.v[k] = v;
        }
    }
    return r;
}

typedef unsigned long long lmask_t;  // one bit per lane: lives in a scalar register pair, combined on the scalar unit
// mask -> per-lane predicate without vector work: selects become v_cndmask with the mask as its condition operand,
// branches become s_and_saveexec
__device__ __forceinline__ bool lane_of(lmask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// State of one fast pass.  Every ring access uses a compile-time index (U is a template parameter and the pass
// starts on a multiple of the unroll length), so the rings live in registers; the 16 steps of one loop iteration
// are expanded with fold expressions, four at a time: t-statistics of 4 indices -> (rare, rolled, out of line)
// exact redo of uncertified ones -> the automaton on those 4 indices.
//
// Samples are NOT staged through LDS: each lane loads the 16 leading samples of the next block (x[i+W2], 32 bytes)
// straight from global memory one block ahead.  A wave touches 64 different 128-byte lines per load instruction;
// each line is consumed over 4 consecutive blocks and stays in L2 meanwhile, so HBM traffic remains one pass over
// the samples and there are no barriers or cooperative loads in the loop.
template <int W1, typename T, bool FLAGGED>
struct LazyPass {
    using C = LzCfg<W1>;
    static constexpr int W2 = C::W2, R = C::R, NP = C::NP, NA = C::NA, NL = C::NL, H1 = C::H1;
    // rings
    double Ps[NP], Pq[NP];        // running prefix sums of x and of fl(x*x); slot of P(k) = k mod NP
    SgkARole ar[NA];              // short A side of window position p at slot p mod NA
    SgkLSide ls[NL];              // long-window estimates of window position p at slot p mod NL
    float t1[4];                  // t-statistics of the current quad
    lmask_t nk[4];                // lanes whose t-statistic of the quad's k-th index is not certified
    lmask_t hc[4];                // lanes whose long window may exceed thr2 at the quad's k-th index
    Lead16<T> cur;                // x[ib + W2 .. ib + W2 + 16)
    // short detector (block-relative positions); boolean state as lane masks
    float sv;
    int sp;
    lmask_t inpk, val, strong;
    lmask_t hist[H1 + 1];           // hist[k]: lanes whose peak_pos was set k+1 indices ago
    uint32_t bw;                  // bitmap word (32 positions) that holds position j - H1 - 1, the usual emitted peak
    // lazy long detector
    int lm, r0;                   // short peak position at the last reset (masked while i <= lm + W1); last reset
    lmask_t hot;
    int nrec;
    // geometry
    uint32_t *ring;               // this lane's bitmap ring in LDS
    LzRun *runs;
    const T *base;
    int lo, hi;                   // legal read-relative load range
    Scale sc;
    int n, s, e, i_begin, ib, jb, flushed;
    unsigned cnt1, cnt2;
    lmask_t done;                 // lanes that have nothing left to do (past their chunk, no pending peak)
    bool slow;                    // wave-uniform: this block takes the predicated steps (read's ends, very old peak)
    bool oldpeak;                 // wave-uniform: some lane's peak may lie outside the bitmap ring
    unsigned long long *bm;       // read's bitmap (global)
    RepairCtx rep;                // FLAGGED only
    int next_t;                   // FLAGGED only
    int dirty;                    // FLAGGED only: block-relative index up to which this lane's own window sums are
                                  // not trusted (an addition of ITS running prefix was inexact, see tstep)

    // Unconditional 32-byte load of x[pos .. pos+16).  Positions outside the readable range are redirected to the
    // nearest readable group: whatever value a position yields is used consistently (it enters the prefix sum once),
    // and no t-statistic whose window reaches outside [0, n) is ever used (events.c:332-338).  Positions before the
    // read are replaced by the read's first sample (lead_fix_head).
    __device__ __forceinline__ void load_lead(Lead16<T> &dst, int pos) const {
        int p = pos > hi - 16 ? hi - 16 : pos;
        p = p < lo ? lo : p;
        constexpr int NV = 16 * (int)sizeof(T) / 16;
        const sgk_u32x4_a4 *src = reinterpret_cast<const sgk_u32x4_a4 *>(base + p);
        sgk_u32x4_a4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = src[k];
        __builtin_memcpy(dst.w, v, sizeof(dst.w));
        if (pos < 0) dst = lead_fix_head<T>(dst, pos, base[0]);
    }

    // phase 1 of index ib+U: advance the prefix ring, form the window sums, the short window's t-statistic and the
    // long window's bound.  No predicates: what a block near the read's ends needs is patched per quad (slow_fix).
    template <int U>
    __device__ __forceinline__ void tstep() {
        const float xn = cur.template get<U>(sc);  // x[i + W2]
        const float xqn = xn * xn;
        // P(i+W2+1) = P(i+W2) + x[i+W2]; it replaces P(i-1)
        Ps[(U + W2 + 1) % NP] = Ps[(U + W2) % NP] + (double)xn;
        Pq[(U + W2 + 1) % NP] = Pq[(U + W2) % NP] + (double)xqn;
        const double p0 = Ps[U % NP], q0 = Pq[U % NP];
        const double b1 = Ps[(U + W1) % NP] - p0, b1q = Pq[(U + W1) % NP] - q0;
        const double b2 = Ps[(U + W2) % NP] - p0, b2q = Pq[(U + W2) % NP] - q0;
        // short window: its A side was ringed W1 indices ago
        bool ok;
        const float v = sgk_tstat_try_ab<W1>(b1, b1q, ar[(U + NA - W1) % NA], ok);
        bool clean = true;
        if constexpr (FLAGGED) {
            ok = ok && sgk_try_domain<W1>(b1, b1q, ar[(U + NA - W1) % NA]);
            // A flagged read failed the magnitude guard: the lane's running prefix can round as well, and not where
            // the reference's sequential scan did (other origin, other magnitudes), so `rep` does not list those
            // places.  TwoSum residual of the two additions above; after an inexact one the window sums of the next
            // 2 * W2 indices (every window with x[i + W2] inside) are taken from the reference's prefix arrays.
            // (Found by the soak: a 2e-5 pA sample 22 indices in front of a plateau of two equal statistics.)
            const double a_s = Ps[(U + W2) % NP], b_s = (double)xn, r_s = Ps[(U + W2 + 1) % NP];
            const double a_q = Pq[(U + W2) % NP], b_q = (double)xqn, r_q = Pq[(U + W2 + 1) % NP];
            const double t_s = r_s - a_s, t_q = r_q - a_q;
            const double e_s = (a_s - (r_s - t_s)) + (b_s - t_s), e_q = (a_q - (r_q - t_q)) + (b_q - t_q);
            clean = U > dirty;
            if (e_s != 0.0 || e_q != 0.0) dirty = dirty > U + 2 * W2 ? dirty : U + 2 * W2;
            ok = ok && clean;
        }
        ar[U % NA] = sgk_arole<W1, !FLAGGED>(b1, b1q);
        // long window: bound only
        const SgkLSide lb = sgk_lside<W2>(b2, b2q);
        bool cold = sgk_long_cold<W2>(ls[(U + NL - W2) % NL], lb);
        if constexpr (FLAGGED) cold = cold && clean && sgk_lside_domain(ls[(U + NL - W2) % NL]) && sgk_lside_domain(lb);
        ls[U % NL] = lb;
        t1[U & 3] = v;
        nk[U & 3] = ~__ballot(ok);
        hc[U & 3] = ~__ballot(cold);
    }

    // One step of the short detector (events.c:383-440, k = 0) and of the lazy long detector's bookkeeping, on lane
    // masks.  u: block-relative index (an inline constant in the fast form); live: lanes that take the step.
    template <bool SLOW>
    __device__ __forceinline__ void dstep_core(const int u, const float v, const lmask_t hck, const lmask_t live) {
        constexpr float ph = DetParam<W1>::ph, thr1 = DetParam<W1>::thr1;
        const float d1 = v - sv;
        const float ee = lane_of(inpk) ? d1 : -d1;   // in a peak: v - peak_value; before one: peak_value - v
        lmask_t P = __ballot(ee > 0.0f);             // v > peak_value (in a peak) / v < peak_value (before one)
        lmask_t Q = __ballot(ee < -ph);              // peak_value - v > ph (in a peak) / v - peak_value > ph
        const lmask_t Tt = __ballot(v > thr1);
        if constexpr (SLOW) {
            P &= live;
            Q &= live;
        }
        const lmask_t ent = Q & ~inpk;                      // a peak starts here: peak_pos = i
        const lmask_t pos = (inpk & P) | ent;               // peak_pos = i
        strong = (pos & Tt) | (strong & ~pos);              // peak_value > threshold
        lmask_t dom = inpk & strong;                        // events.c:414-422: the short detector dominates the long one
        if constexpr (SLOW) dom &= live;
        val = inpk & (val | (Q & strong));
        // (i - peak_pos) > w/2  <=>  peak_pos was not set during the last w/2 indices (nor at this one: ~P)
        lmask_t recent = hist[0];
#pragma unroll
        for (int k = 1; k < H1; ++k) recent |= hist[k];
        lmask_t em = val & ~P & ~recent;
        if constexpr (SLOW) em &= live;
        const lmask_t upd = P | ent | em;
        // Emission.  A strong peak stays strong and in a peak until it is emitted, so the emission step is the LAST
        // step at which this peak resets the long detector: masked_to and the reset index are taken here.
        // the usual emitted peak was set exactly H1+1 indices ago: its position is the same in every lane, and so
        // is its bit in the lane's current bitmap word
        const uint32_t bit = 1u << ((jb + u - H1 - 1) & 31);
        if (SLOW && oldpeak) {
            // some lane holds a peak older than the bitmap ring reaches (or one from before the pass)
            if (lane_of(em)) {
                lz_emit_slow(ring, bm, flushed, i_begin, s, e, jb + sp, jb + u);
                lm = sp;
                r0 = u;
            }
        } else {
            if (lane_of(em)) {
                bw |= bit;
                lm = sp;
                r0 = u;
            }
            const lmask_t erare = em & ~hist[H1];
            if (erare != 0ull) {
                if (lane_of(erare)) {  // an older peak: undo the bit, set the right one (its word is in the ring)
                    bw &= ~bit;
                    const int p = jb + sp, own = s - i_begin;
                    if (p >= own) atomicOr(&ring[(p >> 5) & (LZ_RING_WORDS - 1)], 1u << (p & 31));
                    else if (jb >= own) ring[LZ_PRE] = (uint32_t)(i_begin + p);  // inherited, see lz_emit_slow
                }
            }
        }
        sv = lane_of(upd) ? v : sv;
        sp = lane_of(pos) ? u : sp;
        inpk = (inpk & ~em) | ent;
        val = val & ~em;
        if constexpr (SLOW) {
            // a frozen lane's history does not age
#pragma unroll
            for (int k = H1; k >= 1; --k) hist[k] = (hist[k - 1] & live) | (hist[k] & ~live);
            hist[0] = pos | (hist[0] & ~live);
        } else {
#pragma unroll
            for (int k = H1; k >= 1; --k) hist[k] = hist[k - 1];
            hist[0] = pos;
        }
        // lazy long detector: the run that ends at this reset is recorded if it was hot; a new run starts at the
        // peak's last reset (its emission); resets in between leave nothing behind
        const lmask_t rec = dom & hot;
        if (rec != 0ull) {
            if (lane_of(rec)) nrec = lz_record(runs, nrec, ib + max(r0, lm + W1 + 1), ib + u, s, e, n);
        }
        lmask_t on = __ballot(lm < u - W1);
        if constexpr (SLOW) on &= live;
        hot = (hot & ~dom) | (on & hck & (~dom | em));
    }
    // the bitmap word moves on when position j - H1 - 1 enters the next 32-position span
    __device__ __forceinline__ void bw_advance() {
        const int p = jb - 1;  // last position of the span that is complete (jb is a multiple of 32 here)
        atomicOr(&ring[(p >> 5) & (LZ_RING_WORDS - 1)], bw);
        bw = 0u;
    }
    template <int U>
    __device__ __forceinline__ void dstep() {
        if constexpr (U == H1 + 1) {
            if ((jb & 16) == 0) bw_advance();
        }
        dstep_core<false>(U, t1[U & 3], hc[U & 3], ~0ull);
    }
    // a step of a block near the read's ends (or holding a very old peak): lanes that are done and indices behind
    // the read's end are frozen
    template <int U>
    __device__ __forceinline__ void dstep_edge() {
        if constexpr (U == H1 + 1) {
            if ((jb & 16) == 0) bw_advance();
        }
        const lmask_t live = ~done & __ballot((unsigned)(ib + U) < (unsigned)n);
        dstep_core<true>(U, t1[U & 3], hc[U & 3], live);
    }
    // the statistic is defined as 0 at the read's first / last W1 indices (events.c:332-338)
    // ... and so is the long window's at the last W2: it cannot exceed thr2 there.  (Its window sums reach behind the
    // read there -- whatever lies behind it made the long detector "hot" at the end of most short RNA reads, and each
    // of them paid for an exact replay of its last run.)
    __device__ __forceinline__ void slow_fix(const int u0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in1 = (unsigned)(ib + u0 + k - W1) < cnt1;
            t1[k] = in1 ? t1[k] : 0.0f;
            nk[k] &= __ballot(in1);
            hc[k] &= __ballot((unsigned)(ib + u0 + k - W2) < cnt2);
        }
    }

    // four indices U0..U0+3
    template <int U0>
    __device__ __forceinline__ void quad() {
        tstep<U0>();
        tstep<U0 + 1>();
        tstep<U0 + 2>();
        tstep<U0 + 3>();
        if (slow) slow_fix(U0);
        if constexpr (FLAGGED) {
            unsigned bad1 = 0u, bad2 = 0u;
            repair_mark<W1>(rep, next_t, ib + U0, cnt1, cnt2, bad1, bad2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nk[k] |= __ballot(((bad1 >> k) & 1u) != 0u);
                hc[k] |= __ballot(((bad2 >> k) & 1u) != 0u);
            }
        }
        // rare: evaluations whose certificate failed are redone with the reference expression
        const lmask_t anybad = (nk[0] | nk[1] | nk[2] | nk[3]) & ~done;
        if (anybad != 0ull) {
            if (lane_of(anybad)) {
                const unsigned bits = (lane_of(nk[0]) ? 1u : 0u) | (lane_of(nk[1]) ? 2u : 0u) |
                                      (lane_of(nk[2]) ? 4u : 0u) | (lane_of(nk[3]) ? 8u : 0u);
                const Redo4 r4 = redo_quad<W1, T, FLAGGED>(base, sc, rep.P, rep.P2, ib + U0, cnt1, bits, t1[0], t1[1],
                                                           t1[2], t1[3]);
                t1[0] = r4.v[0]; t1[1] = r4.v[1]; t1[2] = r4.v[2]; t1[3] = r4.v[3];
            }
        }
        if (slow) {
            dstep_edge<U0>();
            dstep_edge<U0 + 1>();
            dstep_edge<U0 + 2>();
            dstep_edge<U0 + 3>();
        } else {
            dstep<U0>();
            dstep<U0 + 1>();
            dstep<U0 + 2>();
            dstep<U0 + 3>();
        }
    }

    // ---- ring initialisation: prefix sums from the origin o = i_begin - W2 over w[k] = x[o + k], k < 2*W2;
    // A sides / long estimates of the window positions in front of the first index
    template <int K>
    __device__ __forceinline__ void init_step(double &ps, double &pq, const float (&w)[2 * W2], double (&hs)[W2 + 1],
                                              double (&hq)[W2 + 1]) {
        const float x = w[K];
        ps = ps + (double)x;
        pq = pq + (double)(x * x);
        constexpr int k1 = K + 1;  // now ps = P(o + k1)
        if constexpr (k1 <= W2) { hs[k1] = ps; hq[k1] = pq; }
        if constexpr (k1 >= W2) { Ps[(k1 - W2) % NP] = ps; Pq[(k1 - W2) % NP] = pq; }  // P(i_begin + k1 - W2)
        // short window position p = o + k1 - W1 in [i_begin - W1, i_begin): sums P(o + k1) - P(o + k1 - W1)
        if constexpr (k1 >= W2 && k1 < W2 + W1) {
            constexpr int pr = k1 - W1;
            ar[((k1 - W1 - W2) % NA + NA) % NA] = sgk_arole<W1, !FLAGGED>(ps - hs[pr], pq - hq[pr]);
        }
        // long window position p = o + k1 - W2 in [i_begin - W2, i_begin)
        if constexpr (k1 >= W2 && k1 < 2 * W2) {
            constexpr int pr = k1 - W2;
            ls[((k1 - 2 * W2) % NL + NL) % NL] = sgk_lside<W2>(ps - hs[pr], pq - hq[pr]);
        }
    }
    template <int... Ks>
    __device__ __forceinline__ void init_rings(const float (&w)[2 * W2], std::integer_sequence<int, Ks...>) {
        double ps = 0.0, pq = 0.0;
        double hs[W2 + 1], hq[W2 + 1];
        hs[0] = 0.0;
        hq[0] = 0.0;
        (init_step<Ks>(ps, pq, w, hs, hq), ...);
    }

    template <int... Qs>
    __device__ __forceinline__ void block(std::integer_sequence<int, Qs...>) {
        (quad<4 * Qs>(), ...);
    }
};

// flush 8 ring words (the 256 positions starting at pass-relative position p0, a multiple of 256) of this lane to
// the read's bitmap, in 16-bit units: i_begin is a multiple of 16, so pass-relative units are the bitmap's units, and
// only units inside the lane's own range [own_lo, own_hi) (pass-relative; multiples of 16, or the read's end) are
// written -- every owned unit is written exactly once per pass, zero or not
__device__ __forceinline__ void lz_flush(uint32_t *ring, unsigned long long *bm, int i_begin, int p0, int own_lo,
                                         int own_hi) {
    uint16_t *bm16 = reinterpret_cast<uint16_t *>(bm);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = p0 + 32 * k;
        const int wi = (p >> 5) & (LZ_RING_WORDS - 1);
        const uint32_t w = ring[wi];
        ring[wi] = 0u;
        if (p >= own_lo && p < own_hi) bm16[(i_begin + p) >> 4] = (uint16_t)(w & 0xffffu);
        if (p + 16 >= own_lo && p + 16 < own_hi) bm16[(i_begin + p + 16) >> 4] = (uint16_t)(w >> 16);
    }
}

// Issue priority by REMAINING work (round 5).  A SIMD arbitrates oldest-first among waves of equal priority: of the three
// waves that start together on a SIMD the oldest finishes its detector pass in 0.50 ms, the second in 0.71, the youngest
// in 0.97 (per-wave timestamps, profiles/r05_event_first_round.md) -- and then runs on alone, on a SIMD one wave cannot
// saturate (an instruction per 4.5 - 6 cycles instead of 2.3 - 4.45).  That, not the instruction cache (0.003 % misses)
// or address translation (949 misses per launch), is the "slow first round" of rounds 3 and 4, and the same thing happens
// when a launch drains.  A wave therefore lowers its own priority as it gets on with its read -- 3 while more than 1 200
// steps of its pass are left, 2, 1, 0 for the last 400 and in the builder -- so that whichever wave of a SIMD has most
// left to do issues first and the waves of a SIMD end together.
#ifndef SGK_PRIO_POLICY
#define SGK_PRIO_POLICY 1   // 0: none; 1: by the steps left of the detector pass 3 / 2 / 1 / 0, builder 0; 2: ... builder 3; 3: builder 3 only
#endif
__device__ __forceinline__ uint32_t prio_policy(uint32_t dev) {
#ifdef SGK_DEV
    return ((dev >> 8) & 7u) ? ((dev >> 8) & 7u) - 1u : (uint32_t)SGK_PRIO_POLICY;   // dev bits 8..10: policy + 1
#else
    (void)dev;
    return (uint32_t)SGK_PRIO_POLICY;
#endif
}

// One pass of the lazy detector over the wave's chunks.
//   given  : (per lane) this lane starts from snap.st0 -- the true state at its first index: a re-run of a lane whose
//            speculation failed, or the first lane of a segment that is run from a known state; the other lanes
//            start from the fresh state (true for the first lane of a read, speculative elsewhere)
//   lead   : this lane's warm-up before its chunk start s (0 for a lane that starts from a true state)
//   steps  : indices every lane runs (wave-uniform: warm-up + chunk length)
//   active : whether this lane runs in this pass
// Writes the lane's bitmap words, its hot-run records and (speculative pass) snap.init / snap.at_e.
template <int W1, typename T, bool FLAGGED>
__device__ __forceinline__ void pass_lazy(const ReadCtx<T> &rc, bool given, int lead, int steps, bool active, int s,
                                          int e, LzLds *L, const RepairCtx *rep, bool by_progress = false) {
    using LP = LazyPass<W1, T, FLAGGED>;
    constexpr int W2 = LP::W2, R = LP::R;
    if (!__any(active)) return;
    const int l = lane_id();
    LP f;
    f.n = (int)rc.n;
    // lanes that do not take part in the pass own nothing: nothing they emit or record can land anywhere
    f.s = active ? s : 0x7fffffff;
    f.e = active ? e : 0x7fffffff;
    f.bm = rc.bm;
    f.sc = rc.sc;
    f.base = rc.base;
    f.lo = (int)(rc.lo < -(1 << 30) ? -(1 << 30) : rc.lo);
    f.hi = (int)(rc.hi > 0x7fffffffLL ? 0x7fffffffLL : rc.hi);
    f.ring = L->ring[l];
    f.runs = L->runs[l];
    const int n = f.n;
    const int i_begin = s - lead;  // multiple of 16, never negative
    f.i_begin = i_begin;
#pragma unroll
    for (int k = 0; k < LZ_RING_WORDS; ++k) f.ring[k] = 0u;
    if (active) f.ring[LZ_PRE] = 0xffffffffu;  // no inherited emission in this pass yet
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.t1[k] = 0.0f; f.nk[k] = 0ull; f.hc[k] = 0ull; }
    {
        // x[i_begin - W2 .. i_begin + W2): prefix sums from the origin i_begin - W2
        float w[2 * W2];
#pragma unroll
        for (int k = 0; k < 2 * W2; ++k) {
            int p = i_begin - W2 + k;
            p = p > f.hi - 1 ? f.hi - 1 : p;
            p = p < 0 ? 0 : p;  // lane 0: positions before the read (their window positions are marked below)
            w[k] = to_pa(f.base[p], f.sc);
        }
#pragma unroll
        for (int k = 0; k < LP::NP; ++k) { f.Ps[k] = 0.0; f.Pq[k] = 0.0; }
        f.init_rings(w, std::make_integer_sequence<int, 2 * W2>{});
        // The statistic is defined as 0 at the read's first W1 indices (events.c:332-338): their A side is a window
        // position in front of the read.  Lane 0 marks those ring entries (NaN): the evaluation cannot be certified
        // and the exact path returns the 0.
        if (i_begin == 0) {
#pragma unroll
            for (int k = 0; k < LP::NA; ++k) f.ar[k].va = __builtin_nan("");
        }
    }
    // leading samples of the first block: x[i_begin + W2 .. +16)
    f.load_lead(f.cur, i_begin + W2);

    // detector state
    f.sv = FLT_MAX;
    f.sp = 0;
    f.inpk = 0ull; f.val = 0ull; f.strong = 0ull; f.hot = 0ull;
#pragma unroll
    for (int k = 0; k <= LP::H1; ++k) f.hist[k] = 0ull;
    f.bw = 0u;
    f.lm = LZ_NONE;
    f.r0 = 0;  // the (pseudo) reset a speculative pass starts from; index 0 for lane 0
    if (__any(given)) {
        LzSnapState st = L->snap.st0[l];
        if (!given) { st.sp = -1; st.sv = FLT_MAX; st.lm = LZ_NONE; st.r0 = i_begin; st.bits = 0u; }  // fresh
        f.sv = st.sv;
        f.inpk = __ballot((st.bits & 1u) != 0u);
        f.val = __ballot((st.bits & 2u) != 0u);
        f.strong = __ballot((st.bits & 4u) != 0u);
        f.hot = __ballot((st.bits & 8u) != 0u);
        f.sp = (st.bits & 1u) ? st.sp - i_begin : 0;
        f.lm = st.lm == LZ_NONE ? LZ_NONE : st.lm - W1 - i_begin;  // handed over as masked_to; kept as the peak position
#pragma unroll
        for (int k = 0; k <= LP::H1; ++k) f.hist[k] = __ballot((st.bits & 1u) && st.sp == i_begin - 1 - k);
        f.r0 = st.r0 - i_begin;
    }
    f.nrec = 0;
    f.done = ~__ballot(active);
    f.flushed = 0;
    if constexpr (FLAGGED) {
        f.dirty = 2 * W2;  // the ring was filled by a handful of additions that were not checked
        f.rep = *rep;
        // first event whose influence [t-W2+1, t+W2] is not entirely before this pass' first index
        f.next_t = 0x7fffffff;
        for (int k = 0; k < f.rep.nev; ++k) {
            const int t = f.rep.ev[k];
            if (t + W2 >= i_begin && t < f.next_t) f.next_t = t;
        }
    }
    const int main_steps = steps;
    f.cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    f.cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    // pass-relative range of the positions this lane owns (its bitmap words)
    const int own_lo = lead, own_hi = active ? lead + (e - s) : lead;  // (lane 0: lead = 0)

    auto snapshot = [&](int ib) -> LzSnapState {
        // ib: absolute index of the block about to start (positions are relative to it)
        LzSnapState st;
        const bool ip = lane_of(f.inpk);
        st.sv = f.sv;
        st.sp = ip ? ib + f.sp : -1;
        st.lm = (f.lm + W1 < 0) ? LZ_NONE : ib + f.lm + W1;  // normalised when it no longer masks
        // what the long detector's current run STARTS with: the first index behind the last reset that is not masked.
        // (The reset index alone is not enough once the mask is normalised away: a lane that takes this state over
        // would replay a hot run from the reset, through indices the mask hid from the reference's long detector --
        // found by the soak with a 16-sample warm-up, tests/golden/soak_seed41_*.npz.)
        st.r0 = ib + max(f.r0, f.lm + W1 + 1);
        st.bits = (ip ? 1u : 0u) | ((ip && lane_of(f.val)) ? 2u : 0u) | ((ip && lane_of(f.strong)) ? 4u : 0u) |
                  (lane_of(f.hot) ? 8u : 0u);
        return st;
    };

    // (by_progress: the wave's first pass over its span, under a policy that ranks by what is left to do -- in steps, not
    // as a fraction of the span: a segment of a cut read starts where a whole read is when it has as much left)
    constexpr int PRIO_STEPS = 400;   // a quarter of the pass over a 100 000-sample read
    const int q1 = by_progress ? main_steps - 3 * PRIO_STEPS : -1, q2 = by_progress ? main_steps - 2 * PRIO_STEPS : -1,
              q3 = by_progress ? main_steps - PRIO_STEPS : -1;
    if (by_progress) {
        if (q1 > 0) __builtin_amdgcn_s_setprio(3);
        else if (q2 > 0) __builtin_amdgcn_s_setprio(2);
        else if (q3 > 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    int jb = 0;
    for (;;) {
        const int ib = i_begin + jb;
        if ((jb & 255) == 0 && jb >= 512) {
            lz_flush(f.ring, f.bm, i_begin, jb - 512, own_lo, own_hi);
            f.flushed = jb - 256;
        }
        // blocks that touch the read's last W2 indices (the statistics are defined as 0 there) or its end take the
        // predicated forms of the steps; so does a block in which some lane holds a peak older than the bitmap ring
        // or one from in front of its own range
        const bool lane_edge = ib + R - 1 > n - W2;
        // (... or a peak in front of the lane's own range in the range's first block: it may be emitted there within
        // w/2 indices, as a "usual" peak whose bit the fast steps would put into the bitmap word; in later blocks it is
        // an "older peak", which the fast steps hand to the ring or, in front of the range, to LZ_PRE)
        const bool old_peak = f.sp < -(256 - 2 * R) || (f.sp + jb < own_lo && jb == own_lo);
        f.oldpeak = (__ballot(old_peak) & f.inpk & ~f.done) != 0ull;
        f.slow = f.oldpeak || (__ballot(lane_edge) & ~f.done) != 0ull;
        f.ib = ib;
        f.jb = jb;
        // issue the loads of the NEXT block's leading samples now; consumed one iteration later
        Lead16<T> nxt;
        f.load_lead(nxt, ib + R + W2);
        f.block(std::make_integer_sequence<int, R / 4>{});
        f.cur = nxt;
        // rebase the block-relative positions
        f.sp -= R;
        f.lm = f.lm < LZ_NONE ? LZ_NONE : f.lm - R;
        f.r0 -= R;
        if constexpr (FLAGGED) f.dirty = f.dirty < -(1 << 20) ? f.dirty : f.dirty - R;
        jb += R;
        if (jb == q1) __builtin_amdgcn_s_setprio(2);
        else if (jb == q2) __builtin_amdgcn_s_setprio(1);
        else if (jb == q3) __builtin_amdgcn_s_setprio(0);
        {
            // state snapshots live in LDS (they are only needed after the pass)
            const int nb = i_begin + jb;  // first index of the next block
            if (active && lead > 0 && jb == lead) L->snap.init[l] = snapshot(nb);
            if (active && nb == e) L->snap.at_e[l] = snapshot(nb);
            // a lane stops at the end of its range; a peak still pending there is emitted by the lane behind
            // (lz_emit_slow), and dropped at the read's end as in the reference, whose loop ends at n-1
            const lmask_t reach = __ballot(nb >= e) & ~f.done;
            // the run still open at the end of the read is replayed by the lane that holds the read's last index --
            // recorded HERE: the lane may step on (other lanes of the wave have more to do, and in k_event_multi their
            // reads are longer) through whatever lies behind the read, and its run bookkeeping with it
            if ((reach & f.hot) != 0ull) {
                if (lane_of(reach & f.hot) && e == n) f.nrec = lz_record(f.runs, f.nrec, nb + max(f.r0, f.lm + W1 + 1), n, s, e, n);
            }
            f.done |= reach;
            // ... and it steps on without the exact redo of uncertified statistics: whatever it decides from here on
            // must not land in its own range.  Out of its peak, every position it can emit lies behind the range.
            f.inpk &= ~reach;
            f.val &= ~reach;
            f.strong &= ~reach;
        }
        if (jb >= main_steps && f.done == ~0ull) break;
    }
    // remaining ring words (the current bitmap word first)
    {
        const int p = jb - LP::H1 - 2 < 0 ? 0 : jb - LP::H1 - 2;  // a position inside the word bw stands for
        atomicOr(&f.ring[(p >> 5) & (LZ_RING_WORDS - 1)], f.bw);
    }
    for (int p0 = f.flushed; p0 < jb; p0 += 256) lz_flush(f.ring, f.bm, i_begin, p0, own_lo, own_hi);
    if (active) L->nrec[l] = f.nrec;
}

// Exact replay of the long detector (events.c:383-440, k = 1) over one hot run per lane: from the fresh state a
// reset leaves, over the indices [i, b) of the run (inside a run masked_to does not change and every index is
// processed).  Peaks are ORed into the read's bitmap if they lie in [bits_lo, bits_hi).
// found != null: the peaks are not set in the bitmap, *found is set instead (chain_segment: a peak in front of a seam
// lies in another wave's words)
template <int W1, typename T, bool FLAGGED>
__device__ void replay_run(const ReadCtx<T> &rc, const RepairCtx *rep, bool has, int i, int b, int bits_lo,
                           int bits_hi, EvHeader *hdr, int *found = nullptr) {
    if (has && b > i) atomicAdd(&hdr->n_replay_idx, (unsigned long long)(b - i));
    constexpr int W2 = 2 * W1;
    constexpr float ph = DetParam<W1>::ph, thr2 = DetParam<W1>::thr2;
    const int n = (int)rc.n;
    const unsigned cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    uint32_t *bm32 = reinterpret_cast<uint32_t *>(rc.bm);
    int lp = -1;
    float lv = FLT_MAX;
    bool lvalid = false;
    while (__any(has && i < b)) {
        if (has && i < b) {
            float v2 = 0.0f;
            if ((unsigned)(i - W2) < cnt2) {
                if constexpr (FLAGGED) v2 = tstat_prefix_at(rep->P, rep->P2, i, W2);
                else v2 = tstat_exact_at<T>(rc.base, rc.sc, i, W2);
            }
            if (lp < 0) {
                if (v2 < lv) {
                    lv = v2;
                } else if (v2 - lv > ph) {
                    lv = v2;
                    lp = i;
                }
            } else {
                if (v2 > lv) {
                    lv = v2;
                    lp = i;
                }
                if (lv - v2 > ph && lv > thr2) lvalid = true;
                if (lvalid && (i - lp) > W2 / 2) {
                    if (lp > 0 && lp < n && lp >= bits_lo && lp < bits_hi) {
                        if (found) *found = 1;
                        else atomicOr(&bm32[lp >> 5], 1u << (lp & 31));
                    }
                    lp = -1;
                    lv = v2;
                    lvalid = false;
                }
            }
            ++i;
        }
    }
}
// ... over the recorded hot runs of the wave's lanes.  bits_lo: first index of the span this wave owns (a run that
// began in front of it leaves its peaks in front of the span to whoever replays the span's cross runs)
template <int W1, typename T, bool FLAGGED>
__device__ void replay_long_runs(const ReadCtx<T> &rc, LzLds *L, const RepairCtx *rep, bool active, int bits_lo,
                                 EvHeader *hdr) {
    const int l = lane_id();
    const int nrec = active ? L->nrec[l] : 0;
    for (int k = 0; k < LZ_NREC; ++k) {
        const bool has = k < nrec;
        if (!__any(has)) break;
        const int i = has ? L->runs[l][k].a : 0;
        const int b = has ? L->runs[l][k].b : 0;
        replay_run<W1, T, FLAGGED>(rc, rep, has, i, b, bits_lo, 0x7fffffff, hdr);
    }
}

// speculative pass + verification / re-run loop + replay of the hot long-detector runs over the span [a, b) of a read
// (a multiple of 16; the whole read: a = 0, b = n).
//   mode 0: the state at a is the fresh one (a = 0: the read's start)
//   mode 1: unknown: the first lane warms up in front of a like every other lane; the state it reached at a is
//           left in seg->init0, to be compared with the end state of the span in front (chain_segment)
//   mode 2: the state at a is L->snap.st0[0], put there by the caller
// seg (spans of a read that several waves share; null otherwise) receives the state at b and the hot runs that began
// in front of a.
// Returns 0 when the span is done, 1 when the fast pass cannot take the read (alignment / room around the read), 2
// when a lane met more hot runs than it can record (pathological signal: constant stretches, tiny variances).
//
// MULTI (k_event_multi): the wave holds 64 / lanes reads, `lanes` consecutive lanes each (rc, b and the return code are
// per lane; a = 0, mode 0, no seg): a short read on all 64 lanes spends more steps on warm-ups than on its samples.
template <int W1, typename T, bool FLAGGED, bool MULTI = false>
__device__ __forceinline__ int detect_span(const ReadCtx<T> &rc, EvHeader *hdr, LzLds *L, const RepairCtx *rep,
                                           const int a, const int b, const int mode, const int lead_override,
                                           SegState *seg, const int lanes = 64) {
    const int n = (int)rc.n;
    if constexpr (!MULTI) {
        if (b <= a) return 0;
    }
    // speculative warm-up before every chunk.  RNA events are ~5x longer, so the automata converge later: with 64
    // samples ~1.4 % of the chunk boundaries need a re-run, with 256 about 0.002 %.  A re-run costs the wave one
    // more pass over a chunk (K samples), the warm-up costs `lead` samples per lane: short reads (small K) are
    // better off with a short warm-up and the occasional re-run, long reads with a long one.
    // (DNA, long reads: 32 samples were tried: 6 re-runs per 640 000 chunk boundaries of the benchmark, no gain.)
    const int len = b - a;
    int lead = len < 32768 ? SGK_LEAD_DNA_SHORT : SGK_LEAD_DNA;
    if (W1 == 7) lead = len <= 32768 ? SGK_LEAD_RNA_SHORT : SGK_LEAD_RNA;
    if constexpr (MULTI) {
        // the same rule for the same chunk length: what 64 lanes would see of a read 64 / lanes times as long
        const long long len64 = (long long)len * (64 / lanes);
        lead = len64 < 32768 ? SGK_LEAD_DNA_SHORT : SGK_LEAD_DNA;
        if (W1 == 7) lead = len64 <= 32768 ? SGK_LEAD_RNA_SHORT : SGK_LEAD_RNA;
    }
    if (lead_override > 0) lead = lead_override;
    // the fast pass uses unguarded 4-byte-aligned 32-byte vector loads: it needs 16 readable samples behind the
    // read; other reads take the exact fallback
    const bool no_fast = (reinterpret_cast<uintptr_t>(rc.base) & 3u) != 0 || rc.hi < (int64_t)n + 16;
    if constexpr (!MULTI) {
        if (no_fast) return 1;
    }
    const int c = MULTI ? (lane_id() & (lanes - 1)) : lane_id();  // lane within its read
    int K, s, e0, lead_c;
    if (mode == 1) {
        // every lane warms up: lane c owns [a + cK, a + (c+1)K)
        K = 16 * ((len + 1023) / 1024);
        s = a + c * K;
        e0 = s + K;
        lead_c = lead;
    } else {
        K = MULTI ? chunk_len_lanes(len, lead, lanes) : chunk_len_fast(len, lead);
        s = c == 0 ? a : a + c * K + lead;
        e0 = c == 0 ? a + lead + K : s + K;
        lead_c = c > 0 ? lead : 0;
    }
    int TT = lead + K;
    int Kmax = K;
    if constexpr (MULTI) {  // the passes' step counts are the wave's: the longest of its reads
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(Kmax, d, 64), t = __shfl_xor(TT, d, 64);
            Kmax = o > Kmax ? o : Kmax;
            TT = t > TT ? t : TT;
        }
    }
    const int e = e0 < b ? e0 : b;
    const bool active = s < b && !(MULTI && no_fast);
    {
        LzSnapState z;
        z.sp = -1; z.sv = FLT_MAX; z.lm = LZ_NONE; z.r0 = 0; z.bits = 0u;
        L->snap.init[lane_id()] = z;
        L->snap.at_e[lane_id()] = z;
        L->nrec[lane_id()] = 0;
        L->ring[lane_id()][LZ_PRE] = 0xffffffffu;
    }
    bool run = active;
    bool first = true;
    const int l = lane_id();
    const uint32_t pol = FLAGGED ? 0u : prio_policy(rc.dev);
    for (int iter = 0; iter < 66; ++iter) {
        pass_lazy<W1, T, FLAGGED>(rc, first ? (mode == 2 && c == 0) : true, first ? lead_c : 0, first ? TT : Kmax, run, s, e,
                                  L, rep, first && (pol == 1u || pol == 2u));
        __syncthreads();
        // chunk c is right iff it started (at s) from the state chunk c-1 ended with
        const LzSnapState pe = L->snap.at_e[c > 0 ? l - 1 : l];
        const LzSnapState mine = L->snap.init[l];
        const bool bad = active && c > 0 && !lz_equal(pe, mine);
        const unsigned long long badmask = __ballot(bad);
        if (badmask == 0ull) break;
        __syncthreads();
        if (bad) {
            L->snap.init[l] = pe;
            L->snap.st0[l] = pe;
        }
        run = bad;
        first = false;
        if (l == 0) atomicAdd(&hdr->n_rerun, (uint32_t)__popcll(badmask));
        __syncthreads();
    }
    int rcode = 0;
    if constexpr (MULTI) {
        // per read: the lanes of a read that cannot be taken here, or whose lanes met too many hot runs, stand aside
        const unsigned long long over = __ballot(active && L->nrec[l] > LZ_NREC);
        const unsigned long long grp = (lanes >= 64 ? ~0ull : ((1ull << lanes) - 1ull)) << (l & ~(lanes - 1));
        if (no_fast) rcode = 1;
        else if (over & grp) rcode = 2;
    } else {
        if (__any(active && L->nrec[l] > LZ_NREC)) return 2;
    }
    const bool mine_ok = active && rcode == 0;
    if (seg) {
        // what the neighbours need: the states at both ends, the runs that began in front of the span
        const int last = __popcll(__ballot(active)) - 1;
        if (l == 0) {
            seg->init0 = L->snap.init[0];
            seg->end = L->snap.at_e[last];
        }
        const int nrec = active ? L->nrec[l] : 0;
        int ncross = 0;
        for (int k = 0; k < nrec; ++k) ncross += (L->runs[l][k].a < a) ? 1 : 0;
        const int incl = wave_incl_scan_i(ncross);
        const int total = wave_last_i(incl);
        if (total > SEG_CROSS_MAX) rcode = 2;
        else {
            int at = incl - ncross;
            for (int k = 0; k < nrec; ++k) {
                if (L->runs[l][k].a < a) seg->cross[at++] = L->runs[l][k];
            }
        }
        if (l == 0) seg->n_cross = total > SEG_CROSS_MAX ? 0u : (uint32_t)total;
    }
    // inherited emissions (lz_emit_slow) of the lanes' accepted runs.  Inside the span the bit is set here; one in
    // front of it lies in another wave's words: it is left in seg->pre -- the segment OWNS that boundary (chain_segment
    // builds the event that ends there), the bitmap never shows it.
    const int pre = mine_ok ? (int)L->ring[l][LZ_PRE] : -1;
    const bool pre_out = pre >= 0 && pre < a;
    if (seg) {
        const int npre = pre_out ? 1 : 0;
        const int incl = wave_incl_scan_i(npre);
        const int total = wave_last_i(incl);
        if (total > SEG_PRE_MAX) rcode = 2;
        else if (npre) seg->pre[incl - 1] = pre;
        if (l == 0) seg->n_pre = total > SEG_PRE_MAX ? 0u : (uint32_t)total;
    }
    const unsigned long long hotm = __ballot(mine_ok && L->nrec[l] > 0);
    const bool anypre = __any(pre >= 0);
    if (hotm != 0ull || anypre) {
        if (l == 0 && hotm != 0ull) atomicAdd(&hdr->n_hot_runs, (uint32_t)__popcll(hotm));
        __threadfence_block();
        __syncthreads();  // every lane's bitmap words are in memory before anything is ORed into them
        if (pre >= 0 && !pre_out) atomicOr(reinterpret_cast<uint32_t *>(rc.bm) + (pre >> 5), 1u << (pre & 31));
        if (hotm != 0ull) replay_long_runs<W1, T, FLAGGED>(rc, L, rep, mine_ok, a, hdr);
    }
    return rcode;
}
// one wave, one read
template <int W1, typename T, bool FLAGGED>
__device__ __forceinline__ int detect_read_lazy(const ReadCtx<T> &rc, EvHeader *hdr, LzLds *L, const RepairCtx *rep) {
    const int n = (int)rc.n;
    if (n <= 0) return 0;
    return detect_span<W1, T, FLAGGED>(rc, hdr, L, rep, 0, n, 0, 0, nullptr);
}

__device__ inline bool guard_ok(float mn, float mx, int64_t n) {
    if (!(mx > 0.0f)) return true;  // all samples zero
    // range in which the certified arithmetic of tstat_math.h (sgk_tstat_try_ab) has no subnormal intermediates
    if (mn < 9.5367431640625e-07f || mx > 1048576.0f) return false;
    const int eb = ilogb((double)n * (double)mx), em = ilogb((double)mn);
    if (eb - em > 29) return false;
    const float mnq = mn * mn, mxq = mx * mx;
    if (mnq < FLT_MIN) return false;
    const int ebq = ilogb((double)n * (double)mxq), emq = ilogb((double)mnq);
    if (ebq - emq > 29) return false;
    // the A side's mean is formed with one multiply (tstat_math.h: sgk_arole<W, SHORT>): magnitudes within 2^16
    return ilogbf(mx) - ilogbf(mn) <= 16;
}

// Generic detector over one read by one wave (prefix arrays required).
template <int W1, typename T>
__device__ void detect_read(const ReadCtx<T> &rc, EvHeader *hdr) {
    const int64_t n = rc.n;
    if (n <= 0) return;
    const uint32_t K = chunk_len(n);
    const int c = lane_id();
    const int64_t s = (int64_t)c * K;
    const int64_t e = (s + K < n) ? s + K : n;
    const bool active = s < n;
    const DetState fresh = det_fresh(0);
    DetState at_s = fresh, at_e = fresh;
    detect_pass<W1, T>(rc, LEAD, active, s, e, K, fresh, at_s, at_e);
    DetState init = at_s;
    for (int iter = 0; iter < 64; ++iter) {
        const DetState pe = det_shfl_up(at_e);
        const bool bad = active && c > 0 && !det_equal(pe, init);
        const unsigned long long badmask = __ballot(bad);
        if (badmask == 0ull) break;
        if (bad) init = pe;
        DetState unused = fresh;
        detect_pass<W1, T>(rc, 0, bad, s, e, K, pe, unused, at_e);
        if (c == 0) atomicAdd(&hdr->n_rerun, (uint32_t)__popcll(badmask));
    }
}

// ---------------------------------------------------------------- event builder

// src/events.c:457-473 (create_event)
__device__ inline void store_event(const EvArgs &a, uint64_t slot0, uint64_t cap, uint64_t k, uint32_t ps,
                                   uint32_t pe, double dsum, double dsumsq, bool &overflow) {
    if (k >= cap) { overflow = true; return; }
    const float len = (float)(pe - ps);
    const float m = (float)dsum / len;
    const float dsq = (float)dsumsq;
    const float var = dsq / len - m * m;
    const float sd = sqrtf(fmaxf(var, 0.0f));
    sgk_event_rec_t e;
    e.start = ps;
    e.length = pe - ps;
    e.mean = m;
    e.stdv = sd;
    a.events[slot0 + k] = e;
}

constexpr int BT = 32;      // samples per lane per builder tile: 64 bytes of int16, one 32-bit bitmap word
// create_event (events.c:457-473) for the fast builder: the two divisions by the event length share one refined
// reciprocal (tstat_math.h: bit-identical to `/` inside the range guard); one 16-byte store per event.
struct EvOut {
    uint4 *ev;   // the read's first slot
    uint32_t cap;
#ifdef SGK_DEV
    bool raw;    // SGK_DEV_RAW_EVENTS
#endif
};
__device__ __forceinline__ void store_event_fast(const EvOut &o, uint32_t k, uint32_t ps, uint32_t pe, double dsum,
                                                 double dsumsq, bool &overflow) {
    if (k >= o.cap) { overflow = true; return; }
#ifdef SGK_DEV
    if (o.raw) {
        uint4 e;
        e.x = ps; e.y = pe - ps; e.z = __float_as_uint((float)dsum); e.w = __float_as_uint((float)dsumsq);
        o.ev[k] = e;
        return;
    }
#endif
    const float len = (float)(pe - ps);
    const float r1 = sgk_refined_rcp(len);
    const float m = sgk_div_with_rcp((float)dsum, len, r1);
    const float var = sgk_div_with_rcp((float)dsumsq, len, r1) - m * m;
    const float sd = sqrtf(fmaxf(var, 0.0f));
    uint4 e;
    e.x = ps;
    e.y = pe - ps;
    e.z = __float_as_uint(m);
    e.w = __float_as_uint(sd);
    o.ev[k] = e;
}

#ifndef SGK_BREC
#define SGK_BREC 576
#endif
// Boundary records per tile in LDS: {S, S2} as one 16-byte record + a 16-bit tile-relative position (10.6 KB per
// wave with the lane prefixes).  The detector can emit a boundary every 3 samples (683 per tile), but sizing LDS for
// that costs occupancy; a tile with more than BREC boundaries (events shorter than 3.6 samples on average over 2048
// samples: never seen on nanopore data, sp1 peaks at 425) sends its read to k_event_fallback instead.
constexpr int BREC = SGK_BREC;
struct __attribute__((aligned(16))) BuildRec {
    double S, S2;
};
struct BuildLds {
    BuildRec rec[BREC];
    double pt[64];
    double pt2[64];
    uint16_t p[BREC];  // tile-relative sample index of the boundary
};
static_assert(sizeof(BuildLds) <= 11776, "builder LDS budget: 13 waves per CU");

// min non-zero |x| / max |x| of a read from the extremes of its raw samples (x = (raw + off) * unit is monotone in
// raw); returns false when the read crosses or touches zero pA (the smallest non-zero magnitude is then not known
// from the extremes; such reads fail the guard anyway: it tolerates a ratio of ~64 between the magnitudes)
__device__ inline bool raw_extremes_to_pa(int rmin, int rmax, const Scale &sc, float &mn, float &mx) {
    const float a = ((float)rmin + sc.offf), b = ((float)rmax + sc.offf);
    const float xa = fabsf(a * sc.unit), xb = fabsf(b * sc.unit);
    mn = fminf(xa, xb);
    mx = fmaxf(xa, xb);
    const bool same_sign = (a > 0.0f && b > 0.0f) || (a < 0.0f && b < 0.0f);
    return same_sign && mn > 0.0f && mx < __builtin_inff();
}

typedef short sgk_s2 __attribute__((ext_vector_type(2)));

// One lane's walk over its 32 samples of a tile: lane-relative double prefix sums, one record per boundary bit.
// FULL: every sample of the tile is inside the read (no per-sample validity select).
// A lane's 32 samples of a tile AS THEY SIT IN MEMORY (16 dwords of packed int16, or 32 floats).  Kept packed on purpose:
// as an array of 32 int16 elements the compiler gives every sample a register of its own and unpacks the high halves
// right behind the load -- an s_waitcnt vmcnt directly after the "prefetch" of the next tile, i.e. no prefetch at all
// (round 5: the whole memory latency was exposed once per tile).  The walk converts straight from the packed words
// (v_cvt_f32_i32_sdwa).
template <typename T>
struct TileRegs {
    static constexpr int NW = BT * (int)sizeof(T) / 4;
    uint32_t w[NW];
    __device__ __forceinline__ float pa(int k, const Scale &sc) const {   // k: a constant after unrolling
        if constexpr (std::is_same<T, int16_t>::value) {
            const int v = (k & 1) ? ((int)w[k / 2] >> 16) : (int)(short)(w[k / 2] & 0xffffu);
            return ((float)v + sc.offf) * sc.unit;
        } else {
            return __uint_as_float(w[k]);
        }
    }
};
template <typename T, bool FULL>
__device__ __forceinline__ void build_walk(const TileRegs<T> &buf, uint32_t bits, int nvalid, const Scale &sc, int l,
                                           int excl, BuildLds *L, double &S, double &S2, uint32_t &mnb,
                                           uint32_t &mxb) {
    // (LDS addresses as 32-bit offsets: what ds_write takes)
    typedef __attribute__((address_space(3))) char *LdsBytes;
    typedef __attribute__((address_space(3))) uint16_t *LdsU16;
    uint32_t rr32 = (uint32_t)(uintptr_t)(LdsBytes)(char *)L->rec + (uint32_t)excl * 16u;
    uint32_t rp32 = (uint32_t)(uintptr_t)(LdsBytes)(char *)L->p + (uint32_t)excl * 2u;
#pragma unroll
    for (int k = 0; k < BT; ++k) {
        float x = buf.pa(k, sc);
        if (!FULL && k >= nvalid) x = 0.0f;
        const float xq = x * x;
        if constexpr (std::is_same<T, float>::value) {
            // pA input: the guard's extremes are tracked on the bit patterns (non-negative floats order like
            // unsigned integers; zero - 1 wraps to the top, so it never wins the minimum; inf / nan end up above
            // every finite value and fail the guard)
            const uint32_t ab = __float_as_uint(x) & 0x7fffffffu;
            mxb = ab > mxb ? ab : mxb;
            mnb = (ab - 1u) < mnb ? (ab - 1u) : mnb;
        }
        if ((bits >> k) & 1u) {
            typedef double __attribute__((ext_vector_type(2))) sgk_d2;
            *(__attribute__((address_space(3))) sgk_d2 *)(uintptr_t)rr32 = sgk_d2{S, S2};   // BuildRec {S, S2}
            *(LdsU16)(uintptr_t)rp32 = (uint16_t)(l * BT + k);
            // (in place, under the lane mask: written as `rr += 16` the two pointers come out as an add into a new
            // register plus a move each -- four vector instructions per sample instead of two)
            asm volatile("v_add_u32 %0, 16, %0\n\tv_add_u32 %1, 2, %1" : "+v"(rr32), "+v"(rp32));
        }
        S = S + (double)x;
        S2 = S2 + (double)xq;
    }
}

// SEG: the wave builds the events of one segment [seg_a, seg_b) of a long read (several waves share the read): the
// events that END at a boundary inside the segment, and the read's last event if the segment is the read's last.
// It walks from the last boundary in front of the segment (prev_p; none: from the read's start), at the event rank the
// boundaries in front give (cnt_before); extremes and flags go to st, the read's verdict is chain_segment's.  In front
// of the segment the bitmap words are another wave's: the only boundaries the walk knows there are the one it starts
// at (prev_p) and the ones this segment owns (pre: peaks that were pending at the seam).
template <typename T, bool SEG = false>
__device__ __forceinline__ void build_read(const EvArgs &a, const ReadCtx<T> &rc, uint32_t r, BuildLds *L, bool declined,
                           int64_t seg_a = 0, int64_t seg_b = 0, SegState *st = nullptr, uint32_t cnt_before = 0,
                           int prev_p = -1, const int *pre = nullptr, int n_pre = 0) {
    const int64_t n = rc.n;
    const int l = lane_id();
    const uint64_t slot0 = a.ev_slots[r], cap = a.ev_slots[r + 1] - slot0;
    if (n <= 0) {
        if (l == 0) { a.n_events[r] = 0; a.flags[r] = 0; }
        return;
    }
    const uint32_t *bm32 = reinterpret_cast<const uint32_t *>(rc.bm);
    {
        const uint32_t pol = prio_policy(a.dev);
        if (pol == 2u || pol == 3u) __builtin_amdgcn_s_setprio(3);
        else if (pol == 1u) __builtin_amdgcn_s_setprio(0);
    }
    EvOut eo;
    eo.ev = reinterpret_cast<uint4 *>(a.events + slot0);
    eo.cap = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
#ifdef SGK_DEV
    eo.raw = (a.dev & SGK_DEV_RAW_EVENTS) != 0u;
#endif
    bool overflow = false, dense = false;
    uint32_t rank = 0, prevp = 0;
    // SEG: bits in [bit_lo, bit_hi) count; the first of them (the boundary in front of the segment) ends no event of
    // this segment: its record only starts the next one
    int64_t bit_lo = 0, bit_hi = n, walk0 = 0;
    uint32_t skip_rank = 0xffffffffu;
    if constexpr (SEG) {
        bit_hi = seg_b;
        if (prev_p >= 0) {
            bit_lo = prev_p;
            walk0 = bit_lo & ~(int64_t)31;
            rank = cnt_before - 1u;
            skip_rank = rank;
        }
    }
    double Gprev = 0.0, G2prev = 0.0;  // prefix sums at the previous boundary, relative to the current tile start
    // exactness guard inputs: int16 reads track the extremes of the RAW samples (packed 16-bit min / max, two samples
    // per instruction); pA reads the extremes of the float bit patterns
    uint32_t mnb = 0xffffffffu, mxb = 0u;
    sgk_s2 rmin2 = {32767, 32767}, rmax2 = {-32768, -32768};
    constexpr int NV = BT * (int)sizeof(T) / 16;
    // tile loader: this lane's 32 samples and its 32 bitmap bits.  The next tile is fetched while the
    // current one is processed (register double buffer).
    // (bits of the tile = (raw & keep) | extra: the masks are formed WITHOUT touching the loaded word, so that nothing waits
    // for the load where it is issued -- one `bits &= mask` here put an s_waitcnt vmcnt(0) right behind the prefetch)
    auto load_tile = [&](int64_t tb, TileRegs<T> &buf, uint32_t &raw, uint32_t &keep, uint32_t &extra, int &nvalid) {
        const int64_t pos0 = tb + (int64_t)l * BT;
        raw = (pos0 < n) ? bm32[pos0 >> 5] : 0u;
        keep = 0xffffffffu;
        extra = 0u;
        const int64_t rem = n - pos0;
        nvalid = rem <= 0 ? 0 : (rem >= BT ? BT : (int)rem);
        if (nvalid < BT) keep = (nvalid == 0) ? 0u : ((1u << nvalid) - 1u);
        if constexpr (SEG) {
            if (pos0 < seg_a) {
                uint32_t sb = 0u;
                if (prev_p >= pos0 && prev_p < pos0 + BT) sb |= 1u << (int)(prev_p - pos0);
                for (int k = 0; k < n_pre; ++k) {
                    const int64_t q = pre[k];
                    if (q >= pos0 && q < pos0 + BT) sb |= 1u << (int)(q - pos0);
                }
                extra = sb & keep;
                keep = 0u;
            }
            const int64_t dl = bit_lo - pos0, dh = bit_hi - pos0;
            uint32_t m = 0xffffffffu;
            if (dl > 0) m = dl >= BT ? 0u : ~((1u << (int)dl) - 1u);
            if (dh < BT) m = dh <= 0 ? 0u : (m & ((1u << (int)dh) - 1u));
            keep &= m;
            extra &= m;
        }
        if (pos0 >= n) {
            // lanes behind the read's end (every read's last tile has some): nothing to load
#pragma unroll
            for (int k = 0; k < TileRegs<T>::NW; ++k) buf.w[k] = 0u;
        } else if (rc.vec_ok && pos0 + BT <= rc.hi) {
            const uint4 *src = reinterpret_cast<const uint4 *>(rc.base + pos0);
            uint4 v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = src[k];
            __builtin_memcpy(buf.w, v, sizeof(buf.w));
        } else {
            // (a read on an odd address / at the end of the buffer: element by element, packed by hand -- both branches
            // must define the same dwords, or the compiler unpacks the vector loads to match this one)
            if constexpr (std::is_same<T, int16_t>::value) {
#pragma unroll
                for (int k = 0; k < BT; k += 2) {
                    const uint32_t lo = (k < nvalid) ? (uint32_t)(uint16_t)rc.base[pos0 + k] : 0u;
                    const uint32_t hi = (k + 1 < nvalid) ? (uint32_t)(uint16_t)rc.base[pos0 + k + 1] : 0u;
                    buf.w[k / 2] = lo | (hi << 16);
                }
            } else {
#pragma unroll
                for (int k = 0; k < BT; ++k) buf.w[k] = (k < nvalid) ? __float_as_uint(rc.base[pos0 + k]) : 0u;
            }
        }
    };
    TileRegs<T> nbuf;
    uint32_t nraw, nkeep, nextra;
    int nnvalid;
    load_tile(walk0, nbuf, nraw, nkeep, nextra, nnvalid);
    for (int64_t tb = walk0; tb < bit_hi; tb += 64 * BT) {
        const TileRegs<T> buf = nbuf;
        const uint32_t bits = (nraw & nkeep) | nextra;
        const int nvalid = nnvalid;
        if (tb + 64 * BT < bit_hi) load_tile(tb + 64 * BT, nbuf, nraw, nkeep, nextra, nnvalid);
        const int cnt = __popc(bits);
        const int incl = wave_incl_scan_i(cnt);
        const int excl = incl - cnt;
        const int total = wave_last_i(incl);
        const bool full = tb + 64 * BT <= n;
        if constexpr (std::is_same<T, int16_t>::value) {
            // raw extremes (samples behind the read's end repeat a valid one)
            sgk_s2 w[BT / 2];
            __builtin_memcpy(w, buf.w, sizeof(w));
            if (!full) {
                const sgk_s2 first = {(short)rc.base[0], (short)rc.base[0]};
#pragma unroll
                for (int k = 0; k < BT / 2; ++k) {
                    if (2 * k + 1 >= nvalid) w[k] = (2 * k >= nvalid) ? first : sgk_s2{w[k].x, w[k].x};
                }
            }
#pragma unroll
            for (int k = 0; k < BT / 2; ++k) {
                rmin2 = __builtin_elementwise_min(rmin2, w[k]);
                rmax2 = __builtin_elementwise_max(rmax2, w[k]);
            }
        }
        // walk: lane-relative prefix sums, boundary records.  A tile with more than BREC boundaries is not recorded:
        // its read is redone by the fallback.
        double S = 0.0, S2 = 0.0;
        if (total > BREC) dense = true;
        const uint32_t wbits = total > BREC ? 0u : bits;
        if (full) build_walk<T, true>(buf, wbits, nvalid, rc.sc, l, excl, L, S, S2, mnb, mxb);
        else build_walk<T, false>(buf, wbits, nvalid, rc.sc, l, excl, L, S, S2, mnb, mxb);
        const double inS = wave_incl_scan_d(S), inS2 = wave_incl_scan_d(S2);
        L->pt[l] = inS - S;
        L->pt2[l] = inS2 - S2;
        const double tileS = wave_last_d(inS), tileS2 = wave_last_d(inS2);
        __syncthreads();
        // The next tile's samples and bitmap word were requested before the walk and have long arrived: say so HERE,
        // in front of the rounds' event stores.  Left to the compiler the wait sits at their first use -- behind those
        // stores, and vmcnt counts loads and stores in one queue on gfx9: every tile would wait for its events to be
        // acknowledged by memory.  (vmcnt(0), expcnt / lgkmcnt untouched)
        __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef SGK_DEV
        const int tot = (total > BREC || (a.dev & SGK_DEV_NO_ROUNDS)) ? 0 : total;
#else
        const int tot = total > BREC ? 0 : total;
#endif
        // one event per lane per round.  Prefix sums are kept relative to the tile start (exact under the guard, so
        // no absolute base is needed); the previous boundary of lane l is lane l-1's record, lane 0 takes the
        // carry: the last record of the previous round / tile.
        // The LDS look-ups of a round are two dependent trips (position -> lane -> that lane's prefix); they run two
        // rounds ahead of the arithmetic (round 5: the rounds were 0.41 ms for 7.3 vector instructions per sample --
        // waits, profiles/r05_event_instruction_table.md): records of round i + 2 and lane prefixes of round i + 1 are
        // in flight while round i is evaluated.
        auto rec_at = [&](int k0, uint32_t &pr, double &S_, double &S2_) {
            const int k = k0 + l;
            const int kk = k < tot ? k : tot - 1;
            pr = L->p[kk];
            const BuildRec rcd = L->rec[kk];
            S_ = rcd.S;
            S2_ = rcd.S2;
        };
        uint32_t pr0 = 0u, pr1 = 0u, pr2 = 0u;
        double S0 = 0.0, S20 = 0.0, S1 = 0.0, S21 = 0.0, Sn = 0.0, S2n = 0.0, pt0 = 0.0, pt20 = 0.0, pt1 = 0.0, pt21 = 0.0;
        if (tot > 0) {
            rec_at(0, pr0, S0, S20);
            rec_at(64, pr1, S1, S21);
            pt0 = L->pt[pr0 / BT];
            pt20 = L->pt2[pr0 / BT];
        }
        for (int k0 = 0; k0 < tot; k0 += 64) {
            pt1 = L->pt[pr1 / BT];            // round k0 + 64
            pt21 = L->pt2[pr1 / BT];
            rec_at(k0 + 128, pr2, Sn, S2n);   // round k0 + 128
            const int k = k0 + l;
            const bool act = k < tot;
            const uint32_t p = (uint32_t)tb + pr0;
            const double G = pt0 + S0;
            const double G2 = pt20 + S20;
            const uint32_t pp = (uint32_t)wave_shr1_i((int)p, (int)prevp);
            const double Gp = wave_shr1_d(G, Gprev), G2p = wave_shr1_d(G2, G2prev);
            if (act && (!SEG || rank + (uint32_t)k != skip_rank))
                store_event_fast(eo, rank + (uint32_t)k, pp, p, G - Gp, G2 - G2p, overflow);
            const int last = (tot - k0) < 64 ? (tot - k0 - 1) : 63;  // wave-uniform
            prevp = (uint32_t)__builtin_amdgcn_readlane((int)p, last);
            Gprev = readlane_d(G, last);
            G2prev = readlane_d(G2, last);
            pr0 = pr1; S0 = S1; S20 = S21; pt0 = pt1; pt20 = pt21;
            pr1 = pr2; S1 = Sn; S21 = S2n;
        }
        rank += (uint32_t)tot;
        // rebase the carry to the next tile's start
        Gprev = Gprev - tileS;
        G2prev = G2prev - tileS2;
        __syncthreads();
    }
    // exactness guard (see the file header): reads that fail it are redone by k_event_fallback
    float mn, mx;
    bool known = true;
    if constexpr (std::is_same<T, int16_t>::value) {
        int rmn = rmin2.x < rmin2.y ? rmin2.x : rmin2.y, rmxv = rmax2.x > rmax2.y ? rmax2.x : rmax2.y;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o1 = __shfl_xor(rmn, d, 64), o2 = __shfl_xor(rmxv, d, 64);
            rmn = o1 < rmn ? o1 : rmn;
            rmxv = o2 > rmxv ? o2 : rmxv;
        }
        if constexpr (SEG) { mnb = (uint32_t)rmn; mxb = (uint32_t)rmxv; }
        else known = raw_extremes_to_pa(rmn, rmxv, rc.sc, mn, mx);
    } else {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o1 = (uint32_t)__shfl_xor((int)mnb, d, 64), o2 = (uint32_t)__shfl_xor((int)mxb, d, 64);
            mnb = o1 < mnb ? o1 : mnb;
            mxb = o2 > mxb ? o2 : mxb;
        }
        mn = (mnb == 0xffffffffu) ? FLT_MAX : __uint_as_float(mnb + 1u);
        mx = __uint_as_float(mxb);
        known = mxb < 0x7f800000u;
    }
    if constexpr (SEG) {
        // the read's last segment closes the read's last event; the verdict on the read is chain_segment's
        if (l == 0 && seg_b == n) store_event_fast(eo, rank, prevp, (uint32_t)n, 0.0 - Gprev, 0.0 - G2prev, overflow);
        const bool ovf = __any(overflow);
        if (l == 0) {
            st->ext_lo = mnb;
            st->ext_hi = mxb;
            st->bflags = (dense ? 1u : 0u) | (ovf ? 2u : 0u);
        }
        return;
    }
    const bool flagged = dense || !known || !guard_ok(mn, mx, n) || declined;
    if (l == 0) {
        a.flags[r] = flagged ? 1 : 0;
        if (flagged) {
            const uint32_t k = atomicAdd(&a.hdr->n_flagged, 1u);
            a.flag_list[k] = r;
        } else {
            store_event_fast(eo, rank, prevp, (uint32_t)n, 0.0 - Gprev, 0.0 - G2prev, overflow);
            a.n_events[r] = rank + 1;
            atomicAdd(&a.hdr->n_events_total, (unsigned long long)(rank + 1));
        }
    }
    if (!flagged && __any(overflow) && l == 0) atomicAdd(&a.hdr->n_overflow, 1u);
}

// fallback builder: event sums are differences of the sequential prefix arrays, as in the reference
template <typename T>
__device__ void build_read_prefix(const EvArgs &a, const ReadCtx<T> &rc, uint32_t r) {
    const int64_t n = rc.n;
    const int l = lane_id();
    const uint64_t slot0 = a.ev_slots[r], cap = a.ev_slots[r + 1] - slot0;
    if (n <= 0) {
        if (l == 0) a.n_events[r] = 0;
        return;
    }
    const uint32_t *bm32 = reinterpret_cast<const uint32_t *>(rc.bm);
    bool overflow = false;
    uint32_t rank = 0, prevp = 0;
    const int64_t nwords = (n + 31) >> 5;
    for (int64_t w0 = 0; w0 < nwords; w0 += 64) {
        const int64_t w = w0 + l;
        const int64_t pos0 = w * 32;
        uint32_t bits = (w < nwords) ? bm32[w] : 0u;
        if (w < nwords && n - pos0 < 32) bits &= (1u << (int)(n - pos0)) - 1u;
        const int cnt = __popc(bits);
        const int incl = wave_incl_scan_i(cnt);
        const int excl = incl - cnt;
        const int total = __shfl(incl, 63, 64);
        const unsigned long long m = __ballot(cnt > 0);
        const uint32_t lastp = cnt > 0 ? (uint32_t)(pos0 + 31 - __clz((int)bits)) : 0u;
        const unsigned long long lower = m & ((1ull << l) - 1ull);
        const int src = lower ? 63 - __clzll((long long)lower) : 0;
        uint32_t pl = __shfl(lastp, src, 64);
        if (!lower) pl = prevp;
        int k = 0;
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const uint32_t p = (uint32_t)(pos0 + b);
            store_event(a, slot0, cap, (uint64_t)rank + (uint64_t)(excl + k), pl, p, rc.P[p] - rc.P[pl],
                        rc.P2[p] - rc.P2[pl], overflow);
            pl = p;
            ++k;
        }
        if (m) {
            prevp = __shfl(lastp, 63 - __clzll((long long)m), 64);
            rank += (uint32_t)total;
        }
    }
    if (l == 0) {
        store_event(a, slot0, cap, (uint64_t)rank, prevp, (uint32_t)n, rc.P[n] - rc.P[prevp],
                    rc.P2[n] - rc.P2[prevp], overflow);
        a.n_events[r] = rank + 1;
        atomicAdd(&a.hdr->n_events_total, (unsigned long long)(rank + 1));
    }
    if (__any(overflow) && l == 0) atomicAdd(&a.hdr->n_overflow, 1u);
}

// Sequential double prefix sums, src/events.c:293-303: strictly in order.  Per 2048-sample tile the wave
// converts to pA (and float squares) in parallel into LDS; lane 0 runs the dependent chain of sums and
// lane 1 the chain of squares, writing the prefix values to LDS; then all lanes store the tile to the
// scratch arrays (coalesced) and test every addition for exactness (TwoSum residual): positions where
// the scan rounded are the "events" the repair logic needs.
constexpr int SP_TILE = 2048;
struct PrefixLds {
    float x[SP_TILE];
    float xq[SP_TILE];
    double ps[SP_TILE + 1];   // ps[0] = prefix before the tile, ps[k+1] = prefix after sample k
    double pq[SP_TILE + 1];
};
struct EventList {
    int ev[REP_MAX_EVENTS];
    int count;
};
template <typename T>
__device__ void seq_prefix(const ReadCtx<T> &rc, double *P, double *P2, PrefixLds *L, EventList *E) {
    const int l = lane_id();
    const int64_t n = rc.n;
    double acc = 0.0;
    if (l == 0) { P[0] = 0.0; P2[0] = 0.0; E->count = 0; }
    for (int64_t tb = 0; tb < n; tb += SP_TILE) {
        const int m = (n - tb) < SP_TILE ? (int)(n - tb) : SP_TILE;
        __syncthreads();
        for (int k = l; k < m; k += 64) {
            const float x = to_pa(rc.base[tb + k], rc.sc);
            L->x[k] = x;
            L->xq[k] = x * x;
        }
        __syncthreads();
        if (l < 2) {
            const float *src = (l == 0) ? L->x : L->xq;
            double *dst = (l == 0) ? L->ps : L->pq;
            dst[0] = acc;
            int k = 0;
            for (; k + 8 <= m; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = acc + (double)v[u];
                    dst[k + u + 1] = acc;
                }
            }
            for (; k < m; ++k) {
                acc = acc + (double)src[k];
                dst[k + 1] = acc;
            }
        }
        __syncthreads();
        for (int k = l; k < m; k += 64) {
            const double s0 = L->ps[k], s1 = L->ps[k + 1], q0 = L->pq[k], q1 = L->pq[k + 1];
            P[tb + k + 1] = s1;
            P2[tb + k + 1] = q1;
            const double ys = (double)L->x[k], yq = (double)L->xq[k];
            const double bs = s1 - s0, bq = q1 - q0;
            const double es = (s0 - (s1 - bs)) + (ys - bs), eq = (q0 - (q1 - bq)) + (yq - bq);
            if (es != 0.0 || eq != 0.0) {
                const int idx = atomicAdd(&E->count, 1);
                if (idx < REP_MAX_EVENTS) E->ev[idx] = (int)(tb + k);
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (l == 0) {  // sort the (few) event positions
        const int m = E->count < REP_MAX_EVENTS ? E->count : REP_MAX_EVENTS;
        for (int i = 1; i < m; ++i) {
            const int v = E->ev[i];
            int j = i - 1;
            while (j >= 0 && E->ev[j] > v) { E->ev[j + 1] = E->ev[j]; --j; }
            E->ev[j + 1] = v;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------- kernels

// DNA preset: 168 VGPRs -> 3 waves per SIMD; RNA preset (deeper rings): 242 VGPRs -> 2
#ifndef SGK_DET_WAVES_DNA
#define SGK_DET_WAVES_DNA 3
#endif
#ifndef SGK_DET_WAVES_RNA
#define SGK_DET_WAVES_RNA 2
#endif

// Detector and builder of one read in one wave, back to back: the builder's phases
// that wait on memory (sample tiles, event stores) run under other waves' detector arithmetic instead of in a
// kernel of their own.  The bitmap goes through memory (L2) between the two phases of the same wave.
union EventLds {
    LzLds lz;
    BuildLds b;
};
// span of segment g of a read of n samples
__device__ __forceinline__ void seg_span(uint32_t seg_len, uint32_t g, int64_t n, int &sa, int &sb) {
    const int64_t lo = (int64_t)g * seg_len, hi = lo + seg_len;
    sa = (int)lo;
    sb = (int)(hi < n ? hi : n);
}

// Which reads several wavefronts share, and in segments of which length (0: the read has a wavefront of its own).
//  * long reads (>= long_min samples): a wave per read cannot end before its longest read has;
//  * the TAIL SPLIT (round 4): the reads at dispatch positions >= split_from.  n_reads equal waves over the GPU's
//    resident wave slots run in rounds; the last, partial round costs nearly a whole one (10 000 reads over 3 072 slots:
//    3.26 rounds of work took the time of 3.75).  The reads of that round are cut into split_seg-sample segments -- as
//    many units as fill a round, each a fraction of a read long -- and run FIRST; every other read keeps the fused
//    detector + builder of its own wave (cutting every read costs more than the balance returns: the builder of a
//    cut read is a kernel of its own, profiles/archive/r04_event_experiments.md).
__device__ __forceinline__ uint32_t seg_len_of(const EvArgs &a, uint32_t pos, uint32_t n) {
    if (a.max_segs == 0) return 0u;
    if (n >= a.long_min) return a.seg_len;
    if (pos >= a.split_from && n > a.split_seg) return a.split_seg;
    return 0u;
}
// The list of their segments (one thread per dispatch position; the order of the list does not matter).
__global__ __launch_bounds__(256) void k_seg_plan(EvArgs a) {
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= a.n_reads) return;
    const uint32_t r = a.order ? a.order[pos] : pos;
    const uint32_t n = a.lengths[r];
    const uint32_t seg = seg_len_of(a, pos, n);
    if (seg == 0u) return;
    const uint32_t G = (uint32_t)(((uint64_t)n + seg - 1) / seg);
    const uint32_t s0 = atomicAdd(&a.hdr->n_segs, G), li = atomicAdd(&a.hdr->n_long, 1u);
    // The capacities cover every batch of non-overlapping reads with these totals (event_seg_capacity).  A read that
    // does not fit all the same (overlapping reads) is left to the exact fallback; what it took of the lists is marked
    // as nobody's.
    if ((uint64_t)s0 + G > a.max_segs || li >= a.max_long) {
        for (uint64_t k = s0; k < (uint64_t)s0 + G && k < a.max_segs; ++k) a.segs[k].read = SEG_NONE;
        if (li < a.max_long) {
            LongRead none;
            none.read = r; none.seg0 = 0; none.nseg = 0; none.seg_len = seg;
            none.ext_lo = 0; none.ext_hi = 0; none.flags = 0; none.built = 0;
            a.longs[li] = none;
        }
        a.flags[r] = 1;
        a.flag_list[atomicAdd(&a.hdr->n_flagged, 1u)] = r;
        return;
    }
    LongRead lr;
    lr.read = r; lr.seg0 = s0; lr.nseg = G; lr.seg_len = seg;
    lr.ext_lo = a.dig ? 32767u : 0xffffffffu;          // (int16 input: signed extremes; pA input: bit patterns)
    lr.ext_hi = a.dig ? (uint32_t)-32768 : 0u;
    lr.flags = 0; lr.built = 0;
    a.longs[li] = lr;
    for (uint32_t g = 0; g < G; ++g) {
        SegDesc d;
        d.read = r; d.g = g; d.lread = li; d.pad = 0;
        a.segs[s0 + g] = d;
        a.seg_state[s0 + g].stage = 0u;   // (the chain: nothing of this segment is published yet)
    }
}

template <int W1, typename T>
__global__ __launch_bounds__(64, (W1 == 3 ? SGK_DET_WAVES_DNA : SGK_DET_WAVES_RNA)) void k_event(EvArgs a) {
    __shared__ EventLds L;
    // One read per workgroup, longest first (launch_order): a kernel cannot end before its longest read has, so that one
    // should start first, not wherever it sits in the batch.  Reads that several waves share -- long reads, the tail
    // split -- are k_event_seg's (round 5: a kernel of its own.  While one kernel held both paths the whole-read path --
    // 9 216 of config 2's 10 000 workgroups -- carried the chain's state: 33 spilled registers, 688 bytes of scratch per
    // lane; as a device function called from here the chain spilled in its own detector pass instead.)
    const uint32_t bi = blockIdx.x;
    const uint32_t r = a.order ? a.order[bi] : bi;
    const ReadCtx<T> rc = make_ctx<T>(a, r);
    if (seg_len_of(a, bi, (uint32_t)rc.n) != 0u) return;  // taken by its segments (k_event_seg)
    if (a.multi_lanes && rc.n < (int64_t)a.multi_max) return;  // taken by k_event_multi
#ifdef SGK_DEV  // development builds (tools/build_variant.sh): phases switched off / timestamps, see event_args.h
    unsigned long long t0 = 0ull, t1 = 0ull;
    if (a.dev & SGK_DEV_TRACE) t0 = wall_clock64();
    int rcode = 0;
    if (!(a.dev & SGK_DEV_NO_DETECT)) rcode = detect_span<W1, T, false>(rc, a.hdr, &L.lz, nullptr, 0, (int)rc.n, 0, a.lead_override, nullptr);
    if (a.dev & SGK_DEV_TRACE) t1 = wall_clock64();
#else
    const int rcode = detect_span<W1, T, false>(rc, a.hdr, &L.lz, nullptr, 0, (int)rc.n, 0, a.lead_override, nullptr);
#endif
    // the bitmap words of every lane (and the replay's atomics) are complete before any lane of this workgroup reads
    // them back.  Workgroup scope: the wave's own CU only -- an agent-scope release / acquire pair here writes back and
    // invalidates L2 once per read, which made 5 000-sample reads 1.7x slower than with two kernels.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef SGK_DEV
    if (!(a.dev & SGK_DEV_NO_BUILD)) build_read<T>(a, rc, r, &L.b, rcode != 0);
    if ((a.dev & SGK_DEV_TRACE) && lane_id() == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(a.scratch) + 4ull * blockIdx.x;
        tr[0] = t0;
        tr[1] = t1;
        tr[2] = wall_clock64();
        tr[3] = ((unsigned long long)xcc << 32) | hw;
    }
#else
    build_read<T>(a, rc, r, &L.b, rcode != 0);
#endif
}

// ---- the chain: detector, seam check and builder of one segment in one wave (round 4) -------------------------
// Round 3 ran the segments' detector passes in k_event and left the rest to four kernels behind it (seams, counts,
// builders, verdict): a cut read lost the fusion of detector and builder, and every kernel had a tail of its own.  Now
// the wave of segment g does everything itself and takes what it needs from segment g - 1 -- ALWAYS a lower workgroup
// index, so with workgroups started in index order (what the hardware does, per XCD; decoupled look-back scans rely on
// the same) the chain cannot deadlock; a wait that exceeds ~1 s all the same declines the read (exact fallback).
//   1. detect_span over the segment, speculative start (mode 1): bitmap words of ITS range only;
//   2. wait for segment g - 1's record {final end state, boundaries so far, the last of them, declined?} -- 40 bytes,
//      published with agent-scope atomic stores (write-through) behind an s_waitcnt, read back with agent-scope atomic
//      loads behind one agent acquire (MI355X_MICROARCH.md, inter-workgroup visibility: the XCDs' L2s are not coherent);
//   3. the seam: end(g - 1) != its own init0 -> the segment is run again from the true state (mode 2);
//   4. publish its own record (so the segments behind need not wait for its builder);
//   5. build the events that end at the boundaries it OWNS: those inside its range and the peaks that were pending at
//      the seam and emitted behind it (seg->pre: positions in front of the segment that no bitmap shows) -- from the
//      last boundary of the segments in front, at the rank their boundaries give;
//   6. extremes / flags into the read's record (atomics); the wave that finishes last gives the verdict (exactness
//      guard over the whole read, n_events, counters, fallback list).
// A hot long-detector run that crosses the seam is replayed for the part in front of it as well; a peak found there
// (none on nanopore-like signals) would belong to another wave's events: the read is declined.
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (ONE call site of detect_span for the speculative pass and the re-run from the true state: each inlined copy of the
// detector pass is ~25 KB of code.)
template <int W1, typename T>
__device__ __forceinline__ void chain_segment(const EvArgs &a, uint32_t bx, EventLds *L) {
    if (bx >= a.hdr->n_segs) return;
    const SegDesc d = a.segs[bx];
    if (d.read == SEG_NONE) return;
    const uint32_t r = d.read, g = d.g, lread = d.lread;
    const ReadCtx<T> rc = make_ctx<T>(a, r);
    LongRead *lrp = a.longs + lread;
    const uint32_t seg_len = lrp->seg_len;
    int sa, sb;
    seg_span(seg_len, g, rc.n, sa, sb);
    SegState *st = a.seg_state + bx;
    const int l = lane_id();
    const uint32_t nseg = lrp->nseg;
    bool declined = false;
    uint32_t prev_cum = 0u;
    int prev_last = -1;
    int mode = g == 0 ? 0 : 1;
#ifdef SGK_DEV
    unsigned long long tt0 = 0ull, tt1 = 0ull, tt2 = 0ull;
    if (a.dev & SGK_DEV_TRACE) tt0 = wall_clock64();
#endif
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int rcode = detect_span<W1, T, false>(rc, a.hdr, &L->lz, nullptr, sa, sb, mode, a.lead_override, st);
#ifdef SGK_DEV
        if ((a.dev & SGK_DEV_TRACE) && attempt == 0) tt1 = wall_clock64();
#endif
        if (l == 0) st->status = rcode;
        declined = rcode != 0;
        // (what detect_span's lanes stored -- st->end, cross runs, pre peaks -- is visible to the wave's other lanes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (g == 0 || attempt == 1) break;
        SegState *ps = st - 1;
        int ok = 1;
        if (l == 0) {
            unsigned spins = 0;
            while (ld_agent(&ps->stage) == 0u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > 4000000u) { ok = 0; break; }
            }
        }
        ok = __builtin_amdgcn_readfirstlane(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LzSnapState pe;
        pe.sp = (int)ld_agent(reinterpret_cast<const uint32_t *>(&ps->end.sp));
        pe.sv = __uint_as_float(ld_agent(reinterpret_cast<const uint32_t *>(&ps->end.sv)));
        pe.lm = (int)ld_agent(reinterpret_cast<const uint32_t *>(&ps->end.lm));
        pe.r0 = (int)ld_agent(reinterpret_cast<const uint32_t *>(&ps->end.r0));
        pe.bits = ld_agent(&ps->end.bits);
        prev_cum = ld_agent(&ps->cum_cnt);
        prev_last = (int)ld_agent(reinterpret_cast<const uint32_t *>(&ps->last_pos));
        if (!ok || (ld_agent(&ps->cflags) & 1u)) declined = true;
        if (declined) break;
        // the seam (st->init0 / st->end: this wave's own stores; the states are in LDS as well)
        const LzSnapState mine = L->lz.snap.init[0];
        if (lz_equal(pe, mine)) break;
        __syncthreads();
        if (l == 0) {
            L->lz.snap.st0[0] = pe;
            atomicAdd(&a.hdr->n_seam_rerun, 1u);
        }
        __syncthreads();
        mode = 2;   // once more, from the true state
    }
    if (g > 0 && !declined) {
        // hot runs that began in front of the seam: the part in front of it
        __syncthreads();
        const int nc = (int)st->n_cross;
        if (nc > 0) {
            const bool has = l < nc;
            const LzRun run = has ? st->cross[l] : LzRun{0, 0};
            int found = 0;
            replay_run<W1, T, false>(rc, nullptr, has, run.a, run.b, 0, sa, a.hdr, &found);
            if (__any(found != 0)) declined = true;
        }
    }
    // this wave's bitmap words (and the replay's atomics) are complete before any of its lanes reads them back
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // the boundaries the segment owns: the bits of its range + its pre peaks
    int cnt = 0, last = -1, n_pre = 0;
    if (!declined) {
        const int w0 = sa >> 6, w1 = (sb + 63) >> 6;
        for (int wb = w0; wb < w1; wb += 64) {
            const int w = wb + l;
            unsigned long long v = w < w1 ? rc.bm[w] : 0ull;
            const int rem = sb - (w << 6);
            if (rem < 64) v = rem <= 0 ? 0ull : (v & ((1ull << rem) - 1ull));
            if (v) {
                cnt += __popcll(v);
                last = (w << 6) + 63 - __clzll(v);
            }
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            cnt += __shfl_xor(cnt, dd, 64);
            const int o = __shfl_xor(last, dd, 64);
            last = o > last ? o : last;
        }
        n_pre = (int)st->n_pre;
        if (last < 0) {
            for (int k = 0; k < n_pre; ++k) last = st->pre[k] > last ? st->pre[k] : last;
            if (last < 0) last = prev_last;
        }
    }
    const uint32_t my_cum = prev_cum + (uint32_t)(cnt + n_pre);
    // publish (the end state: what detect_span left in st->end -- lane 0's own store)
    if (l == 0) {
        const LzSnapState e = st->end;
        st_agent(reinterpret_cast<uint32_t *>(&st->end.sp), (uint32_t)e.sp);
        st_agent(reinterpret_cast<uint32_t *>(&st->end.sv), __float_as_uint(e.sv));
        st_agent(reinterpret_cast<uint32_t *>(&st->end.lm), (uint32_t)e.lm);
        st_agent(reinterpret_cast<uint32_t *>(&st->end.r0), (uint32_t)e.r0);
        st_agent(&st->end.bits, e.bits);
        st_agent(&st->cum_cnt, my_cum);
        st_agent(reinterpret_cast<uint32_t *>(&st->last_pos), (uint32_t)(declined ? prev_last : last));
        st_agent(&st->cflags, declined ? 1u : 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(&st->stage, 1u);
    }
    uint32_t fl = declined ? 1u : 0u;
#ifdef SGK_DEV
    if (a.dev & SGK_DEV_TRACE) tt2 = wall_clock64();
#endif
    if (!declined) {
        build_read<T, true>(a, rc, r, &L->b, false, sa, sb, st, prev_cum, prev_last, st->pre, n_pre);
        __syncthreads();
#ifdef SGK_DEV
        if ((a.dev & SGK_DEV_TRACE) && l == 0) {   // (segments: start, detector end, seam + publish end | builder end in the top bits)
            unsigned long long *tr = reinterpret_cast<unsigned long long *>(a.scratch) + 4ull * (a.n_reads + blockIdx.x);
            tr[0] = tt0;
            tr[1] = tt1;
            tr[2] = wall_clock64();
            tr[3] = (1ull << 63) | ((tt2 - tt0) << 16) | (g & 0xffffu);
        }
#endif
        if (l == 0) {
            const uint32_t lo = st->ext_lo, hi = st->ext_hi;
            if constexpr (std::is_same<T, int16_t>::value) {
                atomicMin(reinterpret_cast<int *>(&lrp->ext_lo), (int)lo);
                atomicMax(reinterpret_cast<int *>(&lrp->ext_hi), (int)hi);
            } else {
                atomicMin(&lrp->ext_lo, lo);
                atomicMax(&lrp->ext_hi, hi);
            }
            fl = st->bflags;
        }
    }
    if (l != 0) return;
    if (fl) atomicOr(&lrp->flags, fl);
    // (everything the verdict reads was written with device-scope atomics, which complete in memory: this lane's have
    // before it counts itself done.  NO agent-scope fence: its write-back of the XCD's whole L2 -- megabytes of other
    // waves' event stores -- once per segment made the chain slower than no split at all, 3.95 vs 3.86 ms)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t done = atomicAdd(&lrp->built, 1u);
    if (done + 1u != nseg) return;
    // the verdict (every segment has published its record and added its extremes / flags)
    const uint32_t flags = atomicOr(&lrp->flags, 0u);
    const int64_t n = rc.n;
    bool flagged = (flags & 1u) != 0u;
    if (!flagged) {
        float mn, mx;
        bool known;
        if constexpr (std::is_same<T, int16_t>::value) {
            const int rmn = atomicMin(reinterpret_cast<int *>(&lrp->ext_lo), 32767);
            const int rmx = atomicMax(reinterpret_cast<int *>(&lrp->ext_hi), -32768);
            known = raw_extremes_to_pa(rmn, rmx, rc.sc, mn, mx);
        } else {
            const uint32_t mnb = atomicMin(&lrp->ext_lo, 0xffffffffu), mxb = atomicMax(&lrp->ext_hi, 0u);
            mn = (mnb == 0xffffffffu) ? FLT_MAX : __uint_as_float(mnb + 1u);
            mx = __uint_as_float(mxb);
            known = mxb < 0x7f800000u;
        }
        flagged = !known || !guard_ok(mn, mx, n);
    }
    a.flags[r] = flagged ? 1 : 0;
    if (flagged) {
        a.flag_list[atomicAdd(&a.hdr->n_flagged, 1u)] = r;
    } else {
        const uint32_t nev = ld_agent(&a.seg_state[lrp->seg0 + nseg - 1u].cum_cnt) + 1u;
        a.n_events[r] = nev;
        atomicAdd(&a.hdr->n_events_total, (unsigned long long)nev);
        if (flags & 2u) atomicAdd(&a.hdr->n_overflow, 1u);
    }
}

// The segments' kernel: a workgroup per entry of the segment list (usually much shorter than its capacity).
template <int W1, typename T>
__global__ __launch_bounds__(64, (W1 == 3 ? SGK_DET_WAVES_DNA : SGK_DET_WAVES_RNA)) void k_event_seg(EvArgs a) {
    __shared__ EventLds L;
    chain_segment<W1, T>(a, blockIdx.x, &L);
}

// Short reads, several per wavefront: `lanes` consecutive lanes share a read (detector: detect_span<MULTI>), then the
// wave builds its reads one after the other.  The short reads are the tail of the dispatch order (launch_order sorts by
// length class, longest first; multi_max is a class boundary) or, in a batch without an order, all reads.
template <int W1, typename T>
__global__ __launch_bounds__(64, (W1 == 3 ? SGK_DET_WAVES_DNA : SGK_DET_WAVES_RNA)) void k_event_multi(EvArgs a) {
    __shared__ EventLds L;
    const int lanes = (int)a.multi_lanes, G = 64 / lanes;
    const int l = lane_id();
    const uint32_t first = a.order ? a.order[a.n_reads + len_bucket(a.multi_max)] : 0u;  // reads that are not short
    const uint32_t nshort = a.n_reads - first;
    const uint32_t w0 = blockIdx.x * (uint32_t)G;
    if (w0 >= nshort) return;
    const uint32_t gi = (uint32_t)l / (uint32_t)lanes;
    const bool has = w0 + gi < nshort;
    const uint32_t idx = first + (has ? w0 + gi : w0);
    const uint32_t r = a.order ? a.order[idx] : idx;
    ReadCtx<T> rc = make_ctx<T>(a, r);
    // (with segments shorter than multi_max -- tests -- a read can be short and long at once: the segments have it)
    const bool mine = has && !(a.max_segs && rc.n >= (int64_t)a.long_min);  // (no tail split in a batch with packed reads)
    if (!mine) rc.n = 0;
    const int rcode = detect_span<W1, T, false, true>(rc, a.hdr, &L.lz, nullptr, 0, (int)rc.n, 0, a.lead_override, nullptr,
                                                      lanes);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int g = 0; g < G; ++g) {
        if (w0 + (uint32_t)g >= nshort) break;
        const uint32_t rg = (uint32_t)__builtin_amdgcn_readlane((int)r, g * lanes);
        const int code = __builtin_amdgcn_readlane(rcode, g * lanes);
        if (!__builtin_amdgcn_readlane((int)mine, g * lanes)) continue;
        const ReadCtx<T> rcg = make_ctx<T>(a, rg);
        build_read<T>(a, rcg, rg, &L.b, code != 0);
        __syncthreads();
    }
}

template <int W1, typename T>
__global__ __launch_bounds__(64) void k_event_fallback(EvArgs a) {
    __shared__ PrefixLds L;
    __shared__ LzLds Lz;
    __shared__ EventList events;
    double *P = a.scratch + (uint64_t)blockIdx.x * a.scratch_stride;
    double *P2 = P + a.scratch_stride / 2;
    const uint32_t nf = a.hdr->n_flagged;
    for (;;) {
        uint32_t w = 0;
        if (lane_id() == 0) w = atomicAdd(&a.hdr->fb_next, 1u);
        w = __shfl(w, 0, 64);
        if (w >= nf) break;
        const uint32_t r = a.flag_list[w];
        ReadCtx<T> rc = make_ctx<T>(a, r);
        seq_prefix<T>(rc, P, P2, &L, &events);
        rc.P = P;
        rc.P2 = P2;
        // fast pass + event-local repair; reads the fast pass cannot take (odd alignment, no room around
        // the read) go through the generic pass that takes every window sum from the prefix arrays
        RepairCtx rep;
        rep.P = P;
        rep.P2 = P2;
        rep.ev = events.ev;
        rep.nev = events.count < REP_MAX_EVENTS ? events.count : REP_MAX_EVENTS;
        rep.all_dirty = events.count > REP_MAX_EVENTS;
        const int rcode = detect_read_lazy<W1, T, true>(rc, a.hdr, &Lz, &rep);
        if (rcode) detect_read<W1, T>(rc, a.hdr);
        __threadfence();
        __syncthreads();
        build_read_prefix<T>(a, rc, r);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- launcher

// Side streams.  A batch with short AND longer reads runs two detector kernels: k_event (a wavefront per read, and the
// segments of the long reads) and k_event_multi (several short reads per wavefront).  In one stream the second would
// wait for the last wave of the first -- two tails instead of one, which costs what the packing gains (50 000 RNA-like
// reads, log-normal around 20 000 samples: 6.36 ms against 6.29 ms with a wavefront per read).  k_event_multi therefore
// goes to a side stream of the library's own, forked off the caller's stream behind the dispatch order and joined in
// front of the fallback kernel; the long reads' seam / builder kernels overlap with it as well.
// (Tried for the tail split's segments: a stream of the LOWEST priority, so that they would be dispatched into the slots
// k_event's last waves leave empty -- 5.1 vs 3.8 ms on config 2: its kernels start late and slowly.)
// A small pool per device and kind, handed out round-robin; a stream's mutex is held while one launch enqueues its
// fork .. join on it (the events are the stream's), never across launches of other streams or devices.
constexpr int SIDE_POOL = 4;
static SideStream g_side[64][3][SIDE_POOL];
static std::atomic<unsigned> g_side_next[64][3];
static SideStream *side_acquire_slot(int dev, int kind, int slot, int priority);
// returns a locked side stream (unlock with x->mu.unlock()), or null
SideStream *side_acquire(int priority) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    const int kind = priority < 0 ? 1 : (priority > 0 ? 2 : 0);
    // the whole pool of this device and kind is made by the first call that needs one (stream creation takes
    // milliseconds: made one by one, the first SIDE_POOL launches each paid for one -- a two-step warm-up was not enough)
    static std::atomic<bool> g_side_made[64][3];
    if (!g_side_made[dev][kind].exchange(true))
        for (int k = 1; k < SIDE_POOL; ++k) {
            SideStream *y = side_acquire_slot(dev, kind, k, priority);
            if (y) y->mu.unlock();
        }
    return side_acquire_slot(dev, kind, (int)(g_side_next[dev][kind].fetch_add(1u) % SIDE_POOL), priority);
}
static SideStream *side_acquire_slot(int dev, int kind, int slot, int priority) {
    SideStream &x = g_side[dev][kind][slot];
    x.mu.lock();
    if (!x.s && !x.tried) {
        x.tried = true;
        hipStream_t s = nullptr;
        hipEvent_t f = nullptr, j = nullptr;
        int lo = 0, hi = 0;
        hipError_t e;
        if (priority != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority < 0 ? lo : hi);   // (lo: the numerically greatest = lowest)
        else
            e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess) {
            if (hipEventCreateWithFlags(&f, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&j, hipEventDisableTiming) == hipSuccess) {
                x.s = s; x.fork = f; x.join = j;
            } else {
                if (f) (void)hipEventDestroy(f);
                (void)hipStreamDestroy(s);
            }
        }
    }
    if (!x.s) {
        x.mu.unlock();
        return nullptr;
    }
    return &x;
}
template <typename T>
static int launch_event_t(const EvArgs &a, int rna, uint32_t n_fb_blocks, hipStream_t st) {
    if (a.n_reads == 0) return SGK_OK;
    // (Tried in round 1: cutting the batch into read slices and running the builder of slice s on a side
    // stream under the detector of slice s+1.  Both kernels contend for VALU issue and the detector needs
    // >= 3072 reads in flight to fill its 12 waves/CU, so the overlapped step was 8.8 ms against 7.8 ms.)
    ProfScope whole("path:event", st);
    SGK_HIP_TRY(hipMemsetAsync(a.hdr, 0, sizeof(EvHeader), st));
    EvArgs ao = a;
    if (a.n_reads >= ORDER_MIN_READS && a.order) {
        const int rc = launch_order(a.lengths, a.n_reads, a.order, a.order + a.n_reads, st);
        if (rc != SGK_OK) return rc;
    } else ao.order = nullptr;
    if (ao.max_segs) hipLaunchKernelGGL(k_seg_plan, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, ao);
    // The segments' kernel.  Long reads' segments start FIRST: they stay on the caller's stream and k_event goes to a
    // side stream whose start waits for the fork event (the other way round k_event's workgroups -- ten thousand of them
    // -- take every slot and the chains start late: a ragged batch took 3.81 ms instead of 3.65; stat's long reads taught
    // the same, stat_kernels.hip launch_beside_long).  The tail split's segments go LAST, behind k_event on the caller's
    // stream: small units for the slots the last whole reads leave empty.
    const bool tail_only = ao.max_segs && ao.split_seg && !ao.has_long;
    auto launch_segs = [&](hipStream_t s_) {
        ProfScope ps("k_event_seg", s_);
        if (rna) hipLaunchKernelGGL((k_event_seg<7, T>), dim3(ao.max_segs), dim3(64), 0, s_, ao);
        else hipLaunchKernelGGL((k_event_seg<3, T>), dim3(ao.max_segs), dim3(64), 0, s_, ao);
    };
    // (no dispatch order and packing on: every read is under multi_max.  k_event would have nothing to do -- unless
    // the segments are test-sized and some of those reads are long: their segments are k_event_seg's)
    const bool all_short = ao.multi_lanes && !ao.order && !ao.max_segs;
    SideFork multi_side, whole_side;
    hipStream_t st_multi = st, st_whole = st;
    if (ao.multi_lanes && !all_short && multi_side.open(0, st)) st_multi = multi_side.stream();
    if (ao.max_segs && !tail_only && !all_short && whole_side.open(0, st)) st_whole = whole_side.stream();
    if (ao.max_segs && !tail_only) {
        launch_segs(st);
        SGK_HIP_TRY(hipGetLastError());
    }
    if (ao.multi_lanes) {
        ProfScope ps("k_event_multi", st_multi);
        const uint32_t per_wave = 64u / ao.multi_lanes;
        const uint32_t grid = (a.n_reads + per_wave - 1) / per_wave;
        if (rna) hipLaunchKernelGGL((k_event_multi<7, T>), dim3(grid), dim3(64), 0, st_multi, ao);
        else hipLaunchKernelGGL((k_event_multi<3, T>), dim3(grid), dim3(64), 0, st_multi, ao);
        SGK_HIP_TRY(hipGetLastError());
    }
    if (!all_short) {
        ProfScope ps("k_event", st_whole);
        if (rna) hipLaunchKernelGGL((k_event<7, T>), dim3(a.n_reads), dim3(64), 0, st_whole, ao);
        else hipLaunchKernelGGL((k_event<3, T>), dim3(a.n_reads), dim3(64), 0, st_whole, ao);
    }
    SGK_HIP_TRY(hipGetLastError());
    if (tail_only) launch_segs(st);
    SGK_HIP_TRY(hipGetLastError());
    // join: the fallback kernel (and whatever the caller enqueues next) waits for the side streams as well
    multi_side.join();
    whole_side.join();
    {
        ProfScope ps("k_event_fallback", st);
        if (rna) hipLaunchKernelGGL((k_event_fallback<7, T>), dim3(n_fb_blocks), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_event_fallback<3, T>), dim3(n_fb_blocks), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_event(const EvArgs &a, int rna, bool float_input, uint32_t n_fb_blocks, hipStream_t st) {
    return float_input ? launch_event_t<float>(a, rna, n_fb_blocks, st)
                       : launch_event_t<int16_t>(a, rna, n_fb_blocks, st);
}

}  // namespace sgk
