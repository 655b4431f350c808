// event_kernels.hip -- the `event` hot path (reference: src/events.c:293-573) for gfx950.
//
// One wavefront (64 lanes) per read.  Three kernels:
//
//   k_event_detect   window sums -> two t-statistics -> short/long peak detector.
//                    Lane c owns chunk c of the read (K samples, K a multiple of 64);
//                    samples are staged global -> LDS in 64-sample row tiles (128-byte row
//                    segments, 16 bytes per lane), each lane slides its four window sums in
//                    registers (double; exact), evaluates the reference's mixed float/double
//                    t-statistic expression tree (events.c:338-361) and steps both detector
//                    automata (events.c:383-440).  The automaton is serial in the reference;
//                    here every chunk starts SPECULATIVELY from the fresh state LEAD samples
//                    before its chunk, and the speculation is verified: chunk c is accepted
//                    iff its state at its chunk start equals chunk c-1's state at that
//                    position; mismatching chunks are re-run from the true state until a
//                    fixed point (exact in the general case; re-runs are counted in the
//                    status block).  Output: one bit per sample (peak positions) in a
//                    workspace bitmap.
//
//   k_event_build    bitmap + samples -> event table (events.c:457-504).  Lane-local double
//                    prefix sums, wave scan across lanes, boundary records compacted in LDS,
//                    then one event per lane per round with coalesced SoA stores of
//                    (start, length, mean, stdv).
//
//   k_event_fallback persistent kernel over the reads that fail the exactness guard: lane 0
//                    reproduces compute_sum_sumsq's sequential double prefix scan
//                    (events.c:293-303) into workspace scratch, then the same detector and
//                    builder run with window/event sums taken as differences of those arrays,
//                    exactly as the reference does.
//
// Exactness guard: the reference accumulates double prefix sums sequentially and uses their
// differences; the fast path forms window sums and event sums directly.  Both give the
// real-number sums (hence identical bits) whenever no prefix sum can round: every sample is a
// multiple of 2^g (g = lowest bit of the smallest non-zero |x|) and all partial sums are below
// 2^(g+53).  Per read we check  ilogb(n*max|x|) - ilogb(min|x|!=0) <= 29  for x and for the
// float squares; reads failing the check take the fallback kernel.
#include <type_traits>
#include <utility>

#include "event_args.h"
#include "row_stream.h"
#include "sgk_common.h"
#include "tstat_math.h"

namespace sgk {

#ifndef SGK_LDS_HISTORY_DNA
#define SGK_LDS_HISTORY_DNA 0
#endif
// which fast-pass variant a preset uses: register ring (DNA, W1 = 3) or LDS history (RNA, W1 = 7)
#define USE_LDS_HISTORY(W1) ((W1) == 7 || SGK_LDS_HISTORY_DNA)
#ifndef SGK_LEAD_RNA
#define SGK_LEAD_RNA 256
#endif
constexpr int LEAD = 64;   // speculative warm-up (samples); multiple of 64
constexpr int BACK = 32;   // row margin before the pass start (>= W2 + 1)

__device__ inline uint32_t chunk_len(int64_t n) {
    const int64_t k = (n + 4095) / 4096;
    return (uint32_t)(k < 1 ? 64 : 64 * k);
}

// ---------------------------------------------------------------- t-statistic
// src/events.c:338-361, one rounding per C operator (FLT_EVAL_METHOD 0, no contraction).
template <int W>
__device__ inline float tstat_from_sums(double A, double A2, double B, double B2) {
    const float wf = (float)W;
    const float sum2 = (float)B;
    const float sumsq2 = (float)B2;
    const float mean1 = (float)(A / (double)wf);
    const float mean2 = sum2 / wf;
    const float m1sq = mean1 * mean1;
    const float m2sq = mean2 * mean2;
    const float q2 = sumsq2 / wf;
    double acc = A2 / (double)wf;
    acc = acc - (double)m1sq;
    acc = acc + (double)q2;
    acc = acc - (double)m2sq;
    float cv = (float)acc;
    cv = fmaxf(cv, FLT_MIN);
    const float delta = mean2 - mean1;
    const float cvw = cv / wf;
    return (float)(fabs((double)delta) / sqrt((double)cvw));
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

// per-lane state snapshots of one read's chunks, kept in LDS
struct DetSnap {
    DetState init[64];  // state a chunk's accepted run started from (at its chunk start)
    DetState at_e[64];  // state at the chunk end
    DetState st0[64];   // start state handed to a re-run
};

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
    return rc;
}

// One pass of the detector over the wave's chunks.
//   lead   : samples each lane starts before its chunk start (LEAD: speculative pass, 0: re-run)
//   active : whether this lane runs in this pass
//   st     : state at the pass start (lead == 0 only; the speculative pass starts fresh)
//   at_s   : out, normalised state when the lane reaches its chunk start s (speculative pass)
//   at_e   : out, normalised state when the lane reaches its chunk end e (written only when reached)
//   mn/mx  : min non-zero |x| / max |x| over the lane's own chunk (guard), speculative pass only
template <int W1, typename T, bool PREFIX>
__device__ __attribute__((noinline)) void detect_pass(const ReadCtx<T> &rc, char *lds, int lead, bool active,
                                                      int64_t s, int64_t e, uint32_t K, DetState st,
                                                      DetState &at_s, DetState &at_e, float &mn, float &mx) {
    constexpr int W2 = 2 * W1;
    const int64_t n = rc.n;
    const int64_t i_begin = s - lead;
    RowStream<T, 2> rs;
    rs.lds = lds;
    rs.base = rc.base;
    rs.lo = rc.lo;
    rs.hi = rc.hi;
    rs.rb = i_begin - BACK;
    rs.base_al = rc.vec_ok;
    const unsigned long long rowmask = __ballot(active);
    if (rowmask == 0ull) return;

    double A1 = 0, A1q = 0, B1 = 0, B1q = 0, A2 = 0, A2q = 0, B2 = 0, B2q = 0;
    if (!PREFIX) {
        rs.load_tile(0, rowmask);
        rs.load_tile(1, rowmask);
        if (active) {
            // direct summation of the four windows around i_begin (row index BACK)
#pragma unroll
            for (int k = 1; k <= W2; ++k) {
                const float x = to_pa(rs.get(BACK - k), rc.sc);
                const float xq = x * x;
                A2 = A2 + (double)x; A2q = A2q + (double)xq;
                if (k <= W1) { A1 = A1 + (double)x; A1q = A1q + (double)xq; }
            }
#pragma unroll
            for (int k = 0; k < W2; ++k) {
                const float x = to_pa(rs.get(BACK + k), rc.sc);
                const float xq = x * x;
                B2 = B2 + (double)x; B2q = B2q + (double)xq;
                if (k < W1) { B1 = B1 + (double)x; B1q = B1q + (double)xq; }
            }
        }
    }

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
        const int q = j + BACK;
        if (!PREFIX) {
            if (((q + W2) & 63) == 0 && j > 0) rs.load_tile((q + W2) >> 6, rowmask);
        }
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
                if (PREFIX) {
                    if (t1_ok && i >= W1 && i <= n - W1) {
                        const double p0 = rc.P[i], q0 = rc.P2[i];
                        v1 = sgk_tstat_fast<W1>(p0 - rc.P[i - W1], q0 - rc.P2[i - W1], rc.P[i + W1] - p0,
                                                rc.P2[i + W1] - q0);
                    }
                    if (t2_ok && i >= W2 && i <= n - W2) {
                        const double p0 = rc.P[i], q0 = rc.P2[i];
                        v2 = sgk_tstat_fast<W2>(p0 - rc.P[i - W2], q0 - rc.P2[i - W2], rc.P[i + W2] - p0,
                                                rc.P2[i + W2] - q0);
                    }
                } else {
                    if (t1_ok && i >= W1 && i <= n - W1) v1 = tstat_from_sums<W1>(A1, A1q, B1, B1q);
                    if (t2_ok && i >= W2 && i <= n - W2) v2 = tstat_from_sums<W2>(A2, A2q, B2, B2q);
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
        if (!PREFIX) {
            if (active) {
                // slide the four windows from index i to i+1 (exact in double)
                const float xm2 = to_pa(rs.get(q - W2), rc.sc);
                const float xm1 = to_pa(rs.get(q - W1), rc.sc);
                const float x0 = to_pa(rs.get(q), rc.sc);
                const float xp1 = to_pa(rs.get(q + W1), rc.sc);
                const float xp2 = to_pa(rs.get(q + W2), rc.sc);
                const double d0 = (double)x0, d0q = (double)(x0 * x0);
                A1 = (A1 + d0) - (double)xm1;  A1q = (A1q + d0q) - (double)(xm1 * xm1);
                A2 = (A2 + d0) - (double)xm2;  A2q = (A2q + d0q) - (double)(xm2 * xm2);
                B1 = (B1 + (double)xp1) - d0;  B1q = (B1q + (double)(xp1 * xp1)) - d0q;
                B2 = (B2 + (double)xp2) - d0;  B2q = (B2q + (double)(xp2 * xp2)) - d0q;
                if (lead > 0 && i >= s && i < e) {
                    const float ax = fabsf(x0);
                    mx = fmaxf(mx, ax);
                    if (ax != 0.0f) mn = fminf(mn, ax);
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


// ================================================================ fast detector pass
// Same semantics as detect_pass<.., false>, restructured for issue rate:
//  * window sums come from a register ring of W(p) = sum x[p..p+W1) (and of the float squares):
//      A1 = W(i-W1), B1 = W(i), A2 = W(i-2W1)+W(i-W1), B2 = W(i)+W(i+W1)        (W2 == 2*W1)
//    one new W per index: W(i+W1+1) = W(i+W1) + x[i+2W1] - x[i+W1]; the loop is unrolled by the
//    ring length so every ring index is a compile-time constant (registers, no scratch);
//  * only the leading sample x[i+2W1] is read from LDS per index (one row tile resident);
//  * t-statistics use the exact constant-division / certified-rsqrt forms of tstat_math.h;
//  * the automaton is written with selects (the reference's if/else ladder diverges on every lane);
//  * all indices are 32-bit (a read has < 2^31 samples, src/misc.c:20).
template <int W1>
struct RingCfg {
    static constexpr int R = (W1 == 3) ? 16 : 32;  // >= 3*W1+1, power of two, divides 64
    static constexpr int XR = (W1 == 3) ? 4 : 8;   // >= W1+1, power of two
};

template <int W1>
__device__ __forceinline__ int det_step_sel(DetState &d, int i, float v1, float v2) {
    // Pure boolean algebra on the comparison results (they stay in scalar mask registers) and one
    // select per state variable; no data-dependent branches.
    constexpr int W2 = 2 * W1;
    constexpr float ph = DetParam<W1>::ph, thr1 = DetParam<W1>::thr1, thr2 = DetParam<W1>::thr2;
    int emit;
    {   // short detector (masked_to == 0: only index 0 is skipped)
        const bool on = i > 0;
        const bool inpk = d.sp >= 0;
        const bool lower = v1 < d.sv;
        const bool rise = (v1 - d.sv) > ph;
        const bool higher = v1 > d.sv;
        const bool c1 = on & !inpk;   // no peak recorded yet
        const bool c2 = on & inpk;    // in a peak
        const bool upd = (c2 & higher) | (c1 & (lower | rise));
        const bool pos = (c2 & higher) | (c1 & !lower & rise);
        const float sv = upd ? v1 : d.sv;
        const int sp = pos ? i : d.sp;
        const bool strong = sv > thr1;
        const bool dom = c2 & strong;  // events.c:414-422: the short detector dominates the long one
        d.lmask = dom ? sp + W1 : d.lmask;
        d.lp = dom ? -1 : d.lp;
        d.lv = dom ? FLT_MAX : d.lv;
        const int lvalid0 = dom ? 0 : d.lvalid;
        const bool val = (d.svalid != 0) | (c2 & ((sv - v1) > ph) & strong);
        const bool em = c2 & val & ((i - sp) > W1 / 2);
        emit = em ? sp : -1;
        d.sp = em ? -1 : sp;
        d.sv = em ? v1 : sv;
        d.svalid = (val & !em) ? 1 : 0;
        d.lvalid = lvalid0;
    }
    {   // long detector
        const bool on = !(d.lmask >= i);
        const bool inpk = d.lp >= 0;
        const bool lower = v2 < d.lv;
        const bool rise = (v2 - d.lv) > ph;
        const bool higher = v2 > d.lv;
        const bool c1 = on & !inpk;
        const bool c2 = on & inpk;
        const bool upd = (c2 & higher) | (c1 & (lower | rise));
        const bool pos = (c2 & higher) | (c1 & !lower & rise);
        const float lv = upd ? v2 : d.lv;
        const int lp = pos ? i : d.lp;
        const bool val = (d.lvalid != 0) | (c2 & ((lv - v2) > ph) & (lv > thr2));
        const bool em = c2 & val & ((i - lp) > W2 / 2);
        emit = em ? lp : emit;
        d.lp = em ? -1 : lp;
        d.lv = em ? v2 : lv;
        d.lvalid = (val & !em) ? 1 : 0;
    }
    return emit;
}

// Exact (reference-expression) t-statistic at index i of a read, window sums formed directly from
// the samples in global memory.  Out of line: only reached when a fast evaluation's certificate
// fails (about 2^-13 of the evaluations).
__device__ unsigned long long g_exact_redo_count = 0;  // diagnostics: uncertified evaluations redone

template <typename T>
__device__ __attribute__((noinline)) float tstat_exact_at(const T *base, Scale sc, int i, int w) {
    atomicAdd(&g_exact_redo_count, 1ull);
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
    template <int U>
    __device__ __forceinline__ int16_t raw() const {
        return (int16_t)((U & 1) ? (w[U / 2] >> 16) : (w[U / 2] & 0xffffu));
    }
};
template <>
struct Lead16<float> {
    float w[16];
    template <int U>
    __device__ __forceinline__ float get(const Scale &) const { return w[U]; }
    template <int U>
    __device__ __forceinline__ float raw() const { return w[U]; }
};
typedef uint32_t sgk_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Samples in front of a read (the speculative warm-up of its first chunks reaches there) are whatever the caller's
// buffer holds.  No t-statistic that sees them is used, but they pass through the RUNNING window sums, and a value
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
// (plus uncertified evaluations) are re-evaluated from the scratch prefix arrays.
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

// marks (as "redo exactly") the indices q0..q0+3 that lie within a window length of an event
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

// State of one fast pass.  Every ring access uses a compile-time index (U is a template parameter
// and the pass starts on a multiple of the ring length), so the arrays live in registers; the R
// steps of one loop iteration are expanded with fold expressions, four at a time:
// t-statistics of 4 indices -> (rare, rolled, out of line) exact redo of uncertified ones ->
// both automata on those 4 indices.
//
// Samples are NOT staged through LDS here: each lane loads the 16 leading samples of the next
// block (x[i+2*W1], 32 bytes) straight from global memory one block ahead.  A wave touches 64
// different 128-byte lines per load instruction; each line is consumed over 4 consecutive blocks
// and stays in L2 meanwhile, so HBM traffic remains one pass over the samples and there are no
// barriers or cooperative loads in the loop.
template <int W1, typename T, bool FLAGGED>
struct FastPass {
    static constexpr int W2 = 2 * W1, R = RingCfg<W1>::R, XR = RingCfg<W1>::XR;
    static constexpr int NL = R / 16;  // 16-sample lead groups per block
    double Ws[R], Wq[R];
    float xs[XR], xq[XR];
    float t1[4], t2[4];
    Lead16<T> cur[NL];  // x[ib + W2 .. ib + W2 + R)
    DetState d;
    unsigned long long wcur, wprev;
    unsigned long long *bm;
    const T *base;
    int lo, hi;  // legal read-relative load range
    Scale sc;
    int n, s, e, ib, wb;
    unsigned cnt1, cnt2;
    unsigned bad1, bad2;
    bool done;
    RepairCtx rep;   // FLAGGED only
    int next_t;      // FLAGGED only: smallest event position that can still matter

    // Unconditional 32-byte load of x[pos .. pos+16).  Positions outside the readable range are
    // redirected to the nearest readable group: whatever value a position yields is used
    // consistently when it enters and when it leaves a window, and no t-statistic whose window
    // reaches outside [0, n) is ever used (events.c:332-338).  Positions before the read are
    // replaced by the read's first sample (lead_fix_head); positions behind it only ever enter the
    // leading windows after their last used t-statistic.
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

    // phase 1: window sums and both t-statistics of index ib+U; advance the rings
    template <int U>
    __device__ __forceinline__ void tstep() {
        const int i = ib + U;
        const float xn = cur[U / 16].template get<U % 16>(sc);  // leading sample x[i + 2*W1]
        const float xqn = xn * xn;
        const double a1 = Ws[(U - W1) & (R - 1)], a1q = Wq[(U - W1) & (R - 1)];
        const double b1 = Ws[U & (R - 1)], b1q = Wq[U & (R - 1)];
        const double a2 = Ws[(U - W2) & (R - 1)] + a1, a2q = Wq[(U - W2) & (R - 1)] + a1q;
        const double b2 = b1 + Ws[(U + W1) & (R - 1)], b2q = b1q + Wq[(U + W1) & (R - 1)];
        bool ok1, ok2;
        float v1, v2;
        sgk_tstat_try_pair<W1>(a1, a1q, b1, b1q, a2, a2q, b2, b2q, v1, v2, ok1, ok2);
        const bool in1 = (unsigned)(i - W1) < cnt1, in2 = (unsigned)(i - W2) < cnt2;
        t1[U & 3] = in1 ? v1 : 0.0f;
        t2[U & 3] = in2 ? v2 : 0.0f;
        bad1 |= (in1 && !ok1) ? (1u << (U & 3)) : 0u;
        bad2 |= (in2 && !ok2) ? (1u << (U & 3)) : 0u;
        Ws[(U + W1 + 1) & (R - 1)] = (Ws[(U + W1) & (R - 1)] + (double)xn) - (double)xs[(U + W1) & (XR - 1)];
        Wq[(U + W1 + 1) & (R - 1)] = (Wq[(U + W1) & (R - 1)] + (double)xqn) - (double)xq[(U + W1) & (XR - 1)];
        xs[(U + W2) & (XR - 1)] = xn;
        xq[(U + W2) & (XR - 1)] = xqn;
        // keep the evaluations from being interleaved: their live ranges would otherwise add up
        // to far more than the register budget; latency is hidden across waves instead
        __builtin_amdgcn_sched_barrier(0);
    }
    // phase 2: both automata at index ib+U
    template <int U>
    __device__ __forceinline__ void dstep() {
        const int i = ib + U;
        if (!done && (unsigned)i < (unsigned)n) {
            const int p = det_step_sel<W1>(d, i, t1[U & 3], t2[U & 3]);
            // record the emitted peak if this lane owns its position (selects; the store is the rare case
            // of a peak older than the two bitmap words held in registers)
            const bool own = (p >= s) & (p < e);
            const int wi = p >> 6;
            const unsigned long long bit = 1ull << (p & 63);
            wcur |= (own & (wi == wb)) ? bit : 0ull;
            wprev |= (own & (wi == wb - 1)) ? bit : 0ull;
            if (own & (wi < wb - 1)) bm[wi] |= bit;  // already retired word, owned by this lane only
        }
    }
    // four indices U0..U0+3
    template <int U0>
    __device__ __forceinline__ void quad() {
        bad1 = 0u;
        bad2 = 0u;
        tstep<U0>();
        tstep<U0 + 1>();
        tstep<U0 + 2>();
        tstep<U0 + 3>();
        if constexpr (FLAGGED) repair_mark<W1>(rep, next_t, ib + U0, cnt1, cnt2, bad1, bad2);
        // rare: evaluations whose certificate failed are redone with the reference expression
        if (__any((bad1 | bad2) != 0u))
        while (__any((bad1 | bad2) != 0u)) {
            if ((bad1 | bad2) != 0u) {
                const bool first = bad1 != 0u;
                const unsigned m = first ? bad1 : bad2;
                const int u = __ffs((int)m) - 1;
                float v;
                if constexpr (FLAGGED) v = tstat_prefix_at(rep.P, rep.P2, ib + U0 + u, first ? W1 : W2);
                else v = tstat_exact_at<T>(base, sc, ib + U0 + u, first ? W1 : W2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k == u) {
                        if (first) t1[k] = v;
                        else t2[k] = v;
                    }
                }
                if (first) bad1 &= bad1 - 1u;
                else bad2 &= bad2 - 1u;
            }
        }
        dstep<U0>();
        dstep<U0 + 1>();
        dstep<U0 + 2>();
        dstep<U0 + 3>();
    }
    template <int... Qs>
    __device__ __forceinline__ void block(std::integer_sequence<int, Qs...>) {
        (quad<4 * Qs>(), ...);
    }

    template <int K0>
    __device__ __forceinline__ void init_w(double &a, double &aq, const float (&w)[4 * W1]) {
        // W(p+1) from W(p), p = i_begin - 2*W1 + K0; w[k] = x[i_begin - 2*W1 + k]
        const float xin = w[K0 + W1], xout = w[K0];
        a = (a + (double)xin) - (double)xout;
        aq = (aq + (double)(xin * xin)) - (double)(xout * xout);
        Ws[(-W2 + K0 + 1) & (R - 1)] = a;
        Wq[(-W2 + K0 + 1) & (R - 1)] = aq;
    }
    template <int... Ks>
    __device__ __forceinline__ void init_ws(double &a, double &aq, const float (&w)[4 * W1],
                                            std::integer_sequence<int, Ks...>) {
        (init_w<Ks>(a, aq, w), ...);
    }
};

template <int W1, typename T, bool FLAGGED>
__device__ __forceinline__ void pass_fast(const ReadCtx<T> &rc, int lead, bool active, int s, int e, int K,
                                          DetSnap *snap, const RepairCtx *rep) {
    using FP = FastPass<W1, T, FLAGGED>;
    constexpr int W2 = FP::W2, R = FP::R, XR = FP::XR, NL = FP::NL;
    if (!__any(active)) return;
    FP f;
    f.n = (int)rc.n;
    f.s = s;
    f.e = e;
    f.bm = rc.bm;
    f.sc = rc.sc;
    f.base = rc.base;
    f.lo = (int)(rc.lo < -(1 << 30) ? -(1 << 30) : rc.lo);
    f.hi = (int)(rc.hi > 0x7fffffffLL ? 0x7fffffffLL : rc.hi);
    const int n = f.n;
    const int i_begin = s - lead;  // multiple of 64

    // ring slots are addressed by (position - i_begin) & (R-1); i_begin is a multiple of R
#pragma unroll
    for (int k = 0; k < R; ++k) { f.Ws[k] = 0.0; f.Wq[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.t1[k] = 0.0f; f.t2[k] = 0.0f; }
    {
        // x[i_begin - 2*W1 .. i_begin + 2*W1): W(i_begin-2*W1), then slide to W(i_begin+W1); x ring
        float w[4 * W1];
#pragma unroll
        for (int k = 0; k < 4 * W1; ++k) {
            int p = i_begin - W2 + k;
            p = p > f.hi - 1 ? f.hi - 1 : p;
            p = p < 0 ? 0 : p;  // positions before the read: its first sample (see lead_fix_head)
            w[k] = to_pa(f.base[p], f.sc);
        }
        double a = 0.0, aq = 0.0;
#pragma unroll
        for (int k = 0; k < W1; ++k) {
            a = a + (double)w[k];
            aq = aq + (double)(w[k] * w[k]);
        }
        f.Ws[(-W2) & (R - 1)] = a;
        f.Wq[(-W2) & (R - 1)] = aq;
        f.init_ws(a, aq, w, std::make_integer_sequence<int, 3 * W1>{});
#pragma unroll
        for (int k = 0; k < XR; ++k) { f.xs[k] = 0.0f; f.xq[k] = 0.0f; }
#pragma unroll
        for (int k = 0; k < W1; ++k) {  // x[i_begin + W1 + k]
            f.xs[(W1 + k) & (XR - 1)] = w[W2 + W1 + k];
            f.xq[(W1 + k) & (XR - 1)] = w[W2 + W1 + k] * w[W2 + W1 + k];
        }
    }
    // leading samples of the first block
#pragma unroll
    for (int g = 0; g < NL; ++g) f.load_lead(f.cur[g], i_begin + W2 + 16 * g);

    f.d = (lead > 0) ? det_fresh(i_begin <= 0 ? 0 : -1) : snap->st0[lane_id()];
    f.wcur = 0ull;
    f.wprev = 0ull;
    const int wlo = s >> 6, whi = (e + 63) >> 6;
    f.done = !active;
    if constexpr (FLAGGED) {
        f.rep = *rep;
        // first event whose influence [t-W2+1, t+W2] is not entirely before this pass' first index
        f.next_t = 0x7fffffff;
        for (int k = 0; k < f.rep.nev; ++k) {
            const int t = f.rep.ev[k];
            if (t + W2 >= i_begin && t < f.next_t) f.next_t = t;
        }
    }
    const int main_steps = lead + K;
    f.cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    f.cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;

    int jb = 0;
    for (;; jb += R) {
        if (jb >= main_steps && !__any(!f.done)) break;
        const int ib = i_begin + jb;
        const int wb = ib >> 6;  // bitmap word of every index of this block (R divides 64)
        if ((jb & 63) == 0 && jb > 0 && active) {
            const int wr = wb - 2;
            if (wr >= wlo && wr < whi) f.bm[wr] = f.wprev;
            f.wprev = f.wcur;
            f.wcur = 0ull;
        }
        if (active) {
            // state snapshots live in LDS (they are only needed after the pass): frees ~28 VGPRs
            if (lead > 0 && jb == lead) snap->init[lane_id()] = det_norm(f.d, ib);
            if (ib == e) snap->at_e[lane_id()] = det_norm(f.d, ib);
            if (ib >= e) {
                const bool pend = (f.d.sp >= 0 && f.d.sp < e) || (f.d.lp >= 0 && f.d.lp < e);
                if (!pend || ib >= n) f.done = true;  // the reference's loop ends at n-1: pending peaks are dropped
            }
        }
        f.ib = ib;
        f.wb = wb;
        // issue the loads of the NEXT block's leading samples now; consumed one iteration later
        Lead16<T> nxt[NL];
#pragma unroll
        for (int g = 0; g < NL; ++g) f.load_lead(nxt[g], ib + R + W2 + 16 * g);
        f.block(std::make_integer_sequence<int, R / 4>{});
#pragma unroll
        for (int g = 0; g < NL; ++g) f.cur[g] = nxt[g];
    }
    if (active) {
        const int wbl = (i_begin + jb - 1) >> 6;  // word of the last processed index
        if (wbl - 1 >= wlo && wbl - 1 < whi) f.bm[wbl - 1] = f.wprev;
        if (wbl >= wlo && wbl < whi) f.bm[wbl] = f.wcur;
    }
}

// ---------------------------------------------------------------- fast pass, LDS-history variant
// Same contract as pass_fast.  Instead of a register ring of partial window sums (which needs a
// 3*W1+1 deep ring and an unroll by its length: too much for W1 = 7), it keeps the eight running
// window sums and reads the four trailing samples x[i-2W1], x[i-W1], x[i], x[i+W1] from a per-lane
// history ring in LDS (64 raw samples per lane; the leading 16-sample group of each block is
// written into it once).  16-step unroll for every window size.
template <typename T>
struct HistRing {
    static constexpr int ROW_BYTES = 64 * (int)sizeof(T) + 4;  // odd dword stride: conflict-free lane-per-row
    static constexpr int LDS_BYTES = 64 * ROW_BYTES;
};

template <int W1, typename T, bool FLAGGED>
struct FastPassL {
    static constexpr int W2 = 2 * W1;
    double A1, A1q, B1, B1q, A2, A2q, B2, B2q;
    float t1[4], t2[4];
    Lead16<T> cur;
    DetState d;
    unsigned long long wcur, wprev;
    unsigned long long *bm;
    const T *base;
    char *row;   // this lane's history row in LDS; ring index of position p is (p - W2) & 63
    int lo, hi;
    Scale sc;
    int n, s, e, ib, wb;
    unsigned cnt1, cnt2;
    unsigned bad1, bad2;
    bool done;
    RepairCtx rep;   // FLAGGED only
    int next_t;      // FLAGGED only: smallest event position that can still matter

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
    __device__ __forceinline__ T raw_hist(int pos) const {  // stored sample at pos (within the resident window)
        return *reinterpret_cast<const T *>(row + (((pos - W2) & 63) * (int)sizeof(T)));
    }
    __device__ __forceinline__ float hist(int pos) const { return to_pa(raw_hist(pos), sc); }
    __device__ __forceinline__ void store_group(const Lead16<T> &g, int pos) {  // x[pos..pos+16), pos-W2 multiple of 16
        uint32_t *dst = reinterpret_cast<uint32_t *>(row + (((pos - W2) & 63) * (int)sizeof(T)));
        constexpr int ND = 16 * (int)sizeof(T) / 4;
        uint32_t tmp[ND];
        __builtin_memcpy(tmp, g.w, sizeof(tmp));
#pragma unroll
        for (int k = 0; k < ND; ++k) dst[k] = tmp[k];
    }

    template <int U>
    __device__ __forceinline__ void tstep() {
        const int i = ib + U;
        bool ok1, ok2;
        float v1, v2;
        sgk_tstat_try_pair<W1>(A1, A1q, B1, B1q, A2, A2q, B2, B2q, v1, v2, ok1, ok2);
        const bool in1 = (unsigned)(i - W1) < cnt1, in2 = (unsigned)(i - W2) < cnt2;
        t1[U & 3] = in1 ? v1 : 0.0f;
        t2[U & 3] = in2 ? v2 : 0.0f;
        bad1 |= (in1 && !ok1) ? (1u << (U & 3)) : 0u;
        bad2 |= (in2 && !ok2) ? (1u << (U & 3)) : 0u;
        // slide the four windows from index i to i+1 (exact in double); pA conversion and squares of
        // the five samples involved on packed pairs
        const f32x2 xm = to_pa2(raw_hist(i - W2), raw_hist(i - W1), sc);        // x[i-2W1], x[i-W1]
        const f32x2 xp = to_pa2(raw_hist(i + W1), cur.template raw<U>(), sc);   // x[i+W1], x[i+2W1]
        const float x0 = to_pa(raw_hist(i), sc);
        const f32x2 xmq = xm * xm, xpq = xp * xp;
        const double d0 = (double)x0, d0q = (double)(x0 * x0);
        A1 = (A1 + d0) - (double)xm.y;  A1q = (A1q + d0q) - (double)xmq.y;
        A2 = (A2 + d0) - (double)xm.x;  A2q = (A2q + d0q) - (double)xmq.x;
        B1 = (B1 + (double)xp.x) - d0;  B1q = (B1q + (double)xpq.x) - d0q;
        B2 = (B2 + (double)xp.y) - d0;  B2q = (B2q + (double)xpq.y) - d0q;
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int U>
    __device__ __forceinline__ void dstep() {
        const int i = ib + U;
        if (!done && (unsigned)i < (unsigned)n) {
            const int p = det_step_sel<W1>(d, i, t1[U & 3], t2[U & 3]);
            // record the emitted peak if this lane owns its position (selects; the store is the rare case
            // of a peak older than the two bitmap words held in registers)
            const bool own = (p >= s) & (p < e);
            const int wi = p >> 6;
            const unsigned long long bit = 1ull << (p & 63);
            wcur |= (own & (wi == wb)) ? bit : 0ull;
            wprev |= (own & (wi == wb - 1)) ? bit : 0ull;
            if (own & (wi < wb - 1)) bm[wi] |= bit;  // already retired word, owned by this lane only
        }
    }
    template <int U0>
    __device__ __forceinline__ void quad() {
        bad1 = 0u;
        bad2 = 0u;
        tstep<U0>();
        tstep<U0 + 1>();
        tstep<U0 + 2>();
        tstep<U0 + 3>();
        if constexpr (FLAGGED) repair_mark<W1>(rep, next_t, ib + U0, cnt1, cnt2, bad1, bad2);
        if (__any((bad1 | bad2) != 0u))
        while (__any((bad1 | bad2) != 0u)) {
            if ((bad1 | bad2) != 0u) {
                const bool first = bad1 != 0u;
                const unsigned m = first ? bad1 : bad2;
                const int u = __ffs((int)m) - 1;
                float v;
                if constexpr (FLAGGED) v = tstat_prefix_at(rep.P, rep.P2, ib + U0 + u, first ? W1 : W2);
                else v = tstat_exact_at<T>(base, sc, ib + U0 + u, first ? W1 : W2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k == u) {
                        if (first) t1[k] = v;
                        else t2[k] = v;
                    }
                }
                if (first) bad1 &= bad1 - 1u;
                else bad2 &= bad2 - 1u;
            }
        }
        dstep<U0>();
        dstep<U0 + 1>();
        dstep<U0 + 2>();
        dstep<U0 + 3>();
    }
};

template <int W1, typename T, bool FLAGGED>
__device__ __forceinline__ void pass_fast_lds(const ReadCtx<T> &rc, char *hist_lds, int lead, bool active, int s,
                                              int e, int K, DetSnap *snap, const RepairCtx *rep) {
    using FP = FastPassL<W1, T, FLAGGED>;
    constexpr int W2 = FP::W2, R = 16;
    if (!__any(active)) return;
    FP f;
    f.n = (int)rc.n;
    f.s = s;
    f.e = e;
    f.bm = rc.bm;
    f.sc = rc.sc;
    f.base = rc.base;
    f.lo = (int)(rc.lo < -(1 << 30) ? -(1 << 30) : rc.lo);
    f.hi = (int)(rc.hi > 0x7fffffffLL ? 0x7fffffffLL : rc.hi);
    f.row = hist_lds + lane_id() * HistRing<T>::ROW_BYTES;
    const int n = f.n;
    const int i_begin = s - lead;  // multiple of 64
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.t1[k] = 0.0f; f.t2[k] = 0.0f; }
    // history: x[i_begin - 2*W2 .. i_begin + W2) must be resident before the first block; fill the
    // three 16-sample groups that cover it (ring indices (p - W2) & 63)
#pragma unroll
    for (int g = -3; g < 0; ++g) {
        Lead16<T> tmp;
        f.load_lead(tmp, i_begin + W2 + 16 * g);
        f.store_group(tmp, i_begin + W2 + 16 * g);
    }
    f.load_lead(f.cur, i_begin + W2);
    __syncthreads();
    // window sums at i_begin by direct summation from the history ring
    f.A1 = f.A1q = f.B1 = f.B1q = f.A2 = f.A2q = f.B2 = f.B2q = 0.0;
#pragma unroll
    for (int k = 1; k <= W2; ++k) {
        const float x = f.hist(i_begin - k);
        const float xq = x * x;
        f.A2 = f.A2 + (double)x; f.A2q = f.A2q + (double)xq;
        if (k <= W1) { f.A1 = f.A1 + (double)x; f.A1q = f.A1q + (double)xq; }
    }
    // B windows: x[i_begin .. i_begin+W2) is the tail of the three groups just stored.  They MUST come from
    // the ring as well: a group that load_lead had to redirect (chunk 0 of a read with exactly `lead` samples
    // of head room) holds shifted samples, and every value has to leave a window sum as the same number it
    // entered with -- the running sums never reset, so one mismatch would stay in them for the whole chunk.
#pragma unroll
    for (int k = 0; k < W2; ++k) {
        const float x = f.hist(i_begin + k);
        const float xq = x * x;
        f.B2 = f.B2 + (double)x; f.B2q = f.B2q + (double)xq;
        if (k < W1) { f.B1 = f.B1 + (double)x; f.B1q = f.B1q + (double)xq; }
    }

    f.d = (lead > 0) ? det_fresh(i_begin <= 0 ? 0 : -1) : snap->st0[lane_id()];
    f.wcur = 0ull;
    f.wprev = 0ull;
    const int wlo = s >> 6, whi = (e + 63) >> 6;
    f.done = !active;
    if constexpr (FLAGGED) {
        f.rep = *rep;
        // first event whose influence [t-W2+1, t+W2] is not entirely before this pass' first index
        f.next_t = 0x7fffffff;
        for (int k = 0; k < f.rep.nev; ++k) {
            const int t = f.rep.ev[k];
            if (t + W2 >= i_begin && t < f.next_t) f.next_t = t;
        }
    }
    const int main_steps = lead + K;
    f.cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    f.cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;

    int jb = 0;
    for (;; jb += R) {
        if (jb >= main_steps && !__any(!f.done)) break;
        const int ib = i_begin + jb;
        const int wb = ib >> 6;
        if ((jb & 63) == 0 && jb > 0 && active) {
            const int wr = wb - 2;
            if (wr >= wlo && wr < whi) f.bm[wr] = f.wprev;
            f.wprev = f.wcur;
            f.wcur = 0ull;
        }
        if (active) {
            if (lead > 0 && jb == lead) snap->init[lane_id()] = det_norm(f.d, ib);
            if (ib == e) snap->at_e[lane_id()] = det_norm(f.d, ib);
            if (ib >= e) {
                const bool pend = (f.d.sp >= 0 && f.d.sp < e) || (f.d.lp >= 0 && f.d.lp < e);
                if (!pend || ib >= n) f.done = true;
            }
        }
        f.ib = ib;
        f.wb = wb;
        // publish this block's leading group to the history ring (it becomes x[i+W1], x[i], ... later)
        f.store_group(f.cur, ib + W2);
        Lead16<T> nxt;
        f.load_lead(nxt, ib + R + W2);
        __syncthreads();
        f.template quad<0>();
        f.template quad<4>();
        f.template quad<8>();
        f.template quad<12>();
        f.cur = nxt;
    }
    if (active) {
        const int wbl = (i_begin + jb - 1) >> 6;
        if (wbl - 1 >= wlo && wbl - 1 < whi) f.bm[wbl - 1] = f.wprev;
        if (wbl >= wlo && wbl < whi) f.bm[wbl] = f.wcur;
    }
}

// speculative pass + verification / re-run loop (one inlined copy of pass_fast)
template <int W1, typename T, bool FLAGGED>
__device__ bool detect_read_fast(const ReadCtx<T> &rc, EvHeader *hdr, DetSnap *snap, char *hist_lds,
                                 const RepairCtx *rep) {
    const int n = (int)rc.n;
    if (n <= 0) return true;
    // the fast pass uses unguarded 4-byte-aligned 32-byte vector loads: it needs 64 readable samples
    // before the read (speculative warm-up of chunk 0) and 16 after it; other reads (e.g. a read at
    // the very start of a caller's buffer) take the exact fallback
    if ((reinterpret_cast<uintptr_t>(rc.base) & 3u) != 0 || rc.lo > -((W1 == 7) ? SGK_LEAD_RNA : LEAD) || rc.hi < (int64_t)n + 16) return false;
    const int K = (int)chunk_len(n);
    const int c = lane_id();
    const int s = c * K;
    const int e = (s + K < n) ? s + K : n;
    const bool active = (int64_t)c * K < (int64_t)n;
    snap->init[c] = det_fresh(0);
    snap->at_e[c] = det_fresh(0);
    // speculative warm-up before every chunk.  RNA events are ~5x longer, so the automata converge later: with 64
    // samples ~1.4 % of the chunk boundaries need a re-run, with 256 about 0.002 %.  A re-run costs the wave one
    // more pass over a chunk (K samples), the warm-up costs `lead` samples per lane: short reads (small K) are
    // better off with a short warm-up and the occasional re-run, long reads with a long one.
    int lead = LEAD;
    if (W1 == 7) lead = K <= 128 ? 64 : (K <= 512 ? 128 : SGK_LEAD_RNA);
    bool run = active;
    for (int iter = 0; iter < 66; ++iter) {
        if constexpr (USE_LDS_HISTORY(W1)) pass_fast_lds<W1, T, FLAGGED>(rc, hist_lds, lead, run, s, e, K, snap, rep);
        else pass_fast<W1, T, FLAGGED>(rc, lead, run, s, e, K, snap, rep);
        __syncthreads();
        // chunk c is right iff it started (at s) from the state chunk c-1 ended with
        const DetState pe = snap->at_e[c > 0 ? c - 1 : 0];
        const DetState mine = snap->init[c];
        const bool bad = active && c > 0 && !det_equal(pe, mine);
        const unsigned long long badmask = __ballot(bad);
        if (badmask == 0ull) break;
        __syncthreads();
        if (bad) {
            snap->init[c] = pe;
            snap->st0[c] = pe;
        }
        run = bad;
        lead = 0;
        if (c == 0) atomicAdd(&hdr->n_rerun, (uint32_t)__popcll(badmask));
        __syncthreads();
    }
    return true;
}

__device__ inline bool guard_ok(float mn, float mx, int64_t n) {
    if (!(mx > 0.0f)) return true;  // all samples zero
    const int eb = ilogb((double)n * (double)mx), em = ilogb((double)mn);
    if (eb - em > 29) return false;
    const float mnq = mn * mn, mxq = mx * mx;
    if (mnq < FLT_MIN) return false;
    const int ebq = ilogb((double)n * (double)mxq), emq = ilogb((double)mnq);
    return ebq - emq <= 29;
}

// Detector over one read by one wave.  Returns true when the exactness guard fails
// (fast path only).
template <int W1, typename T, bool PREFIX>
__device__ bool detect_read(const ReadCtx<T> &rc, char *lds, EvHeader *hdr) {
    const int64_t n = rc.n;
    if (n <= 0) return false;
    const uint32_t K = chunk_len(n);
    const int c = lane_id();
    const int64_t s = (int64_t)c * K;
    const int64_t e = (s + K < n) ? s + K : n;
    const bool active = s < n;
    const DetState fresh = det_fresh(0);
    DetState at_s = fresh, at_e = fresh;
    float mn = FLT_MAX, mx = 0.0f;
    detect_pass<W1, T, PREFIX>(rc, lds, LEAD, active, s, e, K, fresh, at_s, at_e, mn, mx);
    DetState init = at_s;
    for (int iter = 0; iter < 64; ++iter) {
        const DetState pe = det_shfl_up(at_e);
        const bool bad = active && c > 0 && !det_equal(pe, init);
        const unsigned long long badmask = __ballot(bad);
        if (badmask == 0ull) break;
        if (bad) init = pe;
        float mn2 = FLT_MAX, mx2 = 0.0f;
        DetState unused = fresh;
        detect_pass<W1, T, PREFIX>(rc, lds, 0, bad, s, e, K, pe, unused, at_e, mn2, mx2);
        if (c == 0) atomicAdd(&hdr->n_rerun, (uint32_t)__popcll(badmask));
    }
    return false;
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
    a.ev_start[slot0 + k] = ps;
    a.ev_length[slot0 + k] = pe - ps;
    a.ev_mean[slot0 + k] = m;
    a.ev_stdv[slot0 + k] = sd;
}

#ifndef SGK_BT
#define SGK_BT 32
#endif
constexpr int BT = SGK_BT;                       // samples per lane per builder tile (16 or 32)
#ifndef SGK_BREC
#define SGK_BREC 512
#endif
// Boundary records per tile in LDS (18 bytes each; 10 KB per wave with the lane prefixes: 16 waves per CU).  The detector can emit a boundary every 3 samples (64 * 11 per
// tile), but sizing LDS for that costs occupancy; a tile with more than BREC boundaries (events shorter than 4
// samples on average over 2048 samples) sends its read to k_event_fallback instead, which has no such limit.
constexpr int BREC = SGK_BREC;
struct BuildLds {
    double S[BREC];
    double S2[BREC];
    double pt[64];
    double pt2[64];
    uint16_t p[BREC];  // tile-relative sample index of the boundary
};
static_assert(64 * BT <= 65536 && sizeof(BuildLds) <= 10240, "builder LDS budget: 16 waves per CU");

template <typename T>
__device__ void build_read(const EvArgs &a, const ReadCtx<T> &rc, uint32_t r, BuildLds *L) {
    const int64_t n = rc.n;
    const int l = lane_id();
    const uint64_t slot0 = a.ev_slots[r], cap = a.ev_slots[r + 1] - slot0;
    if (n <= 0) {
        if (l == 0) { a.n_events[r] = 0; a.flags[r] = 0; }
        return;
    }
    const uint32_t *bm32 = reinterpret_cast<const uint32_t *>(rc.bm);
    bool overflow = false, dense = false;
    uint32_t rank = 0, prevp = 0;
    double Gprev = 0.0, G2prev = 0.0;  // prefix sums at the previous boundary, relative to the current tile start
    // exactness guard: min non-zero |x| and max |x| over the read, tracked on the bit patterns (non-negative
    // floats order like unsigned integers; zero - 1 wraps to the top, so it never wins the minimum; inf / nan
    // end up above every finite value and fail the guard)
    uint32_t mnb = 0xffffffffu, mxb = 0u;
    constexpr int NV = BT * (int)sizeof(T) / 16;
    // tile loader: this lane's 32 samples and its 32 bitmap bits.  The next tile is fetched while the
    // current one is processed (register double buffer).
    auto load_tile = [&](int64_t tb, T (&buf)[BT], uint32_t &bits, int &nvalid) {
        const int64_t pos0 = tb + (int64_t)l * BT;
        bits = (pos0 < n) ? (bm32[pos0 >> 5] >> (pos0 & 31)) : 0u;
        if (BT < 32) bits &= (1u << (BT & 31)) - 1u;
        const int64_t rem = n - pos0;
        nvalid = rem <= 0 ? 0 : (rem >= BT ? BT : (int)rem);
        if (nvalid < BT) bits &= (nvalid == 0) ? 0u : ((1u << nvalid) - 1u);
        if (rc.vec_ok && pos0 + BT <= rc.hi && pos0 < n) {
            const uint4 *src = reinterpret_cast<const uint4 *>(rc.base + pos0);
            uint4 v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = src[k];
            __builtin_memcpy(buf, v, sizeof(buf));
        } else {
#pragma unroll
            for (int k = 0; k < BT; ++k) buf[k] = (k < nvalid) ? rc.base[pos0 + k] : (T)0;
        }
    };
    T nbuf[BT];
    uint32_t nbits;
    int nnvalid;
    load_tile(0, nbuf, nbits, nnvalid);
    for (int64_t tb = 0; tb < n; tb += 64 * BT) {
        const int64_t pos0 = tb + (int64_t)l * BT;
        T buf[BT];
#pragma unroll
        for (int k = 0; k < BT; ++k) buf[k] = nbuf[k];
        const uint32_t bits = nbits;
        const int nvalid = nnvalid;
        if (tb + 64 * BT < n) load_tile(tb + 64 * BT, nbuf, nbits, nnvalid);
        const int cnt = __popc(bits);
        const int incl = wave_incl_scan_i(cnt);
        const int excl = incl - cnt;
        const int total = wave_last_i(incl);
        // walk: lane-relative prefix sums, boundary records.  Only the last tile of a read can hold lanes with
        // fewer than BT samples: every other tile skips the per-sample validity select.
        double S = 0.0, S2 = 0.0;
        auto walk = [&](auto full_tag, auto checked_tag) {
            constexpr bool FULL = decltype(full_tag)::value, CHECKED = decltype(checked_tag)::value;
            int off = excl * 2;  // byte offset of the next record in p[]; four times that in S[] and S2[]
            char *const rp = reinterpret_cast<char *>(L->p), *const rs = reinterpret_cast<char *>(L->S),
                        *const rs2 = reinterpret_cast<char *>(L->S2);
#pragma unroll
            for (int k = 0; k < BT; ++k) {
                float x = to_pa(buf[k], rc.sc);
                if (!FULL && k >= nvalid) x = 0.0f;
                const float xq = x * x;
                const uint32_t ab = __float_as_uint(x) & 0x7fffffffu;
                mxb = ab > mxb ? ab : mxb;
                mnb = (ab - 1u) < mnb ? (ab - 1u) : mnb;
                if ((bits >> k) & 1u) {
                    if (!CHECKED || off < BREC * 2) {
                        *reinterpret_cast<uint16_t *>(rp + off) = (uint16_t)(l * BT + k);
                        *reinterpret_cast<double *>(rs + 4 * off) = S;
                        *reinterpret_cast<double *>(rs2 + 4 * off) = S2;
                    }
                    off += 2;
                }
                S = S + (double)x;
                S2 = S2 + (double)xq;
            }
        };
        // a tile with more than BREC boundaries takes the bounds-checked walk; its read is redone by the fallback
        if (tb + 64 * BT <= n && total <= BREC) walk(std::true_type{}, std::false_type{});
        else walk(std::false_type{}, std::true_type{});
        const double inS = wave_incl_scan_d(S), inS2 = wave_incl_scan_d(S2);
        L->pt[l] = inS - S;
        L->pt2[l] = inS2 - S2;
        const double tileS = wave_last_d(inS), tileS2 = wave_last_d(inS2);
        __syncthreads();
        const int tot = total < BREC ? total : BREC;
        if (total > BREC) dense = true;
        // one event per lane per round.  Prefix sums are kept relative to the tile start (exact under the guard, so
        // no absolute base is needed); the previous boundary of lane l is lane l-1's record, lane 0 takes the
        // carry: the last record of the previous round / tile.
        for (int k0 = 0; k0 < tot; k0 += 64) {
            const int k = k0 + l;
            const bool act = k < tot;
            const int kk = act ? k : tot - 1;
            const uint32_t pr = L->p[kk], p = (uint32_t)tb + pr;
            const int ln = (int)(pr / BT);
            const double G = L->pt[ln] + L->S[kk];
            const double G2 = L->pt2[ln] + L->S2[kk];
            const uint32_t pp = (uint32_t)wave_shr1_i((int)p, (int)prevp);
            const double Gp = wave_shr1_d(G, Gprev), G2p = wave_shr1_d(G2, G2prev);
            if (act) store_event(a, slot0, cap, (uint64_t)rank + (uint64_t)k, pp, p, G - Gp, G2 - G2p, overflow);
            const int last = (tot - k0) < 64 ? (tot - k0 - 1) : 63;  // wave-uniform
            prevp = (uint32_t)__builtin_amdgcn_readlane((int)p, last);
            Gprev = readlane_d(G, last);
            G2prev = readlane_d(G2, last);
        }
        rank += (uint32_t)tot;
        // rebase the carry to the next tile's start
        Gprev = Gprev - tileS;
        G2prev = G2prev - tileS2;
        __syncthreads();
    }
    // exactness guard (see the file header): reads that fail it are redone by k_event_fallback
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)mnb, d, 64), o2 = (uint32_t)__shfl_xor((int)mxb, d, 64);
        mnb = o1 < mnb ? o1 : mnb;
        mxb = o2 > mxb ? o2 : mxb;
    }
    const float mn = (mnb == 0xffffffffu) ? FLT_MAX : __uint_as_float(mnb + 1u), mx = __uint_as_float(mxb);
    const bool flagged = dense || mxb >= 0x7f800000u || !guard_ok(mn, mx, n) || a.flags[r] == 2;
    if (l == 0) {
        a.flags[r] = flagged ? 1 : 0;
        if (flagged) {
            const uint32_t k = atomicAdd(&a.hdr->n_flagged, 1u);
            a.flag_list[k] = r;
        } else {
            store_event(a, slot0, cap, (uint64_t)rank, prevp, (uint32_t)n, 0.0 - Gprev, 0.0 - G2prev, overflow);
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
    bool overflow = false, dense = false;
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

template <int W1, typename T>
__global__ __launch_bounds__(64, 3) void k_event_detect(EvArgs a) {
    __shared__ DetSnap snap;
    __shared__ __attribute__((aligned(16))) char hist[USE_LDS_HISTORY(W1) ? HistRing<T>::LDS_BYTES : 16];
    const uint32_t r = blockIdx.x;
    const ReadCtx<T> rc = make_ctx<T>(a, r);
    const bool ok = detect_read_fast<W1, T, false>(rc, a.hdr, &snap, hist, nullptr);
    if (lane_id() == 0) a.flags[r] = ok ? 0 : 2;  // 2: declined by the fast pass -> exact fallback
}

template <typename T>
__global__ __launch_bounds__(64, 4) void k_event_build(EvArgs a) {
    __shared__ BuildLds L;
    const uint32_t r = blockIdx.x;
    const ReadCtx<T> rc = make_ctx<T>(a, r);
    build_read<T>(a, rc, r, &L);
}

template <int W1, typename T>
__global__ __launch_bounds__(64) void k_event_fallback(EvArgs a) {
    __shared__ PrefixLds L;
    __shared__ DetSnap snap;
    __shared__ EventList events;
    __shared__ __attribute__((aligned(16))) char hist[USE_LDS_HISTORY(W1) ? HistRing<T>::LDS_BYTES : 16];
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
        const bool fast = detect_read_fast<W1, T, true>(rc, a.hdr, &snap, hist, &rep);
        if (!fast) detect_read<W1, T, true>(rc, nullptr, a.hdr);
        __threadfence();
        __syncthreads();
        build_read_prefix<T>(a, rc, r);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- launcher

template <typename T>
static int launch_event_t(const EvArgs &a, int rna, uint32_t n_fb_blocks, hipStream_t st) {
    if (a.n_reads == 0) return SGK_OK;
    // (Tried in round 1: cutting the batch into read slices and running the builder of slice s on a side
    // stream under the detector of slice s+1.  Both kernels contend for VALU issue and the detector needs
    // >= 3072 reads in flight to fill its 12 waves/CU, so the overlapped step was 8.8 ms against 7.8 ms.)
    ProfScope whole("path:event", st);
    SGK_HIP_TRY(hipMemsetAsync(a.hdr, 0, sizeof(EvHeader), st));
    {
        ProfScope ps("k_event_detect", st);
        if (rna) hipLaunchKernelGGL((k_event_detect<7, T>), dim3(a.n_reads), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_event_detect<3, T>), dim3(a.n_reads), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    {
        ProfScope ps("k_event_build", st);
        hipLaunchKernelGGL((k_event_build<T>), dim3(a.n_reads), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    {
        ProfScope ps("k_event_fallback", st);
        if (rna) hipLaunchKernelGGL((k_event_fallback<7, T>), dim3(n_fb_blocks), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_event_fallback<3, T>), dim3(n_fb_blocks), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

unsigned long long debug_exact_redo_count(bool reset) {
    unsigned long long v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_exact_redo_count), sizeof v);
    if (reset) {
        const unsigned long long z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exact_redo_count), &z, sizeof z);
    }
    return v;
}

int launch_event(const EvArgs &a, int rna, bool float_input, uint32_t n_fb_blocks, hipStream_t st) {
    return float_input ? launch_event_t<float>(a, rna, n_fb_blocks, st)
                       : launch_event_t<int16_t>(a, rna, n_fb_blocks, st);
}

}  // namespace sgk
