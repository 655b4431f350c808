// event_fast.h -- the fast path of `event` (round 3): FastPass + the per-event finish.  Included by event_kernels.hip
// (namespace sgk; uses its ReadCtx / Lead16 / lane-mask helpers / LzSnapState / lz_record).
//
// What changed against the round-2 pass (LazyPass, still used by k_event_fallback for flagged reads):
//
//  * DECISIONS, not values.  The t-statistics never leave the kernel; the peak positions do, and they depend on the
//    statistics only through three threshold tests per index (events.c:383-440).  The pass evaluates
//    tq = (t / w)(1 + eps), |eps| <= 2^-20.9, in plain f32 (tstat_math.h: sgk_a3 / sgk_tq; only the roundings that
//    are signal at the scale of a decision are reproduced) and steps the automaton on it with an uncertainty band
//    around every threshold.  A decision inside the band (5e-4 of the indices on nanopore-like data, nearly all of
//    them exact ties between the statistics of two neighbouring indices whose windows hold the same samples) is
//    taken out of line: ties are recognised from the samples, the rest is decided on the reference expression
//    (tstat_exact_at).  oracle/verify_math.cpp (#9..#12) checks the error bound, and the whole scheme against
//    events.c on synthetic reads.
//
//  * NO SECOND SAMPLE WALK, NO BITMAP.  The pass keeps the exact running prefix sums P(i), P2(i) anyway.  When the
//    automaton sets peak_pos it snapshots P at that index (LDS); when the peak is emitted the lane appends one
//    16-byte record {pos, -, (float)(P(pos) - P(prev)), (float)(P2(pos) - P2(prev))} -- create_event's two sums,
//    events.c:457-473 -- to ITS chunk's range of the read's event slots.  After the pass a per-event loop turns the
//    records into events in place (the final index of an event never exceeds its provisional one), one event per
//    lane per round, with coalesced 16-byte loads and stores; the first event of every chunk (it straddles a chunk
//    seam) and the last event of the read are summed from the samples.
//
//  * The long window's bound comes from the float sums of its two short halves (sgk_l2 / sgk_cold2).
//
// Pipeline of one step U of a block (index d = ib + U; the samples run W2 ahead):
//     dstep(d):  automaton on tq(d), not-cold(d), cv-not-ok(d)                 (values produced by earlier steps)
//     tstep(d):  x[d+W2] -> P(d+W2+1);  window position p = d+W1+1: S(p), S2(p) -> s, sq (float), A role of p,
//                tq(p) (with the A role of p-W1);  long window position q = d+1 = p-W1: estimates from s(q), s(p) ->
//                cold(q) against the estimates of q-W2
// A pass starts PRE indices before its first automaton step with tsteps only (the rings fill themselves).
#pragma once

template <int W1>
struct FpCfg {
    static constexpr int W2 = 2 * W1;
    static constexpr int R = 16;
    static constexpr int NP = (W1 == 3) ? 8 : 16;   // prefix ring: P(d) .. P(d+W2+1)
    static constexpr int NR = (W1 == 3) ? 4 : 8;    // rings per short window position (s, sq, A role, tq): W1+1 deep
    static constexpr int NL = (W1 == 3) ? 8 : 16;   // long estimates: W2+1 deep
    static constexpr int H1 = W1 / 2;
    static constexpr int PRE = (W1 == 3) ? 16 : 32; // pre-roll >= 2*W2 + 1 indices, whole blocks
};
static_assert(FpCfg<3>::NR == 3 + 1 && FpCfg<7>::NR == 7 + 1, "tq(p) is produced W1+1 steps before its dstep");
static_assert(FpCfg<3>::PRE > 2 * 6 && FpCfg<7>::PRE > 2 * 14, "pre-roll fills every ring");

constexpr int FP_NREC = 4;  // hot long-detector runs a lane can record per pass (more: read -> exact fallback)

struct __attribute__((aligned(16))) FpPQ {
    double s, q;
};
template <int W1>
struct FpLds {
    // P(k), P2(k) between their last use by the window sums (step k - W1 - 1) and the automaton's step at index k,
    // which takes them if it sets peak_pos there: W1 + 1 steps in LDS instead of 4 registers each
    FpPQ pring[FpCfg<W1>::NR][64];
    // the long window's estimates of position q between their own step and the step of q + W2, where they are the
    // other side of the test: W2 steps in LDS instead of 2 registers each
    SgkL2 lring[FpCfg<W1>::NL][64];
    LzSnapState init[64];        // state a chunk's accepted run started from (at its chunk start; a re-run's start state)
    LzSnapState at_e[64];        // state at the chunk end
    LzRun runs[64][FP_NREC];
    int nrec[64];
    uint32_t cnt[64];            // boundary records written by each lane
};
static_assert(sizeof(FpLds<3>) <= 13 * 1024, "fast pass LDS budget: 12 waves per CU");

typedef short sgk_s2 __attribute__((ext_vector_type(2)));

__device__ inline int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
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

// 8 consecutive samples starting at an even sample offset, as they sit in memory
template <typename T>
struct Lead8;
template <>
struct Lead8<int16_t> {
    uint32_t w[4];
    template <int U>
    __device__ __forceinline__ float get(const Scale &sc) const {
        const int v = (U & 1) ? ((int)w[U / 2] >> 16) : (int)(short)(w[U / 2] & 0xffffu);
        const float shifted = (float)v + sc.offf;
        return shifted * sc.unit;
    }
};
template <>
struct Lead8<float> {
    float w[8];
    template <int U>
    __device__ __forceinline__ float get(const Scale &) const { return w[U]; }
};

// extremes of the samples a lane has seen (exactness guard): packed 16-bit min / max for raw input, the bit
// patterns of |x| for pA input (non-negative floats order like unsigned integers; zero - 1 wraps to the top, so it
// never wins the minimum; inf / nan end up above every finite value and fail the guard)
template <typename T>
struct FpExt;
template <>
struct FpExt<int16_t> {
    sgk_s2 mn, mx;
    __device__ __forceinline__ void init() { mn = sgk_s2{32767, 32767}; mx = sgk_s2{-32768, -32768}; }
    __device__ __forceinline__ void add(const Lead8<int16_t> &g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sgk_s2 w;
            __builtin_memcpy(&w, &g.w[k], 4);
            mn = __builtin_elementwise_min(mn, w);
            mx = __builtin_elementwise_max(mx, w);
        }
    }
    // samples at read-relative positions pos .. pos+7 of which only those inside [0, n) count
    __device__ __forceinline__ void add_masked(const Lead8<int16_t> &g, int pos, int n) {
        short tmp[8];
        __builtin_memcpy(tmp, g.w, sizeof(tmp));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((unsigned)(pos + k) < (unsigned)n) {
                mn.x = tmp[k] < mn.x ? tmp[k] : mn.x;
                mx.x = tmp[k] > mx.x ? tmp[k] : mx.x;
            }
        }
    }
};
template <>
struct FpExt<float> {
    uint32_t mnb, mxb;
    __device__ __forceinline__ void init() { mnb = 0xffffffffu; mxb = 0u; }
    __device__ __forceinline__ void one(float x) {
        const uint32_t ab = __float_as_uint(x) & 0x7fffffffu;
        mxb = ab > mxb ? ab : mxb;
        mnb = (ab - 1u) < mnb ? (ab - 1u) : mnb;
    }
    __device__ __forceinline__ void add(const Lead8<float> &g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) one(g.w[k]);
    }
    __device__ __forceinline__ void add_masked(const Lead8<float> &g, int pos, int n) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((unsigned)(pos + k) < (unsigned)n) one(g.w[k]);
    }
};

// out of line (rare: blocks at a read's ends)
template <typename T>
__device__ __forceinline__ FpExt<T> fp_ext_masked(FpExt<T> x, Lead8<T> g, int pos, int n) {
    x.add_masked(g, pos, n);
    return x;
}

// ---- out of line (rare): a decision inside the uncertainty band that is not a tie ------------------------------
struct FpRes {
    float v, sv;     // the statistics of index i and of the index sv came from, exact, in units of t / w
    uint32_t bits;   // 1: P (ee > 0), 2: Q (ee < -peak_height), 4: v > threshold -- the reference's own tests
};
__device__ unsigned long long g_fp_calls = 0;     // diagnostics: decisions inside the uncertainty band ...
__device__ unsigned long long g_fp_resolved = 0;  // ... of which taken on the reference expression (the others are ties)

// The statistics of index i and of the index sp (where sv was set; any value while sv == FLT_MAX: no peak_value yet)
// by the reference expression, and the reference's three tests on them.  in_peak: the automaton's branch.
// INLINE, with rolled loops: a call from inside the unrolled steps makes the register allocator keep the pass' state
// out of the caller-saved registers, and the spills then land on the hot path (measured: 1-2 scratch round trips per
// step); as plain cold code it costs nothing until it runs (4e-5 of the indices).
template <int W1, typename T>
__device__ __forceinline__ FpRes fp_exact(const T *base, Scale sc, int n, int i, int sp, float sv, bool in_peak) {
    constexpr float ph = DetParam<W1>::ph, thr1 = DetParam<W1>::thr1;
    constexpr float rw = 1.0f / (float)W1;
    const unsigned cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    const bool have_s = __float_as_uint(sv) != 0x7f7fffffu;
#ifdef SGK_DIAG
    atomicAdd(&g_fp_resolved, 1ull);
#endif
    float Tk[2];
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        const int idx = k ? sp : i;
        float t = (k && !have_s) ? FLT_MAX : 0.0f;
        if ((k == 0 || have_s) && (unsigned)(idx - W1) < cnt1) {
            double A = 0.0, A2 = 0.0, B = 0.0, B2 = 0.0;
#pragma unroll 1
            for (int j = 0; j < W1; ++j) {
                const float xa = to_pa(base[idx - W1 + j], sc);
                const float xb = to_pa(base[idx + j], sc);
                A = A + (double)xa;
                A2 = A2 + (double)(xa * xa);
                B = B + (double)xb;
                B2 = B2 + (double)(xb * xb);
            }
            t = sgk_tstat_ref_inl<W1>(A, A2, B, B2);
        }
        Tk[k] = t;
    }
    const float Ti = Tk[0], Ts = Tk[1];
    const float ee = in_peak ? Ti - Ts : Ts - Ti;
    FpRes r;
    r.bits = (ee > 0.0f ? 1u : 0u) | (ee < -ph ? 2u : 0u) | (Ti > thr1 ? 4u : 0u);
    r.v = Ti * rw;
    r.sv = have_s ? Ts * rw : FLT_MAX;
    return r;
}

// out of line (rare): the 8 samples at read-relative positions pos .. pos+7, pos < 0.  Positions in front of the
// read take the value of its first sample (whatever the buffer holds there must not pass through the running prefix
// sums: a value far outside the read's own range would leave a rounding residue in them, see lead_fix_head).
template <typename T>
__device__ __forceinline__ Lead8<T> lead_head(const T *base, int pos, int hi) {
    T tmp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int p = pos + k;
        p = p < 0 ? 0 : p;
        p = p > hi - 1 ? hi - 1 : p;
        tmp[k] = base[p];
    }
    Lead8<T> g;
    __builtin_memcpy(g.w, tmp, sizeof(tmp));
    return g;
}

template <int W1, typename T>
struct FastPass {
    using C = FpCfg<W1>;
    static constexpr int W2 = C::W2, R = C::R, NP = C::NP, NR = C::NR, NL = C::NL, H1 = C::H1, PRE = C::PRE;
    static constexpr float phs = DetParam<W1>::ph / (float)W1, thrs = DetParam<W1>::thr1 / (float)W1;
    // rings (compile-time indices: registers)
    double Ps[NP], Pq[NP];
    float sr[NR], sqr[NR];
    SgkA3 ar[NR];
    SgkL2 la;                     // estimates of the long window position q + 1 - W2 of the NEXT step (read ahead)
    float tr[NR];                 // tq ring; NaN where the variance test failed (floor / tiny / negative / NaN)
    lmask_t hcn;                  // lanes whose long window may exceed thr2 at the NEXT dstep's index
    Lead8<T> cur;                 // x[ib + 8h + W2 .. + 8) of the running half block h
    // short detector (block-relative positions); boolean state as lane masks
    float sv;
    int sp;                       // index at which sv was set (peak_pos while in a peak)
    lmask_t inpk, val, strong;
    lmask_t hist[H1 + 1];         // hist[k]: lanes whose peak_pos was set k+1 indices ago
    // lazy long detector
    int lm, r0;
    lmask_t hot;
    // records
    double pks, pkq;              // prefix sums at the lane's current peak_pos
    double pvs, pvq;              // prefix sums at the lane's previous emitted boundary
    uint32_t roff;                // byte offset of the lane's next record in the read's slot range
    char *recbase;
    FpLds<W1> *L;
    FpPQ *pl;                     // &L->pring[0][lane]
    SgkL2 *ll;                    // &L->lring[0][lane]
    double pds, pdq;              // P(d), P2(d) of the NEXT automaton step, read ahead from the LDS ring
    int own_lo;                   // wave-uniform: block-relative first index of the lanes' own ranges [s, e) ...
    unsigned own_len;             // ... and their length (positions >= n never become peaks)
    // geometry
    const T *base;
    int lo, hi;
    Scale sc;
    int n, s, e, ib;
    unsigned cnt1, cnt2;
    lmask_t done;
    // one step's decision between its two halves (and through the out-of-line exact evaluation)
    float cv;
    lmask_t mP, mQ, mT, mEx;

    __device__ __forceinline__ void load_lead(Lead8<T> &dst, int pos) const {
        if (__builtin_expect(pos < 0, 0)) {  // rare: the pre-roll of a read's first lanes
            dst = lead_head<T>(base, pos, hi);
            return;
        }
        int p = pos > hi - 8 ? hi - 8 : pos;
        p = p < lo ? lo : p;
        constexpr int NV = 8 * (int)sizeof(T) / 16;
        const sgk_u32x4_a4 *src = reinterpret_cast<const sgk_u32x4_a4 *>(base + p);
        sgk_u32x4_a4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = src[k];
        __builtin_memcpy(dst.w, v, sizeof(dst.w));
    }

    // ---- front: sample x[d + W2] -> prefix ring, window position p = d + W1 + 1, long window position q = d + 1
    template <int U, bool SLOW>
    __device__ __forceinline__ void tstep() {
        const float xn = cur.template get<U % 8>(sc);
        const float xq = xn * xn;
        Ps[(U + W2 + 1) % NP] = Ps[(U + W2) % NP] + (double)xn;
        Pq[(U + W2 + 1) % NP] = Pq[(U + W2) % NP] + (double)xq;
        const double S = Ps[(U + W2 + 1) % NP] - Ps[(U + W1 + 1) % NP];
        const double Sq = Pq[(U + W2 + 1) % NP] - Pq[(U + W1 + 1) % NP];
        {
            // P(p): its last use in registers; the automaton's step at index p may want it (peak_pos = p)
            FpPQ pq;
            pq.s = Ps[(U + W1 + 1) % NP];
            pq.q = Pq[(U + W1 + 1) % NP];
            pl[((U + W1 + 1) % NR) * 64] = pq;
        }
        const float s1 = (float)S, sq1 = (float)Sq;
        bool cvok;
        float tq = sgk_tq<W1>(s1, sq1, ar[(U + 1) % NR], cvok);   // A role of p - W1
        ar[(U + W1 + 1) % NR] = sgk_a3<W1>(S, Sq);
        const SgkL2 lb = sgk_l2<W2>(sr[(U + 1) % NR], s1, sqr[(U + 1) % NR], sq1);   // halves q and q + W1 = p
        bool cold = sgk_cold2<W2>(la, lb);   // la: position q - W2, read ahead by the step before
        ll[((U + 1) % NL) * 64] = lb;
        sr[(U + W1 + 1) % NR] = s1;
        sqr[(U + W1 + 1) % NR] = sq1;
        tq = cvok ? tq : __builtin_nanf("");
        if constexpr (SLOW) {
            // the statistics are defined as 0 at the read's first / last w indices (events.c:332-338)
            const bool in1 = (unsigned)(ib + U + 1) < cnt1;            // p - W1 = ib + U + 1
            const bool in2 = (unsigned)(ib + U + 1 - W2) < cnt2;       // q - W2
            tq = in1 ? tq : 0.0f;
            cold = cold || !in2;
        }
        tr[(U + W1 + 1) % NR] = tq;
        hcn = ~__ballot(cold);
        {
            // read ahead: P(d + 1) for the next step's automaton (written W1 steps ago), and the long window
            // estimates of position (q + 1) - W2 (written W2 - 1 steps ago)
            const FpPQ pq = pl[((U + 1) % NR) * 64];
            pds = pq.s;
            pdq = pq.q;
            la = ll[((U + 2 + NL - W2) % NL) * 64];
        }
    }

    // ---- one index of the short detector (events.c:383-440, k = 0), first half: the three tests on the fast
    // statistic, their distance from the thresholds, the ties.  Returns false when some lane's decision has to be
    // taken on the reference expression (mEx): the caller does that out of line and goes on with dstep_b.
    template <int U, bool SLOW>
    __device__ __forceinline__ bool dstep_a(const lmask_t live) {
        constexpr int u = U;
        const float v = tr[U % NR];
        const float d1 = v - sv;
        const float ee = lane_of(inpk) ? d1 : -d1;   // in a peak: v - peak_value; before one: peak_value - v
        lmask_t P = __ballot(ee > 0.0f);             // v > peak_value (in a peak) / v < peak_value (before one)
        lmask_t Q = __ballot(ee < -phs);             // peak_value - v > ph (in a peak) / v - peak_value > ph
        mT = __ballot(v > thrs);
        // distance of the three tests from their thresholds against the error band of the fast statistic
        const float um = __builtin_fminf(__builtin_fminf(__builtin_fabsf(ee), __builtin_fabsf(ee + phs)),
                                         __builtin_fabsf(v - thrs));
        // (NaN -- the variance test failed -- lands here as well; lanes that have left their chunk run on whatever
        // lies behind it and are not asked)
        lmask_t unc = __ballot(!(um > sgk_band(v, phs))) & ~done;
        if constexpr (SLOW) unc &= live;
        cv = v;
        mEx = 0ull;
#ifdef SGK_EXP_NO_RESOLVE
        unc = 0ull;
#endif
        if (__builtin_expect(unc != 0ull, 0)) {
            // Rare (3 % of the wave's steps).  Nearly all of it are exact TIES: the fast values of this index and of
            // the one sv came from are the same float.  Both zero: both statistics ARE zero (tq == 0 <=> delta == 0
            // <=> t == 0).  sv from the index before this one and the three samples at the window seams equal: the
            // windows of i hold the same samples as those of i-1, the reference computes the same float twice.  A tie
            // fails all three tests.  Everything else is decided on the reference expression, out of line.
            bool need = lane_of(unc);
#ifdef SGK_DIAG
            if (need) atomicAdd(&g_fp_calls, 1ull);
#endif
            if (need && v == sv) {   // (false for NaN)
                bool tie = v == 0.0f;
                bool seam = sp == u - 1;
                if constexpr (SLOW) seam = seam && (unsigned)(ib + u - 1 - W1) < cnt1 && (unsigned)(ib + u - W1) < cnt1;
                if (!tie && seam) {
                    const T *x = base + (ib + u);
                    const T xa = x[-W1 - 1], xb = x[-1], xc = x[W1 - 1];
                    tie = xa == xb && xb == xc;
                }
                need = !tie;
            }
            P &= ~unc;
            Q &= ~unc;
            mEx = __ballot(need);
#ifdef SGK_EXP_NO_EXACT
            mEx = 0ull;
#endif
        }
        mP = P;
        mQ = Q;
        return mEx == 0ull;
    }
    // the reference's tests for the lanes of mEx (out of line)
    __device__ __forceinline__ void dstep_exact(const int u) {
        uint32_t rb = 0u;
        if (lane_of(mEx)) {
            const FpRes rr = fp_exact<W1, T>(base, sc, n, ib + u, ib + sp, sv, lane_of(inpk));
            cv = rr.v;
            sv = rr.sv;
            rb = rr.bits;
        }
        mP |= __ballot((rb & 1u) != 0u) & mEx;
        mQ |= __ballot((rb & 2u) != 0u) & mEx;
        mT = (mT & ~mEx) | (__ballot((rb & 4u) != 0u) & mEx);
    }
    // ---- second half: the automaton's transition, the boundary record of an emitted peak, the lazy long
    // detector's bookkeeping
    template <int U, bool SLOW>
    __device__ __forceinline__ void dstep_b(const lmask_t live) {
        constexpr int u = U;
        const float v = cv;
        const lmask_t hck = hcn;
        lmask_t P = mP, Q = mQ;
        const lmask_t Tt = mT;
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
        // (a lane emits peaks of its warm-up, and again behind its chunk while it waits for the others: only
        // positions inside its own range [s, e) are its to record; a lane that is done has none left, and the last
        // lane of a read then runs on whatever lies behind the read)
#ifdef SGK_EXP_NO_RECORDS
        const lmask_t emo = 0ull;
#else
        const lmask_t emo = em & __ballot((unsigned)(sp - own_lo) < own_len) & ~done;
#endif
        if (__builtin_expect(lane_of(emo), 1)) {
            // one boundary record: create_event's two sums as the reference rounds them (events.c:463-468)
            const float dS = (float)(pks - pvs), dSq = (float)(pkq - pvq);
            uint4 rec;
            rec.x = (uint32_t)(ib + sp);
            rec.y = 0u;
            rec.z = __float_as_uint(dS);
            rec.w = __float_as_uint(dSq);
            *reinterpret_cast<uint4 *>(recbase + roff) = rec;
            roff += 16u;
            pvs = pks;
            pvq = pkq;
        }
        if (lane_of(em)) {
            lm = sp;
            r0 = u;
        }
        pks = lane_of(pos) ? pds : pks;   // P(i), P2(i)
        pkq = lane_of(pos) ? pdq : pkq;
        sv = lane_of(upd) ? v : sv;
        sp = lane_of(upd) ? u : sp;
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
        if (__builtin_expect(rec != 0ull, 0)) {
            if (lane_of(rec)) {
                // a hot run [a, b) ended at the reset of index b; the lane whose chunk holds index b replays it (lane
                // c+1 meets the reset at its first index with the state it shares with lane c: exactly one records)
                const int a0 = ib + max(r0, lm + W1 + 1), b0 = ib + u;
                if (b0 >= s && b0 < e) {
                    const int l = lane_id();
                    const int k = L->nrec[l];
                    if (k < FP_NREC) {
                        L->runs[l][k].a = a0;
                        L->runs[l][k].b = b0;
                    }
                    L->nrec[l] = k + 1;
                }
            }
        }
        lmask_t on = __ballot(lm < u - W1);
        if constexpr (SLOW) on &= live;
        hot = (hot & ~dom) | (on & hck & (~dom | em));
    }

    template <int U, bool SLOW>
    __device__ __forceinline__ lmask_t live_of() const {
        if constexpr (SLOW) return ~done & __ballot((unsigned)(ib + U) < (unsigned)n);
        return ~0ull;
    }
    // the half block's samples: the next 8 are fetched while these are consumed
    template <int H>
    __device__ __forceinline__ void half(Lead8<T> &nxt, FpExt<T> &ext, bool lane_edge) {
        if (__builtin_expect(lane_edge, 0)) ext = fp_ext_masked<T>(ext, cur, ib + 8 * H + W2, n);
        else ext.add(cur);
        load_lead(nxt, ib + 8 * (H + 1) + W2);
    }

    // 16 steps without the automaton (pre-roll: the rings fill themselves)
    template <int... Us>
    __device__ __forceinline__ void pre_steps(std::integer_sequence<int, Us...>) {
        (tstep<Us, true>(), ...);
    }
};

// One block of 16 steps with the automaton
template <bool SLOW, int W1, typename T, int... Us>
__device__ __forceinline__ void fp_block_steps(FastPass<W1, T> &f, Lead8<T> &nxt, FpExt<T> &ext, bool lane_edge,
                                               std::integer_sequence<int, Us...>) {
    (([&] {
         if constexpr (Us == 8) {
             f.cur = nxt;
             f.template half<1>(nxt, ext, lane_edge);
         }
         const lmask_t live = f.template live_of<Us, SLOW>();
         if (__builtin_expect(!f.template dstep_a<Us, SLOW>(live), 0)) f.dstep_exact(Us);
         f.template dstep_b<Us, SLOW>(live);
         f.template tstep<Us, SLOW>();
     }()),
     ...);
}
template <bool SLOW, int W1, typename T>
__device__ __forceinline__ void fp_block(FastPass<W1, T> &f, Lead8<T> &nxt, FpExt<T> &ext, bool lane_edge) {
    fp_block_steps<SLOW, W1, T>(f, nxt, ext, lane_edge, std::make_integer_sequence<int, 16>{});
}

// One pass of the fast detector over the wave's chunks.
//   first  : the first pass (every lane starts from the fresh state: true for lane 0, speculative for the others);
//            otherwise a re-run of the lanes whose speculation failed, from L->st0
//   lead   : this lane's warm-up before its chunk start s (0 for lane 0 and in re-runs)
//   steps  : automaton steps every lane runs (wave-uniform: warm-up + chunk length), after the pre-roll
//   active : whether this lane runs in this pass
//   roff0  : byte offset of the lane's record range
//   lead, steps and own_len (the chunk length K) are wave-uniform
template <int W1, typename T>
__device__ __forceinline__ void pass_fast(const ReadCtx<T> &rc, bool first, int lead, int steps, int own_len,
                                          bool active, int s, int e, FpLds<W1> *L, char *recbase, uint32_t roff0,
                                          FpExt<T> &ext) {
    using FP = FastPass<W1, T>;
    constexpr int R = FP::R, PRE = FP::PRE, W2 = FP::W2;
    if (!__any(active)) return;
    const int l = lane_id();
    FP f;
    f.n = (int)rc.n;
    // lanes that do not take part in the pass own nothing: nothing they emit or record can land anywhere
    f.s = active ? s : 0x7fffffff;
    f.e = active ? e : 0x7fffffff;
    f.sc = rc.sc;
    f.base = rc.base;
    f.lo = (int)(rc.lo < -(1 << 30) ? -(1 << 30) : rc.lo);
    f.hi = (int)(rc.hi > 0x7fffffffLL ? 0x7fffffffLL : rc.hi);
    f.L = L;
    f.pl = &L->pring[0][l];
    f.ll = &L->lring[0][l];
    f.la.m = 0.0f;
    f.la.z = 0.0f;
    f.pds = 0.0;
    f.pdq = 0.0;
    f.recbase = recbase;
    f.roff = roff0;
    f.pvs = 0.0;
    f.pvq = 0.0;
    f.pks = 0.0;
    f.pkq = 0.0;
    f.cv = 0.0f;
    f.mP = f.mQ = f.mT = f.mEx = 0ull;
    f.own_lo = PRE + lead;  // = s, relative to the pass' first block (wave-uniform)
    f.own_len = (unsigned)own_len;
    const int n = f.n;
    const int i_begin = s - lead - PRE;  // multiple of 16; negative for the read's first lanes (their dsteps start at >= 0)
#pragma unroll
    for (int k = 0; k < FP::NP; ++k) { f.Ps[k] = 0.0; f.Pq[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < FP::NR; ++k) {
        f.sr[k] = 0.0f; f.sqr[k] = 0.0f; f.ar[k].mean1 = 0.0f; f.ar[k].va3 = 0.0f; f.tr[k] = 0.0f;
    }
    f.hcn = 0ull;
    f.load_lead(f.cur, i_begin + W2);
    // detector state
    f.sv = FLT_MAX;
    f.sp = 0;
    f.inpk = 0ull; f.val = 0ull; f.strong = 0ull; f.hot = 0ull;
#pragma unroll
    for (int k = 0; k <= FP::H1; ++k) f.hist[k] = 0ull;
    f.lm = LZ_NONE;
    f.r0 = PRE;  // the (pseudo) reset a speculative pass starts from (block-relative: its first automaton step)
    if (!first) {
        const LzSnapState st = L->init[l];
        const int i0 = i_begin + PRE;  // first automaton step; positions below are relative to the pass' first block
        f.sv = st.sv;
        f.inpk = __ballot((st.bits & 1u) != 0u);
        f.val = __ballot((st.bits & 2u) != 0u);
        f.strong = __ballot((st.bits & 4u) != 0u);
        f.hot = __ballot((st.bits & 8u) != 0u);
        f.sp = st.sp - i_begin;
        f.lm = st.lm == LZ_NONE ? LZ_NONE : st.lm - W1 - i_begin;  // handed over as masked_to; kept as the peak position
#pragma unroll
        for (int k = 0; k <= FP::H1; ++k) f.hist[k] = __ballot((st.bits & 1u) && st.sp == i0 - 1 - k);
        f.r0 = st.r0 - i_begin;
    }
    if (active) L->nrec[l] = 0;
    f.done = ~__ballot(active);
    f.cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    f.cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    const int main_steps = PRE + steps;

    auto snapshot = [&](int nb) -> LzSnapState {
        // nb: absolute index of the block about to start (positions are relative to it)
        LzSnapState st;
        const bool ip = lane_of(f.inpk);
        st.sv = f.sv;
        st.sp = (__float_as_uint(f.sv) == 0x7f7fffffu) ? -1 : nb + f.sp;
        st.lm = (f.lm + W1 < 0) ? LZ_NONE : nb + f.lm + W1;  // normalised when it no longer masks
        st.r0 = nb + f.r0;
        st.bits = (ip ? 1u : 0u) | ((ip && lane_of(f.val)) ? 2u : 0u) | ((ip && lane_of(f.strong)) ? 4u : 0u) |
                  (lane_of(f.hot) ? 8u : 0u);
        return st;
    };

    int jb = 0;
    for (;;) {
        const int ib = i_begin + jb;
        f.ib = ib;
        // blocks that touch the read's first / last W2 indices or its end take the predicated forms of the steps
        const bool lane_edge = (ib + 1 < W2) || (ib + R > n - W2);
        const bool slow = (__ballot(lane_edge) & ~f.done) != 0ull;
        // the samples of a half block are fetched while the half block before it runs
        Lead8<T> nxt;
        f.template half<0>(nxt, ext, lane_edge);
        if (jb < PRE) {
            f.pre_steps(std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>{});
            f.cur = nxt;
            f.template half<1>(nxt, ext, lane_edge);
            f.pre_steps(std::integer_sequence<int, 8, 9, 10, 11, 12, 13, 14, 15>{});
        } else if (slow) {
            fp_block<true, W1, T>(f, nxt, ext, lane_edge);
        } else {
            fp_block<false, W1, T>(f, nxt, ext, lane_edge);
        }
        f.cur = nxt;
        // rebase the block-relative positions
        f.sp -= R;
        f.own_lo -= R;
        f.lm = f.lm < LZ_NONE ? LZ_NONE : f.lm - R;
        f.r0 -= R;
        jb += R;
        {
            // state snapshots live in LDS (they are only needed after the pass)
            const int nb = i_begin + jb;  // first index of the next block
            if (active && first && lead > 0 && jb == PRE + lead) L->init[l] = snapshot(nb);
            if (active && nb == e) L->at_e[l] = snapshot(nb);
            const bool pend = lane_of(f.inpk) && (nb + f.sp) < e;
            // the reference's loop ends at n-1: peaks still pending there are dropped
            f.done |= __ballot(jb >= PRE && nb >= e && (!pend || nb >= n));
        }
        if (jb >= main_steps && f.done == ~0ull) break;
    }
    // the run still open at the end of the read is replayed by the lane that holds the read's last index
    if (active && lane_of(f.hot) && e == n && s < n) {
        const int k = L->nrec[l];
        if (k < FP_NREC) {
            L->runs[l][k].a = i_begin + jb + max(f.r0, f.lm + W1 + 1);
            L->runs[l][k].b = n;
        }
        L->nrec[l] = k + 1;
    }
    if (active) L->cnt[l] = (f.roff - roff0) >> 4;
}

// Exact replay of the long detector over the recorded hot runs (as replay_long_runs): the fast path only needs to
// know WHETHER it emits -- on nanopore data it never does -- and hands such a read to the exact fallback.
template <int W1, typename T>
__device__ bool replay_long_emits(const ReadCtx<T> &rc, FpLds<W1> *L, bool active) {
    constexpr int W2 = 2 * W1;
    constexpr float ph = DetParam<W1>::ph, thr2 = DetParam<W1>::thr2;
    const int l = lane_id();
    const int n = (int)rc.n;
    const unsigned cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    const int nrec = active ? L->nrec[l] : 0;
    bool emitted = false;
    for (int k = 0; k < FP_NREC; ++k) {
        const bool has = k < nrec;
        if (!__any(has)) break;
        int i = has ? L->runs[l][k].a : 0;
        const int b = has ? L->runs[l][k].b : 0;
        int lp = -1;
        float lv = FLT_MAX;
        bool lvalid = false;
        while (__any(has && i < b)) {
            if (has && i < b) {
                float v2 = 0.0f;
                if ((unsigned)(i - W2) < cnt2) v2 = tstat_exact_at<T>(rc.base, rc.sc, i, W2);
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
                        if (lp > 0 && lp < n) emitted = true;
                        lp = -1;
                        lv = v2;
                        lvalid = false;
                    }
                }
                ++i;
            }
        }
    }
    return __any(emitted);
}

// Chunk layout: lane c owns the indices [cK, (c+1)K) and warms up, from the fresh state, over the `lead` indices in
// front of them (lane 0: indices in front of the read, on which the automaton does not step).  A range of K indices
// holds at most K/3 + 1 boundaries (emitted peaks of the short detector are at least 3 apart): lane c's records go to
// the slots [c (K/3 + 1), ...) of the read.
__device__ inline int chunk_len_uniform(int n) {
    const int k = (n + 1023) / 1024;
    return 16 * (k < 1 ? 1 : k);
}
__device__ __forceinline__ uint32_t fp_chunk_slot(int c, int K) { return (uint32_t)(c * (K / 3 + 1)); }

// speculative pass + verification / re-run loop + replay of the hot long-detector runs.
// Returns 0 when the read's boundary records are in place, 1 when the fast pass cannot take it (alignment / room around
// the read / too few event slots), 2 when a lane met more hot runs than it can record (pathological signal: constant
// stretches, tiny variances), 3 when the long detector emits a peak.
template <int W1, typename T>
__device__ int detect_read_fast(const EvArgs &a, const ReadCtx<T> &rc, uint32_t r, FpLds<W1> *L, FpExt<T> &ext, int &K_out) {
    const int n = (int)rc.n;
    int lead = n < 32768 ? SGK_LEAD_DNA_SHORT : SGK_LEAD_DNA;
    if (W1 == 7) lead = n <= 32768 ? SGK_LEAD_RNA_SHORT : SGK_LEAD_RNA;
    const int K = chunk_len_uniform(n);
    K_out = K;
    // the fast pass uses unguarded 4-byte-aligned 16-byte vector loads: it needs 16 readable samples behind the
    // read; other reads take the exact fallback
    if ((reinterpret_cast<uintptr_t>(rc.base) & 3u) != 0 || rc.hi < (int64_t)n + 16) return 1;
    const int c = lane_id();
    const int s = c * K;
    const int e0 = s + K;
    const int e = e0 < n ? e0 : n;
    const bool active = s < n;
    // the read's slot range must hold every chunk's records (and the read's last event)
    const uint64_t slot0 = a.ev_slots[r], cap = a.ev_slots[r + 1] - slot0;
    const uint32_t q = fp_chunk_slot(c, K);
    const uint32_t my_end = active ? q + (uint32_t)((e - s) / 3 + 1) : 0u;
    uint32_t need = my_end;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)need, d, 64);
        need = o > need ? o : need;
    }
    if ((uint64_t)need + 1 > cap) return 1;
    char *recbase = reinterpret_cast<char *>(a.events + slot0);
    {
        LzSnapState z;
        z.sp = -1; z.sv = FLT_MAX; z.lm = LZ_NONE; z.r0 = 0; z.bits = 0u;
        L->init[c] = z;
        L->at_e[c] = z;
        L->nrec[c] = 0;
        L->cnt[c] = 0u;
    }
    bool run = active;
    bool first = true;
    for (int iter = 0; iter < 66; ++iter) {
        pass_fast<W1, T>(rc, first, first ? lead : 0, first ? lead + K : K, K, run, s, e, L, recbase, q * 16u, ext);
        __syncthreads();
        // chunk c is right iff it started (at s) from the state chunk c-1 ended with
        const LzSnapState pe = L->at_e[c > 0 ? c - 1 : 0];
        const LzSnapState mine = L->init[c];
        const bool bad = active && c > 0 && !lz_equal(pe, mine);
        const unsigned long long badmask = __ballot(bad);
        if (badmask == 0ull) break;
        __syncthreads();
        if (bad) L->init[c] = pe;
        run = bad;
        first = false;
        if (c == 0) atomicAdd(&a.hdr->n_rerun, (uint32_t)__popcll(badmask));
        __syncthreads();
    }
    if (__any(active && L->nrec[c] > FP_NREC)) return 2;
    const unsigned long long hotm = __ballot(active && L->nrec[c] > 0);
    if (hotm != 0ull) {
        if (c == 0) atomicAdd(&a.hdr->n_hot_runs, (uint32_t)__popcll(hotm));
        if (replay_long_emits<W1, T>(rc, L, active)) return 3;
    }
    return 0;
}

// create_event (events.c:457-473): the two divisions by the event length share one refined reciprocal
// (tstat_math.h: bit-identical to `/` inside the range guard)
__device__ __forceinline__ uint4 fp_make_event(uint32_t ps, uint32_t pe, float dsum, float dsumsq) {
    const float len = (float)(pe - ps);
    const float r1 = sgk_refined_rcp(len);
    const float m = sgk_div_with_rcp(dsum, len, r1);
    const float var = sgk_div_with_rcp(dsumsq, len, r1) - m * m;
    const float sd = sqrtf(fmaxf(var, 0.0f));
    uint4 ev;
    ev.x = ps;
    ev.y = pe - ps;
    ev.z = __float_as_uint(m);
    ev.w = __float_as_uint(sd);
    return ev;
}

// exact sums of x and fl(x*x) over the samples [a, b) of a read, by the whole wave (any order gives the reference's
// prefix difference: the exactness guard of this path)
template <typename T>
__device__ inline void fp_wave_sums(const ReadCtx<T> &rc, int a0, int b0, double &S, double &Sq) {
    double s = 0.0, q = 0.0;
    for (int t = a0 + lane_id(); t < b0; t += 64) {
        const float x = to_pa(rc.base[t], rc.sc);
        s = s + (double)x;
        q = q + (double)(x * x);
    }
    S = wave_last_d(wave_incl_scan_d(s));
    Sq = wave_last_d(wave_incl_scan_d(q));
}

// The per-event pass: boundary records (chunk by chunk, at their provisional slots) -> the read's events, in place.
template <int W1, typename T>
__device__ void finish_read(const EvArgs &a, const ReadCtx<T> &rc, uint32_t r, FpLds<W1> *L, int K) {
    const int n = (int)rc.n;
    const int c = lane_id();
    const uint64_t slot0 = a.ev_slots[r];
    uint4 *ev = reinterpret_cast<uint4 *>(a.events + slot0);
    const uint32_t cnt = L->cnt[c];
    const uint32_t q = fp_chunk_slot(c, K);
    const int incl = wave_incl_scan_i((int)cnt);
    const uint32_t rank0 = (uint32_t)incl - cnt;           // events before this chunk's first boundary
    const uint32_t total = (uint32_t)wave_last_i(incl);    // boundaries of the read
    // first / last boundary of every chunk; the boundary in front of a chunk's first one
    uint32_t p0 = 0u, plast = 0u;
    if (cnt > 0u) {
        p0 = ev[q].x;
        plast = ev[q + cnt - 1u].x;
    }
    uint32_t run = plast;  // inclusive running maximum over the lanes (positions grow with the chunk index)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)run, d, 64);
        if (c >= d) run = o > run ? o : run;
    }
    uint32_t pprev = (uint32_t)__shfl_up((int)run, 1, 64);
    if (c == 0) pprev = 0u;
    const uint32_t last_boundary = (uint32_t)__builtin_amdgcn_readlane((int)run, 63);
    // the event that ends at a chunk's first boundary straddles the chunk seam: its sums come from the samples
    uint4 seam = uint4{0u, 0u, 0u, 0u};
    {
        const int len = cnt > 0u ? (int)(p0 - pprev) : 0;
        const int maxlen = wave_max_i(len);
        if (maxlen <= 512) {
            double S = 0.0, Sq = 0.0;
            for (int t = 0; t < maxlen; ++t) {
                if (t < len) {
                    const float x = to_pa(rc.base[pprev + t], rc.sc);
                    S = S + (double)x;
                    Sq = Sq + (double)(x * x);
                }
            }
            if (cnt > 0u) seam = fp_make_event(pprev, p0, (float)S, (float)Sq);
        } else {
            // a long stretch without a peak in front of some chunk: the wave sums those one at a time
            unsigned long long todo = __ballot(cnt > 0u);
            while (todo) {
                const int cc = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int a0 = __builtin_amdgcn_readlane((int)pprev, cc), b0 = __builtin_amdgcn_readlane((int)p0, cc);
                double S, Sq;
                fp_wave_sums<T>(rc, a0, b0, S, Sq);
                if (c == cc) seam = fp_make_event((uint32_t)a0, (uint32_t)b0, (float)S, (float)Sq);
            }
        }
    }
    // the read's last event [last boundary, n)
    uint4 tail;
    {
        double S, Sq;
        fp_wave_sums<T>(rc, (int)last_boundary, n, S, Sq);
        tail = fp_make_event(last_boundary, (uint32_t)n, (float)S, (float)Sq);
    }
    // chunk by chunk, 64 boundaries per round: boundary j of a chunk closes the event [boundary j-1, boundary j)
    unsigned long long chunks = __ballot(cnt > 0u);
    while (chunks) {
        const int cc = __ffsll((long long)chunks) - 1;
        chunks &= chunks - 1ull;
        const uint32_t ccnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, cc);
        const uint32_t cq = (uint32_t)__builtin_amdgcn_readlane((int)q, cc);
        const uint32_t crank = (uint32_t)__builtin_amdgcn_readlane((int)rank0, cc);
        uint4 sm;
        sm.x = (uint32_t)__builtin_amdgcn_readlane((int)seam.x, cc);
        sm.y = (uint32_t)__builtin_amdgcn_readlane((int)seam.y, cc);
        sm.z = (uint32_t)__builtin_amdgcn_readlane((int)seam.z, cc);
        sm.w = (uint32_t)__builtin_amdgcn_readlane((int)seam.w, cc);
        // FB rounds of 64 boundaries in flight (the loads of all of them before the first event is formed)
        constexpr int FB = 4;
        uint32_t carry = 0u;  // position of the boundary in front of this group's first
        for (uint32_t j0 = 0u; j0 < ccnt; j0 += 64u * FB) {
            uint4 rec[FB];
#pragma unroll
            for (int g = 0; g < FB; ++g) {
                const uint32_t j = j0 + 64u * g + (uint32_t)c;
                rec[g] = ev[cq + (j < ccnt ? j : ccnt - 1u)];
            }
#pragma unroll
            for (int g = 0; g < FB; ++g) {
                const uint32_t jg = j0 + 64u * g;
                if (jg < ccnt) {   // wave-uniform
                    const uint32_t j = jg + (uint32_t)c;
                    const uint32_t pp = (uint32_t)wave_shr1_i((int)rec[g].x, (int)carry);
                    uint4 out = fp_make_event(pp, rec[g].x, __uint_as_float(rec[g].z), __uint_as_float(rec[g].w));
                    if (j == 0u) out = sm;
                    const int lastl = (ccnt - jg) < 64u ? (int)(ccnt - jg - 1u) : 63;  // wave-uniform
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)rec[g].x, lastl);
                    if (j < ccnt) ev[crank + j] = out;
                }
            }
        }
    }
    if (c == 0) {
        ev[total] = tail;
        a.n_events[r] = total + 1u;
        atomicAdd(&a.hdr->n_events_total, (unsigned long long)(total + 1u));
    }
}
