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
//  * Emitted peaks go to the read's bitmap as in round 2 (one OR of a wave-uniform bit into a register word for
//    the usual peak, a 512-position ring in LDS for older ones) and k_event's second half (build_read) cuts the
//    events.  A per-event builder was built and measured in this round (the lanes appended 16-byte boundary records
//    {pos, sum, sumsq} chunk by chunk to the read's event slots, a per-event pass compacted them in place): bit-exact,
//    98 instead of 119 vector instructions per sample, and SLOWER -- 6.2 ms against 3.8: 1.9e8 scattered 16-byte
//    stores per step leave L2 as partial-line requests (TCC_EA0_WRREQ 2.4e8, half of them not 64 bytes) and the
//    records are read back once more, 11 GB of HBM traffic instead of 7.5 (profiles/r03_records_experiment.md).
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

constexpr int FP_RING_WORDS = 16;  // per-lane bitmap ring: 512 positions
template <int W1>
struct FpLds {
    uint32_t ring[64][FP_RING_WORDS];
    // the long window's estimates of position q between their own step and the step of q + W2, where they are the
    // other side of the test: W2 steps in LDS instead of 2 registers each
    SgkL2 lring[FpCfg<W1>::NL][64];
    LzSnapState init[64];        // state a chunk's accepted run started from (at its chunk start; a re-run's start state)
    LzSnapState at_e[64];        // state at the chunk end
    LzRun runs[64][FP_NREC];
    int nrec[64];
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
    double Ps[NP], Pq[NP];        // live: P(d+W1+2) .. P(d+W2+1)
    float sr[NR], sqr[NR];
    SgkA3 ar[NR];
#ifdef SGK_LR_REGS
    SgkL2 lr[NL];
#endif
    SgkL2 la;                     // estimates of the long window position q + 1 - W2 of the NEXT step (read ahead)
    float tr[NR];                 // tq ring; NaN where the variance test failed (floor / tiny / negative / NaN)
    lmask_t hcn;                  // lanes whose long window may exceed thr2 at the NEXT dstep's index
    Lead8<T> cur;                 // x[ib + 8h + W2 .. + 8) of the running half block h
    // short detector (block-relative positions); boolean state as lane masks
    float sv;
    int sp;                       // index at which sv was set (peak_pos while in a peak)
    lmask_t inpk, val, strong;
    lmask_t hist[H1 + 1];         // hist[k]: lanes whose peak_pos was set k+1 indices ago
    uint32_t bw;                  // bitmap word (32 positions) that holds position j - H1 - 1, the usual emitted peak
    // lazy long detector
    int lm, r0;
    lmask_t hot;
    // geometry
    FpLds<W1> *L;
    uint32_t *ring;               // this lane's bitmap ring in LDS (positions relative to the pass' first block)
    SgkL2 *ll;                    // &L->lring[0][lane]
    unsigned long long *bm;       // read's bitmap (global)
    const T *base;
    int lo, hi;
    Scale sc;
    int n, s, e, ib, jb, i_begin, flushed;
    unsigned cnt1, cnt2;
    lmask_t done;
    bool oldpeak;                 // wave-uniform: some lane's peak may lie outside the bitmap ring
    // one step's decision between its two halves
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
        const float s1 = (float)S, sq1 = (float)Sq;
        bool cvok;
        float tq = sgk_tq<W1>(s1, sq1, ar[(U + 1) % NR], cvok);   // A role of p - W1
        ar[(U + W1 + 1) % NR] = sgk_a3<W1>(S, Sq);
        const SgkL2 lb = sgk_l2<W2>(sr[(U + 1) % NR], s1, sqr[(U + 1) % NR], sq1);   // halves q and q + W1 = p
#ifdef SGK_LR_REGS
        bool cold = sgk_cold2<W2>(lr[(U + 1 + NL - W2) % NL], lb);
        lr[(U + 1) % NL] = lb;
#else
        bool cold = sgk_cold2<W2>(la, lb);   // la: position q - W2, read ahead by the step before
        ll[((U + 1) % NL) * 64] = lb;
#endif
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
#ifndef SGK_LR_REGS
        // read ahead: the long window estimates of position (q + 1) - W2 (written W2 - 1 steps ago)
        la = ll[((U + 2 + NL - W2) % NL) * 64];
#endif
    }

    // ---- one index of the short detector (events.c:383-440, k = 0), first half: the three tests on the fast
    // statistic, their distance from the thresholds, the ties; the rest on the reference expression (cold, inline)
    template <int U, bool SLOW>
    __device__ __forceinline__ void dstep_a(const lmask_t live) {
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
        if (__builtin_expect(unc != 0ull, 0)) {
            // Rare (3 % of the wave's steps).  Nearly all of it are exact TIES: the fast values of this index and of
            // the one sv came from are the same float.  Both zero: both statistics ARE zero (tq == 0 <=> delta == 0
            // <=> t == 0).  sv from the index before this one and the three samples at the window seams equal: the
            // windows of i hold the same samples as those of i-1, the reference computes the same float twice.  A tie
            // fails all three tests.  Everything else (4e-5 of the indices) is decided on the reference expression.
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
            if (mEx != 0ull) {
                uint32_t rb = 0u;
                if (need) {
                    const FpRes rr = fp_exact<W1, T>(base, sc, n, ib + u, ib + sp, sv, lane_of(inpk));
                    cv = rr.v;
                    sv = rr.sv;
                    rb = rr.bits;
                }
                P |= __ballot((rb & 1u) != 0u) & mEx;
                Q |= __ballot((rb & 2u) != 0u) & mEx;
                mT = (mT & ~mEx) | (__ballot((rb & 4u) != 0u) & mEx);
            }
        }
        mP = P;
        mQ = Q;
    }
    // the bitmap word moves on when position j - H1 - 1 enters the next 32-position span
    __device__ __forceinline__ void bw_advance() {
        const int p = jb - 1;  // last position of the span that is complete (jb is a multiple of 32 here)
        atomicOr(&ring[(p >> 5) & (FP_RING_WORDS - 1)], bw);
        bw = 0u;
    }
    // ---- second half: the automaton's transition, the emitted peak's bit, the lazy long detector's bookkeeping
    template <int U, bool SLOW>
    __device__ __forceinline__ void dstep_b(const lmask_t live) {
        constexpr int u = U;
        if constexpr (U == H1 + 1) {
            if ((jb & 16) == 0) bw_advance();
        }
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
        // Emission.  The usual emitted peak was set exactly H1+1 indices ago: its position is the same in every lane,
        // and so is its bit in the lane's current bitmap word.  A strong peak stays strong and in a peak until it is
        // emitted, so the emission step is the LAST step at which this peak resets the long detector: masked_to and
        // the reset index are taken here.
        const uint32_t bit = 1u << ((jb + u - H1 - 1) & 31);
        if (SLOW && oldpeak) {
            // some lane holds a peak older than the bitmap ring reaches (or one from before the pass)
            if (lane_of(em)) {
                const int p = jb + sp;
                if (p >= flushed && p >= 0) {
                    atomicOr(&ring[(p >> 5) & (FP_RING_WORDS - 1)], 1u << (p & 31));
                } else {
                    const int pa = i_begin + p;
                    if (pa >= s && pa < e) atomicOr(reinterpret_cast<uint32_t *>(bm) + (pa >> 5), 1u << (pa & 31));
                }
            }
        } else {
            if (lane_of(em)) bw |= bit;
            const lmask_t erare = em & ~hist[H1];
            if (__builtin_expect(erare != 0ull, 0)) {
                if (lane_of(erare)) {  // an older peak: undo the bit, set the right one (its word is in the ring)
                    bw &= ~bit;
                    const int p = jb + sp;
                    atomicOr(&ring[(p >> 5) & (FP_RING_WORDS - 1)], 1u << (p & 31));
                }
            }
        }
        if (lane_of(em)) {
            lm = sp;
            r0 = u;
        }
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

    // 8 steps without the automaton (pre-roll: the rings fill themselves)
    template <int... Us>
    __device__ __forceinline__ void pre_steps(std::integer_sequence<int, Us...>) {
        (tstep<Us, true>(), ...);
    }
};

// One block of 16 steps with the automaton; the samples of a half block are fetched while the half block before it runs
template <bool SLOW, int W1, typename T, int... Us>
__device__ __forceinline__ void fp_block_steps(FastPass<W1, T> &f, Lead8<T> &nxt, std::integer_sequence<int, Us...>) {
    (([&] {
         if constexpr (Us == 8) {
             f.cur = nxt;
             f.load_lead(nxt, f.ib + 16 + FastPass<W1, T>::W2);
         }
         const lmask_t live = f.template live_of<Us, SLOW>();
         f.template dstep_a<Us, SLOW>(live);
         f.template dstep_b<Us, SLOW>(live);
         f.template tstep<Us, SLOW>();
     }()),
     ...);
}

// flush 8 ring words (the 256 positions starting at pass-relative position p0, a multiple of 256) of this lane to
// the read's bitmap, in 16-bit units: i_begin is a multiple of 16, so pass-relative units are the bitmap's units, and
// only units inside the lane's own range [own_lo, own_hi) (pass-relative; multiples of 16, or the read's end) are
// written -- every owned unit is written exactly once per pass, zero or not
__device__ __forceinline__ void fp_flush(uint32_t *ring, unsigned long long *bm, int i_begin, int p0, int own_lo,
                                         int own_hi) {
    uint16_t *bm16 = reinterpret_cast<uint16_t *>(bm);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = p0 + 32 * k;
        const int wi = (p >> 5) & (FP_RING_WORDS - 1);
        const uint32_t w = ring[wi];
        ring[wi] = 0u;
        if (p >= own_lo && p < own_hi) bm16[(i_begin + p) >> 4] = (uint16_t)(w & 0xffffu);
        if (p + 16 >= own_lo && p + 16 < own_hi) bm16[(i_begin + p + 16) >> 4] = (uint16_t)(w >> 16);
    }
}

// One pass of the fast detector over the wave's chunks.
//   first  : the first pass (every lane starts from the fresh state: true for lane 0, speculative for the others);
//            otherwise a re-run of the lanes whose speculation failed, from L->init
//   lead   : the lanes' warm-up before their chunk start s (0 in re-runs); steps: automaton steps every lane runs
//            (warm-up + chunk length), after the pre-roll -- both wave-uniform
//   active : whether this lane runs in this pass
// Writes the lane's bitmap units, its hot-run records and (speculative pass) L->init / L->at_e.
template <int W1, typename T>
__device__ __forceinline__ void pass_fast(const ReadCtx<T> &rc, bool first, int lead, int steps, bool active, int s,
                                          int e, FpLds<W1> *L) {
    using FP = FastPass<W1, T>;
    constexpr int R = FP::R, PRE = FP::PRE, W2 = FP::W2;
    if (!__any(active)) return;
    const int l = lane_id();
    FP f;
    f.n = (int)rc.n;
    // lanes that do not take part in the pass own nothing: nothing they emit or record can land anywhere
    f.s = active ? s : 0x7fffffff;
    f.e = active ? e : 0x7fffffff;
    f.bm = rc.bm;
    f.sc = rc.sc;
    f.base = rc.base;
    f.lo = (int)(rc.lo < -(1 << 30) ? -(1 << 30) : rc.lo);
    f.hi = (int)(rc.hi > 0x7fffffffLL ? 0x7fffffffLL : rc.hi);
    f.L = L;
    f.ring = L->ring[l];
    f.ll = &L->lring[0][l];
    f.la.m = 0.0f;
    f.la.z = 0.0f;
#ifdef SGK_LR_REGS
#pragma unroll
    for (int k = 0; k < FP::NL; ++k) { f.lr[k].m = 0.0f; f.lr[k].z = 0.0f; }
#endif
    f.cv = 0.0f;
    f.mP = f.mQ = f.mT = f.mEx = 0ull;
    const int n = f.n;
    const int i_begin = s - lead - PRE;  // multiple of 16; negative for the read's first lanes (their dsteps start at >= 0)
    f.i_begin = i_begin;
#pragma unroll
    for (int k = 0; k < FP_RING_WORDS; ++k) f.ring[k] = 0u;
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
    f.bw = 0u;
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
    f.flushed = 0;
    f.cnt1 = (n - 2 * W1 + 1) > 0 ? (unsigned)(n - 2 * W1 + 1) : 0u;
    f.cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    const int main_steps = PRE + steps;
    // pass-relative range of the positions this lane owns (its bitmap units)
    const int own_lo = PRE + lead, own_hi = active ? own_lo + (e - s) : own_lo;

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
        f.jb = jb;
        if ((jb & 255) == 0 && jb >= 512) {
            fp_flush(f.ring, f.bm, i_begin, jb - 512, own_lo, own_hi);
            f.flushed = jb - 256;
        }
        // blocks that touch the read's first / last W2 indices or its end take the predicated forms of the steps; so
        // does a block in which some lane holds a peak older than the bitmap ring
        const bool lane_edge = (ib + 1 < W2) || (ib + R > n - W2);
        const bool old_peak = f.sp < -(256 - 2 * R) || f.sp + jb < 0;
        f.oldpeak = (__ballot(old_peak) & f.inpk & ~f.done) != 0ull;
        const bool slow = f.oldpeak || (__ballot(lane_edge) & ~f.done) != 0ull;
        // the samples of a half block are fetched while the half block before it runs
        Lead8<T> nxt;
        f.load_lead(nxt, ib + 8 + W2);
        if (jb < PRE) {
            f.pre_steps(std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>{});
            f.cur = nxt;
            f.load_lead(nxt, ib + 16 + W2);
            f.pre_steps(std::integer_sequence<int, 8, 9, 10, 11, 12, 13, 14, 15>{});
        } else if (slow) {
            fp_block_steps<true, W1, T>(f, nxt, std::make_integer_sequence<int, 16>{});
        } else {
            fp_block_steps<false, W1, T>(f, nxt, std::make_integer_sequence<int, 16>{});
        }
        f.cur = nxt;
        // rebase the block-relative positions
        f.sp -= R;
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
    // remaining ring words (the current bitmap word first)
    {
        const int p = jb - FP::H1 - 2 < 0 ? 0 : jb - FP::H1 - 2;  // a position inside the word bw stands for
        atomicOr(&f.ring[(p >> 5) & (FP_RING_WORDS - 1)], f.bw);
    }
    for (int p0 = f.flushed; p0 < jb; p0 += 256) fp_flush(f.ring, f.bm, i_begin, p0, own_lo, own_hi);
}

// Exact replay of the long detector (events.c:383-440, k = 1) over the recorded hot runs: from the fresh state a
// reset leaves, over the indices of the run (inside a run masked_to does not change and every index is processed).
template <int W1, typename T>
__device__ void fp_replay_long(const ReadCtx<T> &rc, FpLds<W1> *L, bool active) {
    constexpr int W2 = 2 * W1;
    constexpr float ph = DetParam<W1>::ph, thr2 = DetParam<W1>::thr2;
    const int l = lane_id();
    const int n = (int)rc.n;
    const unsigned cnt2 = (n - 2 * W2 + 1) > 0 ? (unsigned)(n - 2 * W2 + 1) : 0u;
    const int nrec = active ? L->nrec[l] : 0;
    uint32_t *bm32 = reinterpret_cast<uint32_t *>(rc.bm);
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
                        if (lp > 0 && lp < n) atomicOr(&bm32[lp >> 5], 1u << (lp & 31));
                        lp = -1;
                        lv = v2;
                        lvalid = false;
                    }
                }
                ++i;
            }
        }
    }
}

// Chunk layout: lane c owns the indices [cK, (c+1)K), K a multiple of 16 (a lane owns whole 16-bit units of the
// bitmap), and warms up, from the fresh state, over the `lead` indices in front of them (lane 0: indices in front of
// the read, on which the automaton does not step).
__device__ inline int chunk_len_uniform(int n) {
    const int k = (n + 1023) / 1024;
    return 16 * (k < 1 ? 1 : k);
}

// speculative pass + verification / re-run loop + replay of the hot long-detector runs.
// Returns 0 when the read's bitmap is in place, 1 when the fast pass cannot take it (alignment / room around the read),
// 2 when a lane met more hot runs than it can record (pathological signal: constant stretches, tiny variances).
template <int W1, typename T>
__device__ int detect_read_fast(const ReadCtx<T> &rc, EvHeader *hdr, FpLds<W1> *L) {
    const int n = (int)rc.n;
    if (n <= 0) return 0;
    // speculative warm-up before every chunk.  RNA events are ~5x longer, so the automata converge later: with 64
    // samples ~1.4 % of the chunk boundaries need a re-run, with 256 about 0.002 %.  A re-run costs the wave one
    // more pass over a chunk (K samples), the warm-up costs `lead` samples per lane: short reads (small K) are
    // better off with a short warm-up and the occasional re-run, long reads with a long one.
    int lead = n < 32768 ? SGK_LEAD_DNA_SHORT : SGK_LEAD_DNA;
    if (W1 == 7) lead = n <= 32768 ? SGK_LEAD_RNA_SHORT : SGK_LEAD_RNA;
    // the fast pass uses unguarded 4-byte-aligned 16-byte vector loads: it needs 16 readable samples behind the
    // read; other reads take the exact fallback
    if ((reinterpret_cast<uintptr_t>(rc.base) & 3u) != 0 || rc.hi < (int64_t)n + 16) return 1;
    const int K = chunk_len_uniform(n);
    const int c = lane_id();
    const int s = c * K;
    const int e0 = s + K;
    const int e = e0 < n ? e0 : n;
    const bool active = s < n;
    {
        LzSnapState z;
        z.sp = -1; z.sv = FLT_MAX; z.lm = LZ_NONE; z.r0 = 0; z.bits = 0u;
        L->init[c] = z;
        L->at_e[c] = z;
        L->nrec[c] = 0;
    }
    bool run = active;
    bool first = true;
    for (int iter = 0; iter < 66; ++iter) {
        pass_fast<W1, T>(rc, first, first ? lead : 0, first ? lead + K : K, run, s, e, L);
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
        if (c == 0) atomicAdd(&hdr->n_rerun, (uint32_t)__popcll(badmask));
        __syncthreads();
    }
    if (__any(active && L->nrec[c] > FP_NREC)) return 2;
    const unsigned long long hotm = __ballot(active && L->nrec[c] > 0);
    if (hotm != 0ull) {
        if (c == 0) atomicAdd(&hdr->n_hot_runs, (uint32_t)__popcll(hotm));
        __threadfence_block();
        __syncthreads();  // every lane's bitmap units are in memory before the replay ORs into them
        fp_replay_long<W1, T>(rc, L, active);
    }
    return 0;
}
