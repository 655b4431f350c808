// seqsum.h -- EXACT wave-parallel evaluation of the reference's sequential float32 sums.
//
// meanf / meani16 / stdvf / stdvi16 (src/stat.h:17-54) and the thresholds of jnn_core / jnnv2 (src/jnn.c:106-107,
// 195-199) accumulate into ONE float, strictly in sample order: `sum += x[i]`, one rounding per sample.  At 100 000
// samples the result is up to 6e-5 away from the exact sum, so the kernels must reproduce the rounding
// SEQUENCE -- but they need not run it serially:
//
//   while the running sum s stays inside one binade [2^E, 2^(E+1)] its unit in the last place is u = 2^(E-23), and
//   fl(s + x) = (S + rne(x/u, parity of S)) * u with S = s/u: the increment depends on s only through the PARITY of
//   its significand (round-to-nearest-even consults it on ties, nothing else).
//
// A lane owns SS_SPL consecutive terms of a tile.  It adds them, with the hardware's own float additions, to two
// SURROGATE starts inside the same binade, 1.5*2^E (S even) and 1.5*2^E + u (S odd); the differences of the float bit
// patterns are the lane's increments f0 / f1 for either incoming parity, ties, and whatever else IEEE does, included.
// Per lane the parity map p -> (p + f_p) & 1 is constant, the identity or the negation; the 64 maps are composed by a
// segmented xor scan on two 64-bit scalar masks, each lane picks its increment, a DPP scan adds them up, and
// S + total <= 2^24 certifies that the true sum never left the binade (all terms of the fast path are non-negative).
// If it did, the first crossing lane runs its terms natively from its true start (S_l * u, exact) and the lanes behind
// it repeat with the new binade: about log2(n / SS_HEAD) repeats per read and sum.  Tiles the argument does not cover (a
// negative term, a term comparable to the sum, a zero / huge / tiny / non-finite sum) are added one term at a time,
// natively, as are the first SS_HEAD terms of a read (the sum is still of the order of a term there).
//
// tools/proto/seqsum_proto.py is the numpy model of this file (same algorithm, checked against the plain loop by
// tests/test_seqsum_model.py); tests/test_gpu_stat.py compares the kernels built on it with the oracle and with the
// lane-per-read kernels, tests/test_gpu_shims.py runs the chains on hostile float arrays (sgk_meanf / sgk_stdvf).
#pragma once
#include "sgk_common.h"

namespace sgk {

constexpr int SS_SPL = 16;            // consecutive terms per lane per tile
constexpr int SS_TILE = 64 * SS_SPL;  // terms per tile
#ifndef SGK_SS_HEAD
#define SGK_SS_HEAD 256
#endif
constexpr int SS_HEAD = SGK_SS_HEAD;          // leading terms of a read that are added natively: the surrogates of a lane have
                                      // room for SS_SPL terms only once the sum is well beyond 4 * SS_SPL terms, and
                                      // every one of the first terms would be a binade crossing

__device__ __forceinline__ uint32_t ss_bits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float ss_float(uint32_t b) { return __uint_as_float(b); }
__device__ __forceinline__ float ss_uniform(float v) {
    return ss_float((uint32_t)__builtin_amdgcn_readfirstlane((int)ss_bits(v)));
}
__device__ __forceinline__ float ss_readlane(float v, int lane /* wave-uniform */) {
    return ss_float((uint32_t)__builtin_amdgcn_readlane((int)ss_bits(v), lane));
}

// an opaque zero: arithmetic that depends on it cannot be moved in front of the point where it is made
__device__ __forceinline__ uint32_t ss_opaque_zero() {
    uint32_t z = 0u;
    asm volatile("" : "+v"(z));
    return z;
}

// per-wave counters of what the chain had to do (debug / bench builds)
struct SsCount {
    uint32_t tiles, generic, walks, crossings, composes, serial;
};

// A chain's terms come from a functor: `template <int E> float get() const` is term lane * SS_SPL + E of the tile
// (oriented like the accumulator, 0 where the tile has no sample); `with(z)` is the same functor reading its samples
// through the opaque zero z.  Terms are recomputed from the packed samples where
// they are needed rather than kept in registers.

// m = fl(m + term) over ALL terms of the lanes l0 .. l1 of the tile, in order (terms the functor masks are 0, and
// fl(m + 0) = m: the accumulator is never -0).  One readlane and one dependent addition per term.
template <int E, typename TF>
__device__ __forceinline__ void ss_serial_lane(float &m, const TF &tf, int l) {
    if constexpr (E < SS_SPL) {
        m = m + ss_readlane(tf.template get<E>(), l);
        ss_serial_lane<E + 1>(m, tf, l);
    }
}
template <typename TF>
__device__ inline float ss_serial(float m, const TF &tf, int l0, int l1) {
    for (int l = l0; l <= l1; ++l) ss_serial_lane<0>(m, tf, l);
    return m;
}
// two chains over the same lanes at once (their additions interleave: twice the work in the same time)
template <int E, typename TA, typename TB>
__device__ __forceinline__ void ss_serial_lane2(float &ma, float &mb, const TA &ta, const TB &tb, int l) {
    if constexpr (E < SS_SPL) {
        ma = ma + ss_readlane(ta.template get<E>(), l);
        mb = mb + ss_readlane(tb.template get<E>(), l);
        ss_serial_lane2<E + 1>(ma, mb, ta, tb, l);
    }
}
template <typename TA, typename TB>
__device__ inline void ss_serial2(float &ma, float &mb, const TA &ta, const TB &tb, int l0, int l1) {
    for (int l = l0; l <= l1; ++l) ss_serial_lane2<0>(ma, mb, ta, tb, l);
}

// the lane's terms added to the two surrogate starts of the accumulator's binade; `neg` collects the terms' sign bits
struct SsWalk {
    float a0, a1;
    uint32_t neg;
};
template <int E, bool NEG, typename TF>
__device__ __forceinline__ void ss_walk_terms(SsWalk &w, const TF &tf) {
    if constexpr (E < SS_SPL) {
        const float x = tf.template get<E>();
        w.a0 = w.a0 + x;
        w.a1 = w.a1 + x;
        if (NEG) w.neg |= ss_bits(x);
        ss_walk_terms<E + 1, NEG>(w, tf);
    }
}
template <bool NEG, typename TF>
__device__ __forceinline__ SsWalk ss_walk(float m, const TF &tf) {
    const uint32_t b0 = (ss_bits(m) & 0x7f800000u) | 0x400000u;
    SsWalk w = {ss_float(b0), ss_float(b0 + 1u), 0u};
    ss_walk_terms<0, NEG>(w, tf);
    return w;
}
template <int E, typename TF>
__device__ __forceinline__ void ss_native_terms(float &v, const TF &tf) {
    if constexpr (E < SS_SPL) {
        v = v + tf.template get<E>();
        ss_native_terms<E + 1>(v, tf);
    }
}
template <int E, typename TF>
__device__ __forceinline__ void ss_scan_terms(bool &nz, bool &kill, float minf, const TF &tf) {
    if constexpr (E < SS_SPL) {
        const float x = tf.template get<E>();
        nz |= x != 0.0f;
        kill |= (x != x) || (x == minf);
        ss_scan_terms<E + 1>(nz, kill, minf, tf);
    }
}

// the lanes' parity maps p -> (p + f_p) & 1 composed in lane order (a segmented xor scan on two wave masks): returns
// the mask of the lanes whose run starts from an odd S, given the parity of S in front of lane 0
__device__ __forceinline__ unsigned long long ss_parity_in(int f0, int f1, int S) {
    const unsigned long long O0 = __ballot((f0 & 1) != 0), O1 = __ballot((f1 & 1) == 0);
    unsigned long long F = ~(O0 ^ O1), V = O0;  // F: constant map (value V); else xor by V
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        V ^= (V << d) & ~F;
        F |= F << d;
    }
    const unsigned long long p0 = (S & 1) ? ~0ull : 0ull;
    const unsigned long long out = V ^ (~F & p0);  // parity after each lane
    return (out << 1) | (p0 & 1ull);               // parity before each lane
}

// The common case of a tile, tried first: no negative term, every surrogate stayed in its binade, and so did the sum
// (~20 vector instructions, + the parity scan when some lane met a tie): returns 0, m advanced.  If the sum leaves its
// binade inside the tile, the crossing is taken here as well, with the increments and the scan already at hand: the
// first lane whose end value lies beyond the binade's top adds its terms natively from its true start (S_l * u, exact),
// m becomes the accumulator behind that lane, `skip` the number of lanes that are done, and ss_finish repeats the lanes
// behind with the new binade (return 2; 0 if that lane was the last).  Anything else -> return 1: ss_finish from scratch.
template <bool NEG, typename TF>
__device__ __forceinline__ int ss_fast(float &m, const SsWalk &w, const TF &tf, int &skip) {
    skip = 0;
    const uint32_t mb = ss_bits(ss_uniform(m));
    const uint32_t ex = (mb >> 23) & 0xffu;
    if ((mb >> 31) || ex < 27u || ex > 227u) return 1;
    const uint32_t b0 = (ex << 23) | 0x400000u, b1 = b0 + 1u;
    const uint32_t c0 = ss_bits(w.a0), c1 = ss_bits(w.a1);
    const int f0 = (int)(c0 - b0), f1 = (int)(c1 - b1);
    uint32_t bad = ((c0 ^ b0) | (c1 ^ b1)) >> 23;
    if (NEG) bad |= w.neg >> 31;
    if (__any(bad != 0u)) return 1;
    const int S = (int)((mb & 0x7fffffu) | 0x800000u);
    int f = f0;
    if (__any(f0 != f1)) f = __builtin_amdgcn_inverse_ballot_w64(ss_parity_in(f0, f1, S)) ? f1 : f0;
    const int incl = wave_incl_scan_i(f);
    const int tot = wave_last_i(incl);
    const float u = ss_float((ex - 23u) << 23);
    if (S + tot <= (1 << 24)) {
        m = (float)(S + tot) * u;
        return 0;
    }
    const int Sl = S + incl - f;
    const unsigned long long cm = __ballot(Sl + f > (1 << 24));
    const int ls = (int)__builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
    float v = (float)Sl * u;
    ss_native_terms<0>(v, tf.with(ss_opaque_zero()));
    m = ss_readlane(v, ls);
    skip = ls + 1;
    return skip >= 64 ? 0 : 2;
}

// One tile of a chain.  m: the accumulator (wave-uniform; the caller keeps it non-negative by orienting the terms);
// w: ss_walk(m, ...) of the same tile (so that the walks of several chains can be issued together, ahead of the
// branching below); NEG: the terms can be negative.  Returns the accumulator after the tile's 64 * SS_SPL terms.
template <bool NEG, typename TF>
__device__ inline float ss_finish(float m, const TF &tf0, SsWalk w, int skip = 0 /* lanes already done (ss_fast) */,
                                  SsCount *cnt = nullptr) {
    const int lane = lane_id();
    if (cnt) ++cnt->generic;
    for (;;) {
        m = ss_uniform(m);
        const uint32_t mb = ss_bits(m);
        const uint32_t ex = (mb >> 23) & 0xffu;
        if (ex == 0xffu || ex == 0u) {
            const TF tf = tf0.with(ss_opaque_zero());
            bool nz = false, kill = false;
            ss_scan_terms<0>(nz, kill, -m, tf);
            if (ex == 0xffu) {
                if (mb & 0x7fffffu) return m;  // NaN stays NaN
                // +-inf: stays unless a NaN or the opposite infinity arrives
                return __any(kill && lane >= skip) ? ss_float(0x7fc00000u) : m;
            }
            if (m == 0.0f && !__any(nz && lane >= skip)) return m;  // fl(0 + 0) = 0
            break;
        }
        if ((mb >> 31) || ex < 27u || ex > 227u) break;
        const int S = (int)((mb & 0x7fffffu) | 0x800000u);
        const uint32_t b0 = (ex << 23) | 0x400000u, b1 = b0 + 1u;
        if (skip) w = ss_walk<NEG>(m, tf0.with(ss_opaque_zero()));
        if (cnt) ++cnt->walks;
        const bool live = lane >= skip;
        const uint32_t c0 = ss_bits(w.a0), c1 = ss_bits(w.a1);
        const bool inr = ((c0 ^ b0) | (c1 ^ b1)) < 0x800000u;  // both surrogates ended in their binade (NaN: no)
        if (__any(live && (!inr || (NEG && (w.neg >> 31))))) break;
        const int f0 = live ? (int)(c0 - b0) : 0, f1 = live ? (int)(c1 - b1) : 0;
        int f = f0;
        if (__any(f0 != f1)) {  // some lane met a tie: the parity of S matters
            if (cnt) ++cnt->composes;
            f = __builtin_amdgcn_inverse_ballot_w64(ss_parity_in(f0, f1, S)) ? f1 : f0;
        }
        const int incl = wave_incl_scan_i(f);
        const int tot = wave_last_i(incl);
        const float u = ss_float((ex - 23u) << 23);
        if (S + tot <= (1 << 24)) return (float)(S + tot) * u;
        // the sum leaves the binade inside this tile: the first lane whose end value is beyond the binade's top runs
        // its terms natively from its true start; the lanes behind it repeat with the new binade
        if (cnt) ++cnt->crossings;
        const int Sl = S + incl - f;
        const unsigned long long cm = __ballot(Sl + f > (1 << 24));
        const int ls = (int)__builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
        float v = (float)Sl * u;
        ss_native_terms<0>(v, tf0.with(ss_opaque_zero()));
        m = ss_readlane(v, ls);
        skip = ls + 1;
        if (skip >= 64) return m;
    }
    if (cnt) ++cnt->serial;
    return ss_serial(m, tf0.with(ss_opaque_zero()), skip, 63);
}

}  // namespace sgk
