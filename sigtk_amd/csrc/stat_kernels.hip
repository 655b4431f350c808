// stat_kernels.hip -- per-read statistics, the JNN segmenter and the adaptor/polyA finder.
//
// Reference semantics that shape these kernels:
//   * meanf/meani16/stdvf/stdvi16 (src/stat.h:17-54) accumulate into ONE float, strictly in sample order; at 100k
//     samples the result differs from the exact value by up to ~6e-5 relative, so the rounding sequence must be
//     reproduced.  Two implementations live here:
//       - round 2 (default): one WAVE per read.  seqsum.h evaluates the sequential sum exactly, 1024 terms at a time
//         (surrogate starts in the sum's binade, parity maps, binade crossings repaired natively): k_stat_wave,
//         k_adaptor_wave, k_jnn_wave, k_polya_wave, with the reads dispatched longest first (launch_order);
//       - round 1 (SGK_LANE_PER_READ=1, and find_polya on large uniform batches): one read per LANE, serial float
//         chain, the 64 reads of a wave streamed through the LDS row stager (row_stream.h): k_moments, k_jnn, k_adaptor,
//         k_polya.  Kept as the independent second implementation the tests compare against, bit for bit.
//   * medians are order statistics (rank n/2, src/stat.h:56-73 + ksort.h:233-259): any exact selection works -> a
//     histogram with one bin per raw value over a window centred on the read's mean (k_stat_wave: 2048 bins per wave,
//     fused into its second pass; k_median: 8192 bins per 256-thread workgroup), a two-level radix select for regions
//     and for reads whose order statistic falls outside the window; pA median = pA(raw order statistic) because the
//     int16 -> pA map is monotone (non-increasing when range/digitisation < 0).
//   * jnn_core (src/jnn.c:190-278) and jnnv2 (src/jnn.c:99-179) are serial automata with thresholds derived from those
//     sequential float moments.  Their per-sample work is integer: in / out-of-range flags of the raw samples, integer
//     rolling totals with the exact constant division (tstat_math.h) and integer thresholds for jnnv2's run finder.
//     Wave-per-read forms: jnn_core in 64 chunks between data-determined sync points, from event to event on 32-bit
//     masks (jnn_chunks); jnnv2's run finder from threshold flip to flip (k_adaptor_wave).
#include <stdlib.h>

#include "row_stream.h"
#include "seqsum.h"
#include "sgk_common.h"
#include "side_stream.h"
#include "stat_args.h"
#include "tstat_math.h"

namespace sgk {

using Stream1 = RowPrefetch;

__device__ inline int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// region of read r a kernel works on (absolute sample index + length)
struct Region {
    int64_t start;
    int64_t len;
};
__device__ inline Region get_region(int mode, const sgk_batch_t &b, const sgk_prefix_rec_t *prec, uint32_t r) {
    Region g;
    g.start = (int64_t)b.offsets[r];
    g.len = (int64_t)b.lengths[r];
    if (mode == REG_ADAPT) {
        const sgk_prefix_rec_t p = prec[r];
        if (p.adapt_y > 0) { g.start += p.adapt_x; g.len = (int64_t)p.adapt_y - p.adapt_x; }
        else g.len = 0;
    } else if (mode == REG_POLYA) {
        const sgk_prefix_rec_t p = prec[r];
        if (p.adapt_y > 0 && p.polya_y > 0) { g.start += (int64_t)p.polya_x + p.adapt_y; g.len = (int64_t)p.polya_y - p.polya_x; }
        else g.len = 0;
    } else if (mode == REG_TAIL) {  // pA[adapt_y .. n), find_polya's input (cfunc.c:186-191)
        const sgk_prefix_rec_t p = prec[r];
        if (p.adapt_y > 0) { g.start += p.adapt_y; g.len -= p.adapt_y; }
        else g.len = 0;
    }
    if (g.len < 0) g.len = 0;
    return g;
}

// what the workgroups of a long read exchange is written and read with agent-scope atomics
__device__ __forceinline__ uint32_t lc_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long lc_ld(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lc_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lc_st(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The record of read r if k_long_list listed it (wave-uniform; null: an ordinary read).  k_long_chains may run BESIDE the
// kernel that asks (stat, jnn: side stream), so who does a read is decided by what k_long_list wrote -- LongSums::rec_off,
// final before either kernel starts -- and never by LongSums::valid, which k_long_chains sets when it is done (prefix'
// k_adaptor_wave, launched behind it, reads the sums under valid).
__device__ inline const LongSums *find_long(const StatArgs &a, uint32_t r, int64_t len) {
    if (!a.longs || len < (int64_t)a.long_min) return nullptr;
    const uint32_t nl = a.long_hdr->n_long, n = nl < LC_CAP ? nl : LC_CAP;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane_id();
        const unsigned long long hit = __ballot(i < n && a.long_list[i] == r);
        if (hit) {
            return a.longs + i0 + (uint32_t)(__ffsll((long long)hit) - 1);
        }
    }
    return nullptr;
}

// The redo launch of a wave kernel (StatArgs::long_redo): wave widx looks at entry widx of the long list and takes its
// read iff k_long_chains declined it (a barrier of its workgroups timed out, lc_barrier).  Usually none: every wave
// returns at once.
__device__ inline bool long_redo_read(const StatArgs &a, uint32_t widx, uint32_t &r) {
    if (!a.longs) return false;
    const uint32_t nl = a.long_hdr->n_long, n = nl < LC_CAP ? nl : LC_CAP;
    if (widx >= n) return false;
    if (a.longs[widx].rec_off == LC_NO_REC || a.long_work[widx].failed == 0u) return false;
    r = a.long_list[widx];
    if (lane_id() == 0) atomicAdd(&a.long_hdr->n_declined, 1u);
    return true;
}

__device__ inline float clampf_raw(int16_t v) {  // rm_outlier, src/jnn.c:61-77
    return v > 1200 ? 1200.0f : (v < 0 ? 0.0f : (float)v);
}
__device__ inline int clampi_raw(int16_t v) {  // rm_outlier as an integer (the float it yields is that integer)
    return v > 1200 ? 1200 : (v < 0 ? 0 : (int)v);
}
__device__ inline float clampf_pa(float v) {     // rm_outlierf, src/jnn.c:79-95
    return v > 1200.0f ? 1200.0f : (v < 0.0f ? 0.0f : v);
}

// Lane-per-row sequential sweep: calls f(j, raw) for j = 0..len-1 in order.  All lanes of the wave
// must call it (cooperative tile loads); lanes with len == 0 just help loading.  Tile t+1 is in
// flight while tile t is consumed out of registers.
template <int K, typename F>
__device__ __forceinline__ void sweep_tile_elems(const uint32_t (&w)[32], int64_t j0, int64_t len, F &f) {
    if constexpr (K < TILE) {
        const int64_t j = j0 + K;
        if (j >= 0 && j < len) f(j, RowPrefetch::sample<K>(w));
        sweep_tile_elems<K + 1>(w, j0, len, f);
    }
}
template <int K, typename F>
__device__ __forceinline__ void sweep_tile_full(const uint32_t (&w)[32], int j0, F &f) {
    if constexpr (K < TILE) {
        f((int64_t)(j0 + K), RowPrefetch::sample<K>(w));
        sweep_tile_full<K + 1>(w, j0, f);
    }
}
template <typename F>
__device__ inline void sweep_rows(RowPrefetch &rs, int skip, int64_t len, F f) {
    const int maxq = wave_max_i((int)(len > 0 ? skip + len : 0));
    const int ntiles = (maxq + TILE - 1) / TILE;
    if (ntiles == 0) return;
    rs.issue(0);
    rs.commit(0);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) rs.issue(t + 1);
        uint32_t w[32];
        rs.row(w);
        const int64_t j0 = (int64_t)t * TILE - skip;
        // common case: the whole tile lies inside the row of every lane that still has samples (lanes whose row
        // has ended, or has not begun, sit the tile out) -> no per-sample predicates
        const bool inside = j0 >= 0 && j0 + TILE <= len, outside = j0 + TILE <= 0 || j0 >= len;
        if (__all(inside || outside)) {
            if (inside) sweep_tile_full<0>(w, (int)j0, f);
        } else if (!outside) sweep_tile_elems<0>(w, j0, len, f);
        if (t + 1 < ntiles) rs.commit(t + 1);
    }
}

// Same contract as sweep_rows, but the tile is consumed in rolled parts of P samples (P/2 registers, a P-step
// unrolled body): for callbacks with a lot of code or state (the jnn automaton) this keeps the kernel small and
// out of scratch.
template <int K, int P, typename F>
__device__ __forceinline__ void sweep_part_elems(const uint32_t (&w)[P / 2], int64_t j0, int64_t len, F &f) {
    if constexpr (K < P) {
        const int64_t j = j0 + K;
        if (j >= 0 && j < len) f(j, RowPrefetch::sample_part<K>(w));
        sweep_part_elems<K + 1, P>(w, j0, len, f);
    }
}
template <int K, int P, typename F>
__device__ __forceinline__ void sweep_part_full(const uint32_t (&w)[P / 2], int j0, F &f) {
    if constexpr (K < P) {
        f((int64_t)(j0 + K), RowPrefetch::sample_part<K>(w));
        sweep_part_full<K + 1, P>(w, j0, f);
    }
}
template <int P, typename F>
__device__ inline void sweep_rows_parts(RowPrefetch &rs, int skip, int64_t len, F f) {
    const int maxq = wave_max_i((int)(len > 0 ? skip + len : 0));
    const int ntiles = (maxq + TILE - 1) / TILE;
    if (ntiles == 0) return;
    rs.issue(0);
    rs.commit(0);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) rs.issue(t + 1);
#pragma unroll 1
        for (int h = 0; h < TILE / P; ++h) {
            uint32_t w[P / 2];
            rs.row_part<P>(h, w);
            const int64_t j0 = (int64_t)t * TILE + h * P - skip;
            const bool inside = j0 >= 0 && j0 + P <= len, outside = j0 + P <= 0 || j0 >= len;
            if (__all(inside || outside)) {  // see sweep_rows
                if (inside) sweep_part_full<0, P>(w, (int)j0, f);
            } else if (!outside) sweep_part_elems<0, P>(w, j0, len, f);
        }
        if (t + 1 < ntiles) rs.commit(t + 1);
    }
}

__device__ inline RowPrefetch make_stream(char *lds, const sgk_batch_t &b, int64_t start, bool wanted, int &skip) {
    RowPrefetch rs;
    const int64_t rb = start & ~(int64_t)7;
    skip = (int)(start - rb);
    rs.init(lds, b.samples, (int64_t)b.n_samples, rb, __ballot(wanted));
    return rs;
}

// (sgk_stat_rec_t::reserved / sgk_prefix_rec_t::reserved between kernels: reads whose median is still to be found)
constexpr uint32_t FLAG_MEDIAN_WHOLE = 1u, FLAG_MEDIAN_ADAPT = 1u, FLAG_MEDIAN_POLYA = 2u;

// ---------------------------------------------------------------- moments (src/stat.h:17-54)
// HIST (stat without the pA output): the deviation pass also counts, per lane, the samples in a window of MH_BINS raw
// values around the read's mean (a column of LDS words per lane: no other lane touches it) and those below it, and the
// read's median (rank n/2 and, for a negative unit, n-1-n/2: src/stat.h:56-73) is read off that: the third pass over
// the samples (k_median) is only needed for reads whose median lies outside the window (flagged for k_median; the sp1
// fixture's reads have their median -10 .. +21 raw values from their mean, 13 at the 99th percentile: a few per cent of
// real reads, whose k_median workgroups cost in proportion).  The lane kernels are bound by HBM and close to bound by
// instruction issue: 125 000 x 100 000 samples 13.0 (k_moments 9.0 + k_median 4.0) -> 10.0 ms with 32 bins (10.8 with a
// branch around the counting instead of the extra row); 64 bins
// (16 KB of LDS per wave) lose the occupancy the kernel streams with (16.5 ms), 64 16-bit counters packed two to a word
// cost more instructions than they save (12.6 ms).
constexpr int MH_BINS = 32;
#ifndef SGK_MOM_WAVES
#define SGK_MOM_WAVES 1   // waves per SIMD the register allocation aims at (3: 168 registers + 42 spilled, 13.0 against 10.1 ms)
#endif
template <int MODE, bool HIST = false>
__global__ __launch_bounds__(64, SGK_MOM_WAVES) void k_moments(StatArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[Stream1::LDS_BYTES];
    __shared__ uint32_t mh[HIST ? (MH_BINS + 1) * 64 : 1];  // (+ a row nobody reads: samples outside the window)
    const int lane = lane_id();
    const uint32_t r = blockIdx.x * 64 + lane;
    const bool valid = r < a.b.n_reads;
    Region g = {0, 0};
    Scale sc = {0.0f, 1.0f};
    if (valid) {
        g = get_region(MODE, a.b, a.prefix, r);
        sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
    }
    int skip;
    Stream1 rs = make_stream(lds, a.b, g.start, valid && g.len > 0, skip);
    const float nf = (float)(int)g.len;
    float sraw = 0.0f, spa = 0.0f;
    sweep_rows(rs, skip, g.len, [&](int64_t, int16_t v) {
        sraw = sraw + (float)v;
        spa = spa + to_pa(v, sc);
    });
    const float mraw = sraw / nf, mpa = spa / nf;
    float qraw = 0.0f, qpa = 0.0f;
    int lo = 0;
    uint32_t below = 0u;
    if (HIST) {
        const int c = (mraw == mraw) ? (int)fminf(fmaxf(mraw, -32768.0f), 32767.0f) : 0;
        lo = c - MH_BINS / 2;
#pragma unroll
        for (int b = 0; b < MH_BINS; ++b) mh[b * 64 + lane] = 0u;
    }
    sweep_rows(rs, skip, g.len, [&](int64_t, int16_t v) {
        const float d = (float)v - mraw;
        qraw = qraw + d * d;
        const float e = to_pa(v, sc) - mpa;
        qpa = qpa + e * e;
        if (HIST) {
            // (no branch: a sample outside the window goes to the extra row; this lane's column)
            const uint32_t b = (uint32_t)((int)v - lo);
            atomicAdd(&mh[(b < (uint32_t)MH_BINS ? b : (uint32_t)MH_BINS) * 64u + (uint32_t)lane], 1u);
            below += b >> 31;
        }
    });
    if (!valid) return;
    const float sdraw = sqrtf(qraw / nf), sdpa = sqrtf(qpa / nf);
    // HIST: the order statistics of ranks n/2 and (negative unit) n-1-n/2 from the window's counts
    int b1 = -1, b2 = -1;
    if (HIST && g.len > 0) {
        const uint32_t k = (uint32_t)(g.len / 2);
        const bool mirrored = sc.unit < 0.0f && g.len - 1 - (int64_t)k != (int64_t)k;  // pA order is the reverse of the raw order
        const uint32_t k2 = mirrored ? (uint32_t)(g.len - 1 - (int64_t)k) : k;
        uint32_t acc = below;
        for (int b = 0; b < MH_BINS; ++b) {
            const uint32_t h = mh[b * 64 + lane];
            if (b1 < 0 && k >= acc && k < acc + h) b1 = b;
            if (b2 < 0 && k2 >= acc && k2 < acc + h) b2 = b;
            acc += h;
        }
    }
    const bool have_median = b1 >= 0 && b2 >= 0;
    if (MODE == REG_WHOLE) {
        sgk_stat_rec_t *o = a.stat + r;
        o->raw_mean = mraw; o->pa_mean = mpa; o->raw_std = sdraw; o->pa_std = sdpa;
        o->n = (uint32_t)g.len;
        o->reserved = 0;
        if (HIST) {
            if (g.len <= 0) { o->raw_median = 0; o->pa_median = 0.0f; }
            else if (have_median) {
                o->raw_median = lo + b1;
                o->pa_median = to_pa((int16_t)(lo + b2), sc);
            } else o->reserved = FLAG_MEDIAN_WHOLE;  // outside the window: k_median (FLAGGED) takes the read
        }
    } else if (MODE == REG_ADAPT) {
        a.prefix[r].adapt_mean = mpa;
        a.prefix[r].adapt_std = sdpa;
        if (HIST && g.len > 0) {
            if (have_median) a.prefix[r].adapt_median = to_pa((int16_t)(lo + b2), sc);
            else a.prefix[r].reserved |= FLAG_MEDIAN_ADAPT;
        }
    } else {
        a.prefix[r].polya_mean = mpa;
        a.prefix[r].polya_std = sdpa;
        if (HIST && g.len > 0) {
            if (have_median) a.prefix[r].polya_median = to_pa((int16_t)(lo + b2), sc);
            else a.prefix[r].reserved |= FLAG_MEDIAN_POLYA;
        }
    }
}

// ---------------------------------------------------------------- median (src/stat.h:56-73)
// Visit every key of x[0..n) with a 256-thread workgroup: 16-byte vector loads over the aligned
// middle (four in flight per thread), scalar loads for the unaligned head and tail.  With PA, pa[i] =
// signal_in_picoamps(x[i]) is written on the way (two 16-byte stores per vector): the fused stat + pa of
// BASELINE config 4 costs no extra pass over the samples.
template <bool PA, typename F>
__device__ __forceinline__ void visit_keys(const int16_t *x, int64_t n, F f, float *pa = nullptr,
                                           Scale sc = Scale{0.0f, 1.0f}) {
    const int t = threadIdx.x;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int64_t head = (int64_t)(((16 - (addr & 15)) & 15) / 2);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / 8;
    const int64_t tail0 = head + nvec * 8;
    if (t < head) {
        f((uint32_t)((int)x[t] + 32768));
        if (PA) pa[t] = to_pa(x[t], sc);
    }
    if (tail0 + t < n && t < 8) {
        f((uint32_t)((int)x[tail0 + t] + 32768));
        if (PA) pa[tail0 + t] = to_pa(x[tail0 + t], sc);
    }
    const uint4 *v = reinterpret_cast<const uint4 *>(x + head);
    for (int64_t i = t; i < nvec; i += 256 * 4) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = i + (int64_t)u * 256;
            q[u] = (k < nvec) ? v[k] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = i + (int64_t)u * 256;
            if (k < nvec) {
                const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                float o[8];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int16_t s0 = (int16_t)(w[d] & 0xffffu), s1 = (int16_t)(w[d] >> 16);
                    f((uint32_t)((int)s0 + 32768));
                    f((uint32_t)((int)s1 + 32768));
                    if (PA) { o[2 * d] = to_pa(s0, sc); o[2 * d + 1] = to_pa(s1, sc); }
                }
                if (PA) {
                    float4 *dst = reinterpret_cast<float4 *>(pa + head + k * 8);  // x + head is 16-byte aligned
                    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
                    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        }
    }
}

// exclusive prefix of one count per thread over a 256-thread workgroup (a DPP scan per wave + the four wave totals
// through part[260 .. 263]; a serial pass of one thread over 256 LDS words cost a short read more than counting its
// samples did)
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t s, uint32_t *part /*264*/) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t incl = (uint32_t)wave_incl_scan_i((int)s);
    if (lane == 63) part[260 + wv] = incl;
    __syncthreads();
    uint32_t before = incl - s;
#pragma unroll
    for (int w = 0; w < 3; ++w) before += w < wv ? part[260 + w] : 0u;
    __syncthreads();
    return before;
}

// rank-k order statistic of the int16 keys of a region, by a 256-thread workgroup
template <bool PA = false>
__device__ int block_select(const int16_t *x, int64_t n, int64_t rank, uint32_t *hist /*4096*/, uint32_t *part /*264*/,
                            float *pa = nullptr, Scale sc = Scale{0.0f, 1.0f}) {
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) hist[i] = 0;
    __syncthreads();
    visit_keys<PA>(x, n, [&](uint32_t key) { atomicAdd(&hist[key >> 4], 1u); }, pa, sc);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += hist[t * 16 + k];
    const uint32_t before = block_excl_scan_256(s, part);
    if ((uint64_t)rank >= before && (uint64_t)rank < (uint64_t)before + s) {
        uint32_t acc = before;
        for (int k = 0; k < 16; ++k) {
            const uint32_t h = hist[t * 16 + k];
            if ((uint64_t)rank < (uint64_t)acc + h) { part[256] = (uint32_t)(t * 16 + k); part[257] = (uint32_t)(rank - acc); break; }
            acc += h;
        }
    }
    __syncthreads();
    const uint32_t bin = part[256], rank2 = part[257];
    __syncthreads();
    if (t < 16) hist[t] = 0;
    __syncthreads();
    visit_keys<false>(x, n, [&](uint32_t key) {
        if ((key >> 4) == bin) atomicAdd(&hist[key & 15u], 1u);
    });
    __syncthreads();
    if (t == 0) {
        uint32_t acc = 0, val = 0;
        for (int k = 0; k < 16; ++k) {
            if (rank2 < acc + hist[k]) { val = (uint32_t)k; break; }
            acc += hist[k];
        }
        part[256] = val;
    }
    __syncthreads();
    const int res = (int)((bin << 4) | part[256]) - 32768;
    __syncthreads();
    return res;
}

// Order statistics of ranks k1 and k2 from ONE pass: an LDS histogram with one bin per raw value over the window
// [lo, lo + RANGE_BINS) (centred on the read's mean, which k_moments has already written); values outside are
// clipped into the two edge bins.  A rank that lands in an edge bin is not trusted (ok = false -> the caller
// falls back to the two-level select).  Nanopore raw signals span a few hundred ADC codes, so this is the path taken.
constexpr int RANGE_BINS = 8192;
template <bool PA, int nb /* bins: RANGE_BINS or a narrower window */>
__device__ bool block_select_range(const int16_t *x, int64_t n, int64_t k1, int64_t k2, int lo,
                                   uint32_t *hist /*RANGE_BINS*/, uint32_t *part /*264*/, int &r1, int &r2,
                                   float *pa, Scale sc) {
    const int t = threadIdx.x;
    constexpr int per = nb / 256;
    for (int i = t; i < nb; i += 256) hist[i] = 0;
    __syncthreads();
    const int base = lo + 32768;
    visit_keys<PA>(x, n, [&](uint32_t key) {
        int b = (int)key - base;
        b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
        atomicAdd(&hist[b], 1u);
    }, pa, sc);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < per; ++k) s += hist[t * per + k];
    const uint32_t before = block_excl_scan_256(s, part);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int64_t rank = w ? k2 : k1;
        if ((uint64_t)rank >= before && (uint64_t)rank < (uint64_t)before + s) {
            uint32_t acc = before;
            for (int k = 0; k < per; ++k) {
                const uint32_t h = hist[t * per + k];
                if ((uint64_t)rank < (uint64_t)acc + h) { part[256 + w] = (uint32_t)(t * per + k); break; }
                acc += h;
            }
        }
    }
    __syncthreads();
    const int b1 = (int)part[256], b2 = (int)part[257];
    __syncthreads();
    r1 = lo + b1;
    r2 = lo + b2;
    return b1 > 0 && b1 < nb - 1 && b2 > 0 && b2 < nb - 1;
}

// FLAGGED: only the reads k_stat_wave could not settle (order statistic outside its window)
template <int MODE, bool PA = false, bool FLAGGED = false>
__global__ __launch_bounds__(256) void k_median(StatArgs a) {
    __shared__ uint32_t hist[MODE == REG_WHOLE ? RANGE_BINS : 4096];
    __shared__ uint32_t part[264];
    const uint32_t r = blockIdx.x;
    if (FLAGGED) {
        const uint32_t fl = MODE == REG_WHOLE ? a.stat[r].reserved : a.prefix[r].reserved;
        if (!(fl & (MODE == REG_POLYA ? 2u : 1u))) return;
    }
    const Region g = get_region(MODE, a.b, a.prefix, r);
    if (g.len <= 0) {
        if (threadIdx.x == 0 && MODE == REG_WHOLE) { a.stat[r].raw_median = 0; a.stat[r].pa_median = 0.0f; a.stat[r].reserved = 0; }
        return;
    }
    const Scale sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
    const int16_t *x = a.b.samples + g.start;
    float *pa = PA ? a.pa_out + g.start : nullptr;
    const int64_t k = g.len / 2;
    const bool mirrored = sc.unit < 0.0f && g.len - 1 - k != k;  // pA order is the reverse of the raw order
    int med, med_for_pa;
    bool done = false;
    if (MODE == REG_WHOLE) {
        // window centred on the read's raw mean (written by k_moments, which runs before this kernel)
        // (a short read gets a narrower window: clearing and scanning 8192 bins costs a 5 000-sample read more than
        // counting its samples; +-1024 raw values around the mean still hold any nanopore read's median)
        const float m = a.stat[r].raw_mean;
        const int nb = g.len <= 65536 ? 2048 : RANGE_BINS;
        int c = (m == m) ? (int)fminf(fmaxf(m, -32768.0f), 32767.0f) : 0;
        int lo = c - nb / 2;
        lo = lo < -32768 ? -32768 : (lo > 32768 - nb ? 32768 - nb : lo);
        const int64_t k2 = mirrored ? g.len - 1 - k : k;
        done = nb == 2048 ? block_select_range<PA, 2048>(x, g.len, k, k2, lo, hist, part, med, med_for_pa, pa, sc)
                          : block_select_range<PA, RANGE_BINS>(x, g.len, k, k2, lo, hist, part, med, med_for_pa, pa, sc);
        pa = nullptr;  // already written
    }
    if (!done) {
        if (PA && pa) med = block_select<true>(x, g.len, k, hist, part, pa, sc);
        else med = block_select<false>(x, g.len, k, hist, part);
        med_for_pa = mirrored ? block_select<false>(x, g.len, g.len - 1 - k, hist, part) : med;
    }
    if (threadIdx.x == 0) {
        const float pm = to_pa((int16_t)med_for_pa, sc);
        if (MODE == REG_WHOLE) { a.stat[r].raw_median = med; a.stat[r].pa_median = pm; a.stat[r].reserved = 0; }
        else if (MODE == REG_ADAPT) a.prefix[r].adapt_median = pm;
        else a.prefix[r].polya_median = pm;
        if (FLAGGED && MODE != REG_WHOLE) a.prefix[r].reserved &= ~(MODE == REG_POLYA ? 2u : 1u);
    }
}

// two samples per packed 16-bit instruction: the outlier clamp of rm_outlier (src/jnn.c:61-77)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 clamp_raw2(uint32_t w) {
    s16x2 v = __builtin_bit_cast(s16x2, w);
    v = __builtin_elementwise_max(v, (s16x2){0, 0});
    return __builtin_elementwise_min(v, (s16x2){1200, 1200});
}

// ---------------------------------------------------------------- one WAVE per read (seqsum.h)
// The sequential float sums of a read (or of a region of it) by one wavefront: tiles of 64 x SS_SPL samples, every
// lane loads its 32 contiguous bytes with two 16-byte loads (tile t + 1 is in flight while tile t is consumed), the
// sums advance through the tile as seqsum.h describes.  Two passes over the samples:
//   1. sum of raw and of pA                                   -> the two means (src/stat.h:17-33)
//   2. sums of the squared deviations (src/stat.h:36-54), the histogram of the raw values over a window of WH_BINS codes
//      centred on the raw mean (median = order statistic of rank n/2, src/stat.h:56-73), and -- fused stat + pa,
//      BASELINE config 4 -- the pA value of every sample
// The second pass of a read follows its first on the same wave; reads whose order statistic falls outside the window
// are flagged (`reserved`) and taken by k_median.  Ragged batches cost what their samples cost: a wave is busy for the
// length of ITS read, not for the longest read among 64 neighbours as in the lane-per-read kernels above.
constexpr int WH_BINS = 2048;
static_assert(WH_BINS == (int)LC_HIST_BINS, "the long reads' histograms in the workspace");

struct WaveTile {
    uint32_t w[SS_SPL / 2];
    template <int E>
    __device__ __forceinline__ int16_t sample() const {
        return (E & 1) ? (int16_t)(w[E / 2] >> 16) : (int16_t)(w[E / 2] & 0xffffu);
    }
};
// This lane's 16 samples of the tile that starts at base-relative index `tile0` (wave-uniform, >= 0, a multiple of 8, as
// is n_total >= 8).  Both 16-byte loads are unconditional, so that the load of tile t + 1 stays in flight under tile t,
// and addressed as uniform base + 32-bit lane offset (scalar address arithmetic).  (Round 5: with the non-temporal hint
// on these loads every wave kernel is slower -- stat+pa 19.2 -> 20.5 ms, jnn 15.5 -> 16.7, jnnv2 13.1 -> 18.7: the
// second 16-byte load of a lane and jnnv2's trailing tile live on the line staying where the first load put it.)
// Near the end of the buffer the offsets are clamped to its last 16 bytes: such samples are outside the region and
// masked by the term functors.
__device__ __forceinline__ void wt_load(WaveTile &t, const int16_t *samples, int64_t n_total, int64_t tile0) {
    const int64_t last = n_total - 8;
    const int64_t t0 = tile0 < last ? tile0 : last;
    const int64_t room64 = (last - t0) * (int64_t)sizeof(int16_t);
    const uint32_t room = room64 > 4096 ? 4096u : (uint32_t)room64;  // largest legal byte offset, a multiple of 16
    const char *tb = reinterpret_cast<const char *>(samples + t0);
    const uint32_t lo = (uint32_t)lane_id() * (uint32_t)(SS_SPL * sizeof(int16_t));
    const uint32_t o0 = lo < room ? lo : room, o1 = lo + 16u < room ? lo + 16u : room;
    const uint4 q0 = *static_cast<const uint4 *>(__builtin_assume_aligned(tb + o0, 16));
    const uint4 q1 = *static_cast<const uint4 *>(__builtin_assume_aligned(tb + o1, 16));
    t.w[0] = q0.x; t.w[1] = q0.y; t.w[2] = q0.z; t.w[3] = q0.w;
    t.w[4] = q1.x; t.w[5] = q1.y; t.w[6] = q1.z; t.w[7] = q1.w;
}

// calls f.template operator()<E>(raw, valid) for this lane's 16 samples (tile-local indices q0 .. q0 + 15)
template <int E, bool INTERIOR, typename F>
__device__ __forceinline__ void wt_each_(const WaveTile &t, int q0, int q_lo, int q_hi, F &f) {
    if constexpr (E < SS_SPL) {
        f.template operator()<E>(t.sample<E>(), INTERIOR || (q0 + E >= q_lo && q0 + E < q_hi));
        wt_each_<E + 1, INTERIOR>(t, q0, q_lo, q_hi, f);
    }
}

struct WaveRead {  // wave-uniform description of the region a wave works on
    const int16_t *samples;
    int64_t n_total, rb, len;
    int skip, ntiles;
    __device__ void init(const sgk_batch_t &b, const Region &g) {
        samples = b.samples;
        n_total = (int64_t)b.n_samples;
        rb = g.start & ~(int64_t)7;
        skip = (int)(g.start - rb);
        len = g.len;
        ntiles = (int)((skip + len + SS_TILE - 1) / SS_TILE);
    }
    __device__ __forceinline__ void load(WaveTile &t, int tile) const {
        wt_load(t, samples, n_total, rb + (int64_t)tile * SS_TILE);
    }
    // a tile strictly inside the region (and not the first one, whose head is added natively): no predicates
    __device__ __forceinline__ bool interior(int tile) const {
        return tile > 0 && (int64_t)(tile + 1) * SS_TILE - skip <= len;
    }
    __device__ __forceinline__ int head() const { return (int)(len < SS_HEAD ? len : SS_HEAD); }
    // tile-local index range [q_lo, q_hi) of the region's samples in `tile`, without its first `drop` samples
    __device__ __forceinline__ void range(int tile, int drop, int &q_lo, int &q_hi) const {
        const int64_t lo = (int64_t)skip + drop - (int64_t)tile * SS_TILE, hi = (int64_t)skip + len - (int64_t)tile * SS_TILE;
        q_lo = lo < 0 ? 0 : (lo > SS_TILE ? SS_TILE : (int)lo);
        q_hi = hi < 0 ? 0 : (hi > SS_TILE ? SS_TILE : (int)hi);
    }
};

// term functors of the four sums (seqsum.h): INTERIOR tiles need no validity test.  `z` is 0; the rare paths of
// ss_finish pass an OPAQUE zero (SsOpaque) so that their term arithmetic stays inside those paths -- the compiler
// otherwise hoists all of it in front of the fast path and keeps 32 terms alive across it.
template <bool INTERIOR>
struct TermBase {
    static constexpr bool interior = INTERIOR;
    const WaveTile &t;
    int q0, q_lo, q_hi;  // q0: tile-local index of this lane's first sample
    uint32_t z;
    template <int E>
    __device__ __forceinline__ bool valid() const { return INTERIOR || (q0 + E >= q_lo && q0 + E < q_hi); }
    template <int E>
    __device__ __forceinline__ int16_t sample() const {
        const uint32_t w = t.w[E / 2] ^ z;
        return (E & 1) ? (int16_t)(w >> 16) : (int16_t)(w & 0xffffu);
    }
    template <int E>
    __device__ __forceinline__ float clamped() const {  // rm_outlier of the sample: the clamp is packed, two per dword
        const s16x2 c = clamp_raw2(t.w[E / 2] ^ z);
        return (float)((E & 1) ? c.y : c.x);
    }
};
template <bool INTERIOR>
struct TermRaw {  // (float)raw, src/stat.h:29-33
    TermBase<INTERIOR> b;
    int sm;  // -1 while the accumulator runs on the negated chain (orientation, as TermPa's unit), else 0
    __device__ __forceinline__ TermRaw with(uint32_t z) const { TermRaw r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        // negated as an integer: a zero sample stays +0 (a -0.0 term would count as negative and end the fast walk)
        const int v = (int)b.template sample<E>();
        return b.template valid<E>() ? (float)((v ^ sm) - sm) : 0.0f;
    }
};
template <bool INTERIOR>
struct TermPa {   // pA, src/stat.h:17-27 on signal_in_picoamps' output
    TermBase<INTERIOR> b;
    Scale so;     // unit carries the orientation of the accumulator
    __device__ __forceinline__ TermPa with(uint32_t z) const { TermPa r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        return b.template valid<E>() ? to_pa(b.template sample<E>(), so) : 0.0f;
    }
};
template <bool INTERIOR>
struct TermDevRaw {  // (raw - mean)^2, src/stat.h:46-54
    TermBase<INTERIOR> b;
    float mean;
    __device__ __forceinline__ TermDevRaw with(uint32_t z) const { TermDevRaw r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        const float d = (float)b.template sample<E>() - mean;
        return b.template valid<E>() ? d * d : 0.0f;
    }
};
template <bool INTERIOR>
struct TermDevPa {   // (pA - mean)^2, src/stat.h:36-44
    TermBase<INTERIOR> b;
    Scale sc;
    float mean;
    __device__ __forceinline__ TermDevPa with(uint32_t z) const { TermDevPa r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        const float d = to_pa(b.template sample<E>(), sc) - mean;
        return b.template valid<E>() ? d * d : 0.0f;
    }
};

#ifndef SGK_SS_EDGE_OPAQUE
#define SGK_SS_EDGE_OPAQUE 1
#endif
__device__ __forceinline__ int ss_edge_zero() { return SGK_SS_EDGE_OPAQUE ? (int)ss_opaque_zero() : 0; }
// the signed value of an oriented accumulator (a zero accumulator stands for +0)
__device__ __forceinline__ float ss_signed(float m, bool negated) { return m == 0.0f ? 0.0f : (negated ? -m : m); }

// one tile of two chains: both walks are issued before either chain's (branching) bookkeeping.  mka / mkb build the
// chains' term functors from a TermBase<INTERIOR>.
template <bool NEG, typename MA, typename MB>
__device__ __forceinline__ void ss_tile2(float &ma, float &mb, const WaveRead &wr, const WaveTile &cur, int t, MA mka, MB mkb,
                                         SsCount *ca = nullptr, SsCount *cb = nullptr) {
    const int q0 = lane_id() * SS_SPL;
    int q_lo, q_hi;
    if (t == 0) {  // the head of the read, natively (the functors mask what lies behind it)
        wr.range(0, 0, q_lo, q_hi);
        const int qh = q_lo + wr.head();
        if (qh > q_lo)
            ss_serial2(ma, mb, mka(TermBase<false>{cur, q0, q_lo, qh, 0u}), mkb(TermBase<false>{cur, q0, q_lo, qh, 0u}),
                       q_lo / SS_SPL, (qh - 1) / SS_SPL);
    }
    wr.range(t, t == 0 ? wr.head() : 0, q_lo, q_hi);
    SsWalk wa, wb;
    if (wr.interior(t)) {
        wa = ss_walk<NEG>(ma, mka(TermBase<true>{cur, q0, q_lo, q_hi, 0u}));
        wb = ss_walk<NEG>(mb, mkb(TermBase<true>{cur, q0, q_lo, q_hi, 0u}));
    } else {
        // (The 32 validity compares of an edge tile are evaluated in front of the branch, on every tile.  Round 5 kept
        // them inside it with an opaque copy of q0, as ss_tile1 does: two registers more, which k_stat_wave does not
        // have -- 127 and 2 spilled, 20.05 against 19.72 ms for stat+pa at 125 000 x 100 000; opaque copies of the
        // scalars q_lo / q_hi instead cost no register and are slower all the same, 19.7 against 19.4.)
        wa = ss_walk<NEG>(ma, mka(TermBase<false>{cur, q0, q_lo, q_hi, 0u}));
        wb = ss_walk<NEG>(mb, mkb(TermBase<false>{cur, q0, q_lo, q_hi, 0u}));
    }
    if (ca) { ++ca->tiles; ++cb->tiles; }
    int ska, skb;
    if (ss_fast<NEG>(ma, wa, mka(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), ska))
        ma = ss_finish<NEG>(ma, mka(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), wa, ska, ca);
    if (ss_fast<NEG>(mb, wb, mkb(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), skb))
        mb = ss_finish<NEG>(mb, mkb(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), wb, skb, cb);
}

// pA of every sample of tile t, written as whole cache lines: the tile is read once more as 4 x 256 samples with 8 bytes
// per lane (L2 hits) so that a store instruction covers 1 KB contiguously (the sums' layout, 64 bytes per lane, would make
// every store instruction touch 32 lines partially).  pa_dst: the pA array at the region's 8-sample base (wr.rb).
__device__ __forceinline__ void pa_write_tile(const WaveRead &wr, int t, const Scale &sc, float *pa_dst) {
    const int lane = lane_id();
    int q_lo, q_hi;
    wr.range(t, 0, q_lo, q_hi);
    const bool pa_interior = q_lo == 0 && q_hi == SS_TILE;
#pragma unroll 1
    for (int sub = 0; sub < SS_TILE / 256; ++sub) {
        const int qs = sub * 256 + lane * 4;
        int64_t pp = wr.rb + (int64_t)t * SS_TILE + qs;
        const int64_t last = wr.n_total - 4;
        pp = pp < last ? pp : last;
        const uint2 rw = *reinterpret_cast<const uint2 *>(wr.samples + pp);
        const float4 o = make_float4(to_pa((int16_t)(rw.x & 0xffffu), sc), to_pa((int16_t)(rw.x >> 16), sc),
                                     to_pa((int16_t)(rw.y & 0xffffu), sc), to_pa((int16_t)(rw.y >> 16), sc));
        float *dst = pa_dst + (int64_t)t * SS_TILE + qs;
        // (a group of four inside the region is stored whole in an edge tile as well: dst is 16-byte aligned)
        if (pa_interior || (qs >= q_lo && qs + 4 <= q_hi)) *reinterpret_cast<float4 *>(dst) = o;
        else {
            if (qs >= q_lo && qs < q_hi) dst[0] = o.x;
            if (qs + 1 >= q_lo && qs + 1 < q_hi) dst[1] = o.y;
            if (qs + 2 >= q_lo && qs + 2 < q_hi) dst[2] = o.z;
            if (qs + 3 >= q_lo && qs + 3 < q_hi) dst[3] = o.w;
        }
    }
}
// The same from the tile the wave already holds (round 5): the re-read above missed the L2 on 60 % of its lines at
// 125 000 x 100 000 (15 of 65 GB fetched; the 50 GB of pA stores go through the same L2), so the wave turns its tile into
// the stores' layout through 2 KB of LDS instead -- `tl`, the wave's histogram, which pass 1 does not use yet.
__device__ __forceinline__ void pa_write_tile_lds(const WaveRead &wr, int t, const Scale &sc, float *pa_dst, const WaveTile &cur,
                                                  uint32_t *tl) {
    const int lane = lane_id();
    int q_lo, q_hi;
    wr.range(t, 0, q_lo, q_hi);
    const bool pa_interior = q_lo == 0 && q_hi == SS_TILE;
    uint4 *row = reinterpret_cast<uint4 *>(tl) + lane * 2;
    row[0] = make_uint4(cur.w[0], cur.w[1], cur.w[2], cur.w[3]);
    row[1] = make_uint4(cur.w[4], cur.w[5], cur.w[6], cur.w[7]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int sub = 0; sub < SS_TILE / 256; ++sub) {
        const int qs = sub * 256 + lane * 4;
        const uint2 rw = *reinterpret_cast<const uint2 *>(tl + sub * 128 + lane * 2);
        const float4 o = make_float4(to_pa((int16_t)(rw.x & 0xffffu), sc), to_pa((int16_t)(rw.x >> 16), sc),
                                     to_pa((int16_t)(rw.y & 0xffffu), sc), to_pa((int16_t)(rw.y >> 16), sc));
        float *dst = pa_dst + (int64_t)t * SS_TILE + qs;
        if (pa_interior || (qs >= q_lo && qs + 4 <= q_hi)) *reinterpret_cast<float4 *>(dst) = o;
        else {
            if (qs >= q_lo && qs < q_hi) dst[0] = o.x;
            if (qs + 1 >= q_lo && qs + 1 < q_hi) dst[1] = o.y;
            if (qs + 2 >= q_lo && qs + 2 < q_hi) dst[2] = o.z;
            if (qs + 3 >= q_lo && qs + 3 < q_hi) dst[3] = o.w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next tile's rows are written behind these reads
    __builtin_amdgcn_wave_barrier();
}
// the histogram window of a read: WH_BINS raw values around its mean
__device__ __forceinline__ int hist_window_lo(float mraw) {
    const int c = (mraw == mraw) ? (int)fminf(fmaxf(mraw, -32768.0f), 32767.0f) : 0;
    const int lo = c - WH_BINS / 2;
    return lo < -32768 ? -32768 : (lo > 32768 - WH_BINS ? 32768 - WH_BINS : lo);
}
// this lane's 16 samples of a tile into the window histogram (LDS)
template <bool INTERIOR>
__device__ __forceinline__ void hist_tile(const WaveTile &cur, int q_lo, int q_hi, int lo, uint32_t *hist) {
    auto each = [&]<int E>(int16_t v, bool valid) {
        if (valid) {
            int b = (int)v - lo;
            b = b < 0 ? 0 : (b > WH_BINS - 1 ? WH_BINS - 1 : b);
            atomicAdd(&hist[b], 1u);
        }
    };
    wt_each_<0, INTERIOR>(cur, lane_id() * SS_SPL, q_lo, q_hi, each);
}
// The end of stat for one region, by one wave: the order statistics of ranks k (raw median) and, for a negative unit,
// n-1-k (the pA median's raw value) from the window histogram `hist` (LDS, complete and visible to this wave), and the
// record.  A median outside the window leaves the read flagged for k_median.
template <int MODE>
__device__ inline void stat_finish(const StatArgs &a, uint32_t r, const Region &g, const Scale &sc, int lo, const uint32_t *hist,
                                   float mraw, float mpa, float sdraw, float sdpa) {
    const int lane = lane_id();
    const int64_t k = g.len / 2;
    const bool mirrored = sc.unit < 0.0f && g.len - 1 - k != k;  // pA order is the reverse of the raw order
    constexpr int PER = WH_BINS / 64;
    uint32_t cnt[PER], lsum = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) { cnt[i] = hist[lane * PER + i]; lsum += cnt[i]; }
    const uint32_t incl = (uint32_t)wave_incl_scan_i((int)lsum), excl = incl - lsum;
    int found[2] = {0, 0};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const uint32_t rank = (uint32_t)(w ? (mirrored ? g.len - 1 - k : k) : k);
        int bin = 0;
        if (rank >= excl && rank < incl) {
            uint32_t acc = excl;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                if (rank >= acc && rank < acc + cnt[i]) bin = lane * PER + i;
                acc += cnt[i];
            }
        }
        const unsigned long long own = __ballot(rank >= excl && rank < incl);
        found[w] = own ? __builtin_amdgcn_readlane(bin, __builtin_amdgcn_readfirstlane(__ffsll((long long)own) - 1)) : 0;
    }
    const bool trusted = g.len > 0 && found[0] > 0 && found[0] < WH_BINS - 1 && found[1] > 0 && found[1] < WH_BINS - 1;
    if (lane == 0) {
        const int med = lo + found[0];
        const float pm = to_pa((int16_t)(lo + found[1]), sc);
        const bool pending = g.len > 0 && !trusted;
        if (MODE == REG_WHOLE) {
            sgk_stat_rec_t *o = a.stat + r;
            o->raw_mean = mraw; o->pa_mean = mpa; o->raw_std = sdraw; o->pa_std = sdpa;
            o->raw_median = g.len > 0 ? med : 0;
            o->pa_median = g.len > 0 ? pm : 0.0f;
            o->n = (uint32_t)g.len;
            o->reserved = pending ? FLAG_MEDIAN_WHOLE : 0u;
        } else if (MODE == REG_ADAPT) {
            a.prefix[r].adapt_mean = mpa;
            a.prefix[r].adapt_std = sdpa;
            if (g.len > 0) a.prefix[r].adapt_median = pm;
            if (pending) a.prefix[r].reserved |= FLAG_MEDIAN_ADAPT;
        } else {
            a.prefix[r].polya_mean = mpa;
            a.prefix[r].polya_std = sdpa;
            if (g.len > 0) a.prefix[r].polya_median = pm;
            if (pending) a.prefix[r].reserved |= FLAG_MEDIAN_POLYA;
        }
    }
}

#ifndef SGK_STAT_WAVES
#define SGK_STAT_WAVES 4  // waves per SIMD the register allocation aims at (5: spills, measured slower)
#endif
template <int MODE, bool PA>
__global__ __launch_bounds__(256, SGK_STAT_WAVES) void k_stat_wave(StatArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[4][WH_BINS];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t widx = blockIdx.x * 4 + wv;  // wave-uniform, and known to be: everything derived from it is scalar
    uint32_t r;
    if (MODE == REG_WHOLE && a.long_redo) {
        if (!long_redo_read(a, widx, r)) return;
    } else {
        if (widx >= a.b.n_reads) return;  // (no workgroup barrier anywhere in this kernel)
        r = a.order ? a.order[widx] : widx;
    }
    uint32_t *hist = hist_all[wv];
    const Region g = get_region(MODE, a.b, a.prefix, r);
    const Scale sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
    WaveRead wr;
    wr.init(a.b, g);
    const float nf = (float)(int)g.len;
    // a long read's record (and pA) is k_long_chains' work
    if (MODE == REG_WHOLE && !a.long_redo) {
        const LongSums *lg = find_long(a, r, g.len);
        if (lg && lg->rec_off != LC_NO_REC) return;
    }
#ifdef SGK_SS_COUNT  // development: what the four chains of read 0 had to do
    SsCount counts[4] = {};
#define SS_CNT(i) (&counts[i])
#else
#define SS_CNT(i) nullptr
#endif

    // ---- pass 1: sum of raw, sum of pA (oriented so that the running sum is non-negative); fused stat + pa: the pA of
    // every sample is written here, under the lighter arithmetic of the two passes
    // (both chains: a read whose running raw sum is negative -- signed ADC codes -- would otherwise fail the fast
    // walk's sign test on every tile and be added term by term)
    float m_raw = 0.0f, m_pa = 0.0f, sg = sc.unit < 0.0f ? -1.0f : 1.0f;
    int sraw = 0;
    {
        WaveTile cur, nxt;
        if (wr.ntiles > 0) wr.load(cur, 0);
        float *pa_dst = PA ? a.pa_out + wr.rb : nullptr;
        for (int t = 0; t < wr.ntiles; ++t) {
            if (t + 1 < wr.ntiles) wr.load(nxt, t + 1);
            const Scale so = {sc.offf, sc.unit * sg};
            if (PA) pa_write_tile_lds(wr, t, sc, pa_dst, cur, hist);
            ss_tile2<true>(
                m_raw, m_pa, wr, cur, t, [&](auto b) { return TermRaw<decltype(b)::interior>{b, sraw}; },
                [&](auto b) { return TermPa<decltype(b)::interior>{b, so}; }, SS_CNT(0), SS_CNT(1));
            if (m_pa < 0.0f) { m_pa = -m_pa; sg = -sg; }
            if (m_raw < 0.0f) { m_raw = -m_raw; sraw = ~sraw; }
            cur = nxt;
        }
    }
    // (a zero accumulator stands for +0: the reference's sum starts at +0 and x + (-x), +0 + -0 are +0 under
    // round-to-nearest, whichever way the chain was oriented)
    const float mraw = ss_signed(m_raw, sraw != 0) / nf;
    const float mpa = ss_signed(m_pa, sg < 0.0f) / nf;

    // ---- pass 2: squared deviations, window histogram
    const int lo = hist_window_lo(mraw);
#pragma unroll
    for (int i = 0; i < WH_BINS / 64; ++i) hist[i * 64 + lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float q_raw = 0.0f, q_pa = 0.0f;
    {
        WaveTile cur, nxt;
        if (wr.ntiles > 0) wr.load(cur, 0);
        for (int t = 0; t < wr.ntiles; ++t) {
            if (t + 1 < wr.ntiles) wr.load(nxt, t + 1);
            int q_lo, q_hi;
            wr.range(t, 0, q_lo, q_hi);
            if (wr.interior(t)) hist_tile<true>(cur, q_lo, q_hi, lo, hist);
            else hist_tile<false>(cur, q_lo, q_hi, lo, hist);
            ss_tile2<false>(
                q_raw, q_pa, wr, cur, t, [&](auto b) { return TermDevRaw<decltype(b)::interior>{b, mraw}; },
                [&](auto b) { return TermDevPa<decltype(b)::interior>{b, sc, mpa}; }, SS_CNT(2), SS_CNT(3));
            cur = nxt;
        }
    }
    const float sdraw = sqrtf(q_raw / nf), sdpa = sqrtf(q_pa / nf);

    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef SGK_SS_COUNT
    if (lane == 0 && r == 0)
        for (int i = 0; i < 4; ++i)
            printf("chain %d: tiles %u generic %u walks %u crossings %u composes %u serial %u\n", i, counts[i].tiles,
                   counts[i].generic, counts[i].walks, counts[i].crossings, counts[i].composes, counts[i].serial);
#endif
    stat_finish<MODE>(a, r, g, sc, lo, hist, mraw, mpa, sdraw, sdpa);
}

// ---------------------------------------------------------------- jnn_core automaton (src/jnn.c:190-278)
struct JnnAuto {
    float top, bot, first_min;
    int window, error, seg_dist;
    int hi_i, lo_i;  // integer form of the thresholds for integer-valued samples: in  <=>  lo_i < iv < hi_i
    int first_min_i; // (float)c >= first_min  <=>  c >= first_min_i
    int keep_min;    // an ended segment is kept iff c >= keep_min
    int open_m;      // -1 while a segment is open, else 0 (all predicates are kept as 0 / -1 lane masks)
    int err, run_err, c, w, start, nseg, last_x, last_y;
    __device__ void init(float top_, float bot_, int corrector, int seg_dist_, int window_, float stall_len, int error_) {
        top = top_; bot = bot_; window = window_; error = error_; seg_dist = seg_dist_;
        first_min = (float)window_ * stall_len;
        first_min_i = (int)ceilf(first_min);
        keep_min = first_min_i < window_ ? first_min_i : window_;  // c >= window || (nseg == 0 && c >= first_min_i)
        // v < top <=> iv < ceil(top), v > bot <=> iv > floor(bot) for an integer iv in [0, 1200]; NaN thresholds
        // compare false with everything
        hi_i = (top_ != top_) ? -0x40000000 : (top_ > 4000.0f ? 4000 : (top_ < -4.0f ? -4 : (int)ceilf(top_)));
        lo_i = (bot_ != bot_) ? 0x40000000 : (bot_ > 4000.0f ? 4000 : (bot_ < -4.0f ? -4 : (int)floorf(bot_)));
        open_m = 0; err = 0; run_err = 0; c = 0; w = corrector; start = 0; nseg = 0; last_x = 0; last_y = 0;
    }
    // in-range test of a clamped raw sample as a lane mask (no compare -> scalar-mask -> select round trips)
    __device__ __forceinline__ int in_mask_raw(int iv) const { return ((iv - hi_i) & (lo_i - iv)) >> 31; }
    __device__ __forceinline__ int in_mask_f(float v) const { return ((v < top) & (v > bot)) ? -1 : 0; }

    // One sample of jnn_core (src/jnn.c:213-271).  emit(k, x, y) is called when segment k can no longer change.
    // The lanes of a wave run different reads, and a lone wave spends its time waiting on dependent
    // compare -> SGPR -> select chains, so the per-sample bookkeeping is integer mask algebra on the vector
    // unit (0 / -1 masks, "x - mask" adds one); one wave-level test guards the two rare events (the c % w
    // correction and the end of a segment).
    template <typename E>
    __device__ __forceinline__ void step(int i, int in, E emit) {
        const int opn = open_m;
        const int errlt = (err - error) >> 31;           // err < error
        const int tol = ~in & opn & errlt;               // tolerated out-of-range sample
        const int rest = ~in & opn & ~errlt;             // the segment ends (closed or abandoned)
        const int cnt = in | tol;
        const int opening = in & ~opn;
        start = (opening & i) | (~opening & start);
        const int c1 = c - cnt;
        const int w1 = w - in;
        int err1 = err - tol;
        run_err = (run_err - tol) & ~in;
        // "if (c >= window && c >= w && c % w == 0) err--" (jnn.c:228, 238): c >= w needs more tolerated
        // samples in the segment than in-range samples before it
        // cheap necessary conditions, evaluated on every sample: c1 >= w1 for the correction, and for keeping an
        // ended segment c >= keep_min (= window, or min(window, first_min_i) while no segment has been kept yet)
        const int fix = cnt & ((w1 - 1 - c1) >> 31);
        const int keep = rest & ((keep_min - 1 - c) >> 31);
        if (__any((fix | keep) != 0)) {
            if (fix && c1 >= window) {
                if ((c1 % w1) == 0) --err1;
            }
            if (keep) {
                const int end = i - run_err;
                if (nseg > 0 && start - last_y < seg_dist) {
                    last_y = end;
                } else {
                    if (nseg > 0) emit(nseg - 1, last_x, last_y);
                    last_x = start; last_y = end;
                    ++nseg;
                }
                keep_min = window;  // "first segment" rule (jnn.c:243) no longer applies
            }
        }
        open_m = (open_m | in) & ~rest;
        c = c1 & ~rest;
        err = err1 & ~rest;
        run_err &= ~rest;
        w = w1;
    }
    template <typename E>
    __device__ void finish(E emit) {
        if (nseg > 0) emit(nseg - 1, last_x, last_y);
    }
};

JnnP jnn_preset(int rna) {  // JNNV1_DRNA_R9_PARAM / JNNV1_CDNA_R9_PARAM, src/jnn.h:29-49
    JnnP p;
    p.std_scale = 0.75f; p.corrector = 50; p.seg_dist = 50; p.error = 5; p.top = 0.0f; p.bot = 0.0f;
    if (rna) { p.window = 1000; p.stall_len = 1.0f; }
    else { p.window = 150; p.stall_len = 0.25f; }
    return p;
}
__host__ __device__ inline JnnP jnn_polya_params() {  // JNNV1_R9_POLYA == JNNV1_RNA004_POLYA, src/jnn.h:52-72
    return JnnP{-1.0f, 50, 200, 250, 1.0f, 30, 0.0f, 0.0f};
}
JnnP jnn_polya_preset() {  // src/jnn.h:52-72
    JnnP p;
    p.std_scale = -1.0f; p.corrector = 50; p.seg_dist = 200; p.window = 250; p.stall_len = 1.0f; p.error = 30;
    p.top = 0.0f; p.bot = 0.0f;
    return p;
}
AdaptP adaptor_preset(int pore) {  // JNNV2_RNA_R9_ADAPTOR / JNNV2_RNA_RNA004_ADAPTOR, src/jnn.h:84-98
    AdaptP p;
    p.std_scale = (pore == SGK_PORE_RNA004) ? 0.7f : 0.5f;
    p.seg_dist = 1500;
    p.lo_thresh = (pore == SGK_PORE_RNA004) ? 500 : 2000;
    p.hi_thresh = 200000;
    return p;
}

// jnn_raw (src/jnn.c:282-293): jnn_core over rm_outlier(raw) with any jnn_param_t
__global__ __launch_bounds__(64) void k_jnn(StatArgs a, JnnP p) {
    __shared__ __attribute__((aligned(16))) char lds[Stream1::LDS_BYTES];
    const uint32_t r = blockIdx.x * 64 + lane_id();
    const bool valid = r < a.b.n_reads && (!a.jnn_redo || a.n_segs[r] == JNN_REDO_MARK);
    if (a.jnn_redo && !__any(valid)) return;
    Region g = {0, 0};
    if (valid) g = get_region(REG_WHOLE, a.b, nullptr, r);
    int skip;
    Stream1 rs = make_stream(lds, a.b, g.start, valid && g.len > 0, skip);
    float top = p.top, bot = p.bot;
    if (p.std_scale > 0.0f) {  // src/jnn.c:195-199
        const float nf = (float)(int)g.len;
        float s = 0.0f;
        sweep_rows(rs, skip, g.len, [&](int64_t, int16_t v) { s = s + clampf_raw(v); });
        const float mn = s / nf;
        float q = 0.0f;
        sweep_rows(rs, skip, g.len, [&](int64_t, int16_t v) {
            const float d = clampf_raw(v) - mn;
            q = q + d * d;
        });
        const float sd = sqrtf(q / nf);
        const float band = sd * p.std_scale;
        top = mn + band;
        bot = mn - band;
    }
    JnnAuto A;
    A.init(top, bot, p.corrector, p.seg_dist, p.window, p.stall_len, p.error);
    const uint64_t slot0 = valid ? a.seg_slots[r] : 0, cap = valid ? a.seg_slots[r + 1] - slot0 : 0;
    bool overflow = false;
    auto emit = [&](int k, int x, int y) {
        if ((uint64_t)k < cap) { a.seg_x[slot0 + k] = x; a.seg_y[slot0 + k] = y; }
        else overflow = true;
    };
    sweep_rows_parts<16>(rs, skip, g.len, [&](int64_t j, int16_t v) { A.step((int)j, A.in_mask_raw(clampi_raw(v)), emit); });
    A.finish(emit);
    if (valid) a.n_segs[r] = (uint32_t)A.nseg;
    if (overflow) atomicAdd(a.err_count, 1u);
}

// ---------------------------------------------------------------- jnn_raw, one WAVE per read
// jnn_core's state (open / err / prev_err / c) is reset whenever a segment ends, and every open segment ends inside a
// streak of more than `error` consecutive out-of-range samples: behind such a streak the automaton is closed, whatever
// happened before it -- SYNC POINTS that depend on the data alone.  The read is cut into 64 chunks; lane c scans from
// the nominal start of chunk c to the first sync point at or behind it, runs the automaton from there (closed) to the
// first sync point at or behind the nominal start of chunk c + 1, where lane c + 1 has started: exact, no speculation,
// nothing to verify; reads without such streaks degenerate to fewer, longer lane runs.  (Valid while the `err--`
// correction of jnn.c:228,238 cannot fire, i.e. error < corrector as in every preset; other parameters use k_jnn.)
// Thresholds: the sequential float moments of the clamped signal through seqsum.h (two coalesced passes).  A lane
// stages its kept segments in its own part of the upper half of the read's slots; the merge (src/jnn.c:246-258) is a
// flag scan: a kept segment opens a new merged segment iff its start is seg_dist or more behind the previous end.
template <bool INTERIOR>
struct TermClamp {  // rm_outlier(raw), src/jnn.c:61-77
    TermBase<INTERIOR> b;
    __device__ __forceinline__ TermClamp with(uint32_t z) const { TermClamp r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        return b.template valid<E>() ? b.template clamped<E>() : 0.0f;
    }
};
template <bool INTERIOR>
struct TermDevClamp {  // (rm_outlier(raw) - mean)^2
    TermBase<INTERIOR> b;
    float mean;
    __device__ __forceinline__ TermDevClamp with(uint32_t z) const { TermDevClamp r = *this; r.b.z = z; return r; }
    template <int E>
    __device__ __forceinline__ float get() const {
        const float d = b.template clamped<E>() - mean;
        return b.template valid<E>() ? d * d : 0.0f;
    }
};
// one tile of one chain (see ss_tile2)
template <bool NEG, typename MK>
__device__ __forceinline__ void ss_tile1(float &m, const WaveRead &wr, const WaveTile &cur, int t, MK mk) {
    const int q0 = lane_id() * SS_SPL;
    int q_lo, q_hi;
    if (t == 0) {
        wr.range(0, 0, q_lo, q_hi);
        const int qh = q_lo + wr.head();
        if (qh > q_lo) m = ss_serial(m, mk(TermBase<false>{cur, q0, q_lo, qh, 0u}), q_lo / SS_SPL, (qh - 1) / SS_SPL);
    }
    wr.range(t, t == 0 ? wr.head() : 0, q_lo, q_hi);
    // (q0 of the edge branch through an opaque copy made there: its 32 validity compares were otherwise evaluated in
    // front of the branch, on every tile -- a third of the vector time of an interior tile's walk, round 5)
    SsWalk w;
    if (wr.interior(t)) w = ss_walk<NEG>(m, mk(TermBase<true>{cur, q0, q_lo, q_hi, 0u}));
    else w = ss_walk<NEG>(m, mk(TermBase<false>{cur, q0 + ss_edge_zero(), q_lo, q_hi, 0u}));
    int sk;
    if (ss_fast<NEG>(m, w, mk(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), sk))
        m = ss_finish<NEG>(m, mk(TermBase<false>{cur, q0, q_lo, q_hi, 0u}), w, sk);
}

constexpr int JW_BLOCK = 32;  // samples a lane takes per step of the chunked pass

// The chunked pass of jnn_core shared by k_jnn_wave and k_polya_wave: in <=> lo_r < raw < hi_r; `candidate(x, y, c)`
// is called, per lane in sample order, for every segment that ended after c >= keep_min samples.
__device__ __forceinline__ int jnn_chunk_lanes(int64_t nq) { return nq >= 512 ? (nq / 256 >= 64 ? 64 : (int)(nq / 256)) : 1; }
// (C chunks in all; this wave's lane l takes chunk gchunk0 + l: one wave per read has C <= 64 and gchunk0 = 0, the waves
// of a long read share its C = 64 x waves chunks)
// ... and of a long read on LC_WAVES waves: chunks of at least 512 samples, at most 64 per wave
__device__ __forceinline__ int jnn_long_chunks(int64_t nq) {
    const int64_t c = nq / 512;
    const int lanes = jnn_chunk_lanes(nq);
    return c > 64 * LC_WAVES ? 64 * LC_WAVES : (c > lanes ? (int)c : lanes);
}
// slots per chunk of a long read's staging area (the upper half of its slots); below 4 the read stays with k_jnn_wave
__device__ __forceinline__ uint32_t jnn_long_cap(const StatArgs &a, uint32_t r, int64_t nq) {
    const uint64_t cap = a.seg_slots[r + 1] - a.seg_slots[r];
    return (uint32_t)((cap - cap / 2) / (uint32_t)jnn_long_chunks(nq));
}
template <typename CAND>
__device__ __forceinline__ void jnn_chunks(const WaveRead &wr, int64_t n, int hi_r, int lo_r, int error, int keep_min,
                                           CAND &candidate, int C, int gchunk0) {
    const int lane = lane_id();
    const int E1 = error + 1;
    // ---- chunks in q space (q = sample index + wr.skip; chunk bounds are multiples of 8 -> 16-byte aligned loads)
    const int64_t nq = wr.skip + n;
    const int64_t K = ((nq + C - 1) / C + 7) & ~(int64_t)7;
    const int LEAD = (E1 + 7) & ~7;
    const int gc = gchunk0 + lane;
    const bool active = gc < C;
    const int64_t cs = (int64_t)gc * K, ce = cs + K;             // nominal chunk of this lane
    int64_t qb = gc == 0 ? 0 : cs - LEAD;                        // where this lane starts reading
    int runm = (gc == 0) ? -1 : 0, srchm = (active && gc != 0) ? -1 : 0;  // -1 / 0 lane masks
    int opn = 0, err = 0, run = 0, start = 0, oc = 0;
    // A block of 32 samples as bit masks (bit e: sample e is in / out of range; samples outside the read are neither).
    // The automaton goes from EVENT to event -- a segment opens at the next set bit of `inm`; it ends at the
    // (error + 1 - err)-th set bit of `outm` behind that -- instead of sample by sample: a segment lives for ~13 samples
    // on nanopore data, so a block holds a handful of events.  Positions [lo, hi) of the block belong to this lane's run.
    auto run_block = [&](uint32_t inm, uint32_t outm, int i0, int lo, int hi) {
        const uint32_t range = (lo >= 32 ? 0u : (0xffffffffu >> lo) << lo) & (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u));
        inm &= range;
        outm &= range;
        int pos = lo;
        for (;;) {
            const uint32_t keep = pos >= 32 ? 0u : (0xffffffffu >> pos) << pos;  // bits at positions >= pos
            if (!opn) {
                const uint32_t m = inm & keep;
                if (!m) break;
                const int e = __ffs((int)m) - 1;
                start = i0 + e; opn = -1; err = 0; run = 0; pos = e + 1;
            } else {
                uint32_t mo = outm & keep;
                const int need = error - err + 1;
                const int pc = __popc(mo);
                if (pc < need) {
                    err += pc;
                    const uint32_t mi = inm & keep;
                    run = mi ? __clz((int)mi) - (32 - hi) : run + (hi - pos);
                    break;
                }
                for (int k = 1; k < need; ++k) mo &= mo - 1u;
                const int e = __ffs((int)mo) - 1;
                const uint32_t mi = inm & keep & ((1u << e) - 1u);  // in-range samples in [pos, e)
                const int perr = mi ? e - (32 - __clz((int)mi)) : run + (e - pos);
                const int i = i0 + e;
                if (i - start >= keep_min) candidate(start, i - perr, i - start);
                opn = 0; err = 0; run = 0; pos = e + 1;
            }
        }
    };
    // bit e set: sample e ends a streak of at least E1 out-of-range samples (oc_in of them in front of the block):
    // behind it the automaton is closed (E1 <= 32)
    auto sync_bits = [&](uint32_t outm, int oc_in) -> uint32_t {
        const uint32_t prev = oc_in >= 32 ? 0xffffffffu : ~(0xffffffffu >> oc_in);  // the oc_in samples in front
        unsigned long long x = ((unsigned long long)outm << 32) | prev;
        for (int k = 1; k < E1;) {
            const int st = k < E1 - k ? k : E1 - k;
            x &= x << st;
            k += st;
        }
        return (uint32_t)(x >> 32);
    };

    // Every lane streams its own chunk, a whole 128-byte line (two blocks) per step, from a line boundary: the masks of
    // both blocks are formed first, the next line is requested into the registers that frees, and is in flight under
    // the automaton's two blocks.  (Round 5.  Until then a lane took 64 bytes per step with the next 64 in flight: the
    // two halves of a line were requested a whole step apart, 3.6 us during which 4 MB pass through an XCD's 4 MB L2,
    // and 60 % of the lines were fetched twice -- 15 of the kernel's 90 GB at 125 000 x 100 000, in a kernel that runs
    // at 5.8 TB/s.  Earlier attempts kept two whole lines per lane in registers: 115 registers instead of 89, four waves
    // per SIMD instead of five, 16.4 ms against 15.6; the LDS row stager of the lane-per-read kernels lost to its
    // barriers.)
    // lo_r < v < hi_r  <=>  lp1 <= v <= hm1 on int16 samples (thresholds beyond the int16 range: always / never)
    const bool never = hi_r <= -32768 || lo_r >= 32767;
    const int hm1_i = hi_r - 1 > 32767 ? 32767 : hi_r - 1, lp1_i = lo_r + 1 < -32768 ? -32768 : lo_r + 1;
    const s16x2 hm1 = {(short)hm1_i, (short)hm1_i}, lp1 = {(short)lp1_i, (short)lp1_i};
    auto spread16 = [](uint32_t x) {  // bit k -> bit 2k
        x = (x | (x << 8)) & 0x00FF00FFu;
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        return x;
    };
    constexpr int JW_LINE = 2 * JW_BLOCK;  // samples per 128-byte line
    qb -= (wr.rb + qb) & (int64_t)(JW_LINE - 1);  // (starting earlier only adds to what the sync search knows)
    uint32_t ln[JW_LINE / 2];
    auto load_line = [&](int64_t q) {
        const int64_t last = wr.n_total - 8;
#pragma unroll
        for (int v = 0; v < JW_LINE / 8; ++v) {
            int64_t pp = wr.rb + q + 8 * v;
            pp = pp < last ? pp : last;
            pp = pp < 0 ? 0 : pp;
            const uint4 u = *reinterpret_cast<const uint4 *>(wr.samples + pp);
            ln[4 * v] = u.x; ln[4 * v + 1] = u.y; ln[4 * v + 2] = u.z; ln[4 * v + 3] = u.w;
        }
    };
    // bit e of the result: lo_r < sample e < hi_r of the block in ln[16 h ..].  Two samples per instruction (one by one
    // this was 18 issue cycles per sample): saturating packed subtractions leave the sign of (hi_r - 1) - v and of
    // v - (lo_r + 1) in bits 15 / 31 of a word -- either set: out of range --, the words' flags are collected by
    // shifting (even samples in the low half, odd ones in the high half) and the two halves interleaved at the end.
    auto block_mask = [&](int h) -> uint32_t {
        uint32_t acc = 0u;
#pragma unroll
        for (int k = 0; k < JW_BLOCK / 2; ++k) {
            const s16x2 v = __builtin_bit_cast(s16x2, ln[h * (JW_BLOCK / 2) + k]);
            const uint32_t d = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(hm1, v)) |
                               __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(v, lp1));
            acc = (d & 0x80008000u) | ((acc >> 1) & 0x7fff7fffu);
        }
        return never ? 0u : ~(spread16(acc & 0xffffu) | (spread16(acc >> 16) << 1));
    };
    // the automaton over the block of 32 samples at q
    auto step = [&](uint32_t inm, int64_t q) {
        const bool busy = active && (srchm | runm) && q < nq;
        if (!busy) return;
        uint32_t vmask = 0xffffffffu;
        if (q < wr.skip || q + JW_BLOCK > nq) {  // a block on the read's edge
            const int64_t a0 = wr.skip - q, a1 = nq - q;
            const int lo = a0 < 0 ? 0 : (a0 > 32 ? 32 : (int)a0), hi = a1 > 32 ? 32 : (a1 < 0 ? 0 : (int)a1);
            vmask = (lo >= 32 ? 0u : (0xffffffffu >> lo) << lo) & (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u));
        }
        inm &= vmask;
        const uint32_t outm = ~inm & vmask;
        int lo = 0, hi = 32;
        bool ends_here = false;
        if (srchm || q + JW_BLOCK >= ce) {  // the sync logic is in play (the block holds sample ce - 1 or lies behind it)
            const uint32_t sy = sync_bits(outm, oc);
            if (srchm) {  // the run starts behind the first sync sample at position >= cs - 1
                const int64_t f = cs - q - 1;
                const uint32_t m = f >= 32 ? 0u : (f <= 0 ? sy : (sy >> f) << f);
                if (m) {
                    lo = __ffs((int)m);  // position behind that sample
                    srchm = 0;
                    runm = q + lo >= ce ? 0 : -1;
                } else lo = 32;
            }
            if (runm && q + JW_BLOCK >= ce) {  // ... and ends with the first sync sample at position >= ce - 1
                int64_t f = ce - q - 1;
                if (f < lo) f = lo;
                const uint32_t m = f >= 32 ? 0u : (f <= 0 ? sy : (sy >> f) << f);
                if (m) { hi = __ffs((int)m); ends_here = true; }  // (hi can be 32: the sync sample is the block's last)
            }
        }
        if (runm && lo < hi) run_block(inm, outm, (int)(q - wr.skip), lo, hi);
        if (ends_here) runm = 0;  // done
        const uint32_t stop = inm | ~vmask;  // samples that are not out of range
        oc = stop ? __clz((int)stop) : oc + 32;
    };
    load_line(qb);
    for (;;) {
        if (!__any(active && (srchm | runm) && qb < nq)) break;
        const uint32_t m0 = block_mask(0), m1 = block_mask(1);
        load_line(qb + JW_LINE);
        step(m0, qb);
        step(m1, qb + JW_BLOCK);
        qb += JW_LINE;
    }
}

// integer form of jnn_core's range test for thresholds top / bot (JnnAuto::init above), on the UNCLAMPED sample:
// lo_i < clamp(v) < hi_i  <=>  lo_r < v < hi_r; keep_min: the shortest segment that can matter
struct JnnThr {
    int hi_r, lo_r, keep_min;
};
__device__ __forceinline__ JnnThr jnn_thresholds(float top, float bot, const JnnP &p) {
    const int hi_i = (top != top) ? -0x40000000 : (top > 4000.0f ? 4000 : (top < -4.0f ? -4 : (int)ceilf(top)));
    const int lo_i = (bot != bot) ? 0x40000000 : (bot > 4000.0f ? 4000 : (bot < -4.0f ? -4 : (int)floorf(bot)));
    JnnThr t;
    t.hi_r = hi_i <= 0 ? -40000 : (hi_i > 1200 ? 40000 : hi_i);
    t.lo_r = lo_i >= 1200 ? 40000 : (lo_i < 0 ? -40000 : lo_i);
    const int first_min_i = (int)ceilf((float)p.window * p.stall_len);  // (float)c >= window * stall_len
    t.keep_min = first_min_i < p.window ? first_min_i : p.window;
    return t;
}

// The merge of the kept segments (src/jnn.c:246-258), 64 chunks per round, in chunk order.  A chunk's kept segments are
// [its first candidate, if that is strong or the first candidate of the read] + its staged strong ones; a kept segment
// opens a new merged segment iff its start is seg_dist or more behind the previous kept segment's end.  Between rounds
// the carry holds the last kept segment (its end is written once the next kept segment turns out to open a new merged
// one, or by jnn_merge_flush) and the number of merged segments so far.
struct JnnCarry {
    bool has, seen, overflow;  // a kept segment so far; a candidate so far; some slot range was too small
    int y;                     // end of the last kept segment
    uint32_t idx;              // merged segments opened so far
};
template <bool AGENT>
__device__ __forceinline__ int jnn_ld(const int32_t *p) {
    if constexpr (AGENT) return (int)lc_ld(reinterpret_cast<const uint32_t *>(p));
    else return *p;
}
template <bool AGENT>  // AGENT: the staged segments were written by other workgroups (agent-scope atomics)
__device__ inline void jnn_merge_round(JnnCarry &cy, int has_first, int fx, int fy, int fstrong, uint32_t cnt, uint32_t cap_l,
                                       const int32_t *stage_x, const int32_t *stage_y, int seg_dist, int32_t *out_x,
                                       int32_t *out_y, uint32_t half) {
    const int lane = lane_id();
    const unsigned long long hasf = __ballot(has_first != 0);
    const int firstlane = (!cy.seen && hasf) ? __ffsll((long long)hasf) - 1 : -1;
    const bool keep_first = has_first && (fstrong || lane == firstlane);
    bool overflow = cnt > cap_l;
    if (cnt > cap_l) cnt = cap_l;
    const uint32_t kcnt = cnt + (keep_first ? 1u : 0u);
    // y of the last kept segment of the nearest lane in front that has one (or the carry's)
    int last_y_own = 0;
    if (kcnt) last_y_own = cnt ? jnn_ld<AGENT>(stage_y + cnt - 1) : fy;
    const unsigned long long nonempty = __ballot(kcnt != 0u);
    const unsigned long long before = nonempty & ((1ull << lane) - 1ull);
    const int src_prev = before ? 63 - __clzll((long long)before) : 0;
    int prev_y_in = __shfl(last_y_own, src_prev, 64);
    bool has_prev = before != 0ull;
    if (!has_prev) { prev_y_in = cy.y; has_prev = cy.has; }
    auto entry = [&](uint32_t k, int &x, int &y) {
        if (keep_first) {
            if (k == 0) { x = fx; y = fy; return; }
            --k;
        }
        x = jnn_ld<AGENT>(stage_x + k); y = jnn_ld<AGENT>(stage_y + k);
    };
    // pass 1: how many merged segments start in this lane; is this lane's first kept segment one of them?
    uint32_t nnew = 0u;
    bool first_is_new = false;
    {
        int py = prev_y_in;
        bool hp = has_prev;
        for (uint32_t k = 0; k < kcnt; ++k) {
            int x, y;
            entry(k, x, y);
            const bool nw = !hp || !(x - py < seg_dist);
            if (k == 0) first_is_new = nw;
            nnew += nw ? 1u : 0u;
            py = y; hp = true;
        }
    }
    const uint32_t incl = (uint32_t)wave_incl_scan_i((int)nnew), base = cy.idx + incl - nnew;
    const uint32_t total = (uint32_t)wave_last_i((int)incl);
    // the end of the last kept segment in front of this round, if this round's first kept segment opens a new merged one
    const int firstne = nonempty ? __ffsll((long long)nonempty) - 1 : -1;
    if (cy.has && lane == firstne && first_is_new && cy.idx - 1u < half) out_y[cy.idx - 1u] = cy.y;
    // is the kept segment behind this lane's last one the start of a new merged segment?  (the round's last kept
    // segment: decided by the next round or the flush)
    const unsigned long long after = lane == 63 ? 0ull : (nonempty & ~((2ull << lane) - 1ull));
    const int src_next = after ? __ffsll((long long)after) - 1 : 0;
    const bool next_new = __shfl(first_is_new ? 1 : 0, src_next, 64) != 0 && after != 0ull;
    // pass 2: x of every segment that starts a merged one, y of every segment that ends one
    {
        int py = prev_y_in;
        bool hp = has_prev;
        uint32_t idx = base;  // merged segments started so far (in front of and inside this lane)
        int x = 0, y = 0;
        if (kcnt) entry(0, x, y);
        for (uint32_t k = 0; k < kcnt; ++k) {
            const bool nw = !hp || !(x - py < seg_dist);
            if (nw) {
                if (idx < half) out_x[idx] = x; else overflow = true;
                ++idx;
            }
            int xn = 0, yn = 0;
            bool ends;
            if (k + 1 < kcnt) {
                entry(k + 1, xn, yn);
                ends = !(xn - y < seg_dist);
            } else ends = next_new;
            if (ends && idx - 1 < half) out_y[idx - 1] = y;
            py = y; hp = true;
            x = xn; y = yn;
        }
    }
    if (nonempty) {
        cy.y = __builtin_amdgcn_readlane(last_y_own, 63 - __clzll((long long)nonempty));
        cy.has = true;
    }
    cy.idx += total;
    cy.seen = cy.seen || hasf != 0ull;
    cy.overflow = cy.overflow || __any(overflow);
}
// the end of the read's last kept segment; returns the number of merged segments (JNN_REDO_MARK: the slots did not do)
__device__ inline uint32_t jnn_merge_flush(const JnnCarry &cy, int32_t *out_y, uint32_t half) {
    if (cy.has && lane_id() == 0 && cy.idx - 1u < half) out_y[cy.idx - 1u] = cy.y;
    return (cy.overflow || cy.idx > half) ? JNN_REDO_MARK : cy.idx;
}

__global__ __launch_bounds__(256) void k_jnn_wave(StatArgs a, JnnP p) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t widx = blockIdx.x * 4 + wv;
    uint32_t r;
    if (a.long_redo) {
        if (!long_redo_read(a, widx, r)) return;
    } else {
        if (widx >= a.b.n_reads) return;
        r = a.order ? a.order[widx] : widx;
    }
    const Region g = get_region(REG_WHOLE, a.b, nullptr, r);
    const int64_t n = g.len;
    if (n <= 0) {
        if (lane == 0) a.n_segs[r] = 0u;
        return;
    }
    WaveRead wr;
    wr.init(a.b, g);
    float top = p.top, bot = p.bot;
    if (p.std_scale > 0.0f) {  // src/jnn.c:195-199
        const float nf = (float)(int)n;
        float s = 0.0f, q = 0.0f;
        // a long read is k_long_chains' (sums, automaton and merge) if its slots have room for 4 096 chunks' headers
        const LongSums *lg = a.long_redo ? nullptr : find_long(a, r, n);
        if (lg && lg->rec_off != LC_NO_REC && jnn_long_cap(a, r, wr.skip + n) >= 4u) return;
#ifndef SGK_JNN_ABL
#define SGK_JNN_ABL 0  // development: 1 / 2 / 4 leave out the pass of the sum / of the deviations / of the automaton
#endif
        if (!(SGK_JNN_ABL & 1)) {
            WaveTile cur, nxt;
            wr.load(cur, 0);
            for (int t = 0; t < wr.ntiles; ++t) {
                if (t + 1 < wr.ntiles) wr.load(nxt, t + 1);
                ss_tile1<false>(s, wr, cur, t, [&](auto b) { return TermClamp<decltype(b)::interior>{b}; });
                cur = nxt;
            }
        } else s = nf * 480.0f;
        const float mn = s / nf;
        if (!(SGK_JNN_ABL & 2)) {
            WaveTile cur, nxt;
            wr.load(cur, 0);
            for (int t = 0; t < wr.ntiles; ++t) {
                if (t + 1 < wr.ntiles) wr.load(nxt, t + 1);
                ss_tile1<false>(q, wr, cur, t, [&](auto b) { return TermDevClamp<decltype(b)::interior>{b, mn}; });
                cur = nxt;
            }
        } else q = nf * 2500.0f;
        const float band = sqrtf(q / nf) * p.std_scale;
        top = mn + band;
        bot = mn - band;
    }
    const JnnThr th = jnn_thresholds(top, bot, p);

    // ---- the automaton, in chunks between sync points (jnn_chunks); kept segments are staged in the upper half of the
    // read's slots (a part per lane), the merged segments go to the lower half
    const int64_t nq = wr.skip + n;
    const int C = jnn_chunk_lanes(nq);
    const uint64_t slot0 = a.seg_slots[r], cap = a.seg_slots[r + 1] - slot0;
    const uint32_t half = (uint32_t)(cap / 2), capL = (uint32_t)((cap - half) / (uint32_t)C);
    int32_t *stage_x = a.seg_x + slot0 + half + (uint64_t)lane * capL, *stage_y = a.seg_y + slot0 + half + (uint64_t)lane * capL;

    int fx = 0, fy = 0, fstrong = 0, has_first = 0;
    uint32_t cnt = 0u;

    // a segment that ended with c >= keep_min samples: the lane's first one is kept in registers (whether it is kept
    // depends on the lanes in front), later ones only matter if c >= window
    auto candidate = [&](int sx, int sy, int c) {
#ifdef SGK_JNN_DEBUG
        printf("cand lane %d: %d %d c %d\n", lane, sx, sy, c);
#endif
        const int strong = c >= p.window ? 1 : 0;
        if (!has_first) { has_first = 1; fx = sx; fy = sy; fstrong = strong; }
        else if (strong) {
            if (cnt < capL) { stage_x[cnt] = sx; stage_y[cnt] = sy; }
            ++cnt;  // (more than capL: jnn_merge_round reports the overflow)
        }
    };
    if (!(SGK_JNN_ABL & 4)) jnn_chunks(wr, n, th.hi_r, th.lo_r, p.error, th.keep_min, candidate, C, 0);

    // A lane's staging part is sized for chunks that end where they should; a read with too few sync points (a lane ran
    // on through many chunks and kept more segments than its part holds) is handed to the lane-per-read kernel instead
    JnnCarry cy = {false, false, false, 0, 0u};
    jnn_merge_round<false>(cy, has_first, fx, fy, fstrong, cnt, capL, stage_x, stage_y, p.seg_dist, a.seg_x + slot0,
                           a.seg_y + slot0, half);
    const uint32_t total = jnn_merge_flush(cy, a.seg_y + slot0, half);
    if (lane == 0) a.n_segs[r] = total;
}

// ---------------------------------------------------------------- find_polya, one WAVE per read
// jnn_pa on pA[adapt_y .. n) with the fixed thresholds of cfunc.c:191 and the polyA preset (src/jnn.h:52-72), first
// merged segment only.  x -> rm_outlierf(signal_in_picoamps(x)) is monotone in the raw value, so the in-range test
// bot < pA < top is an interval test on the raw sample (its bounds by bisection over the 65 536 raw values, with the
// float expression itself): no pA is formed.  With error = 30 the automaton has next to no sync points (31 out-of-range
// samples in a row), so instead of chunks the wave goes through the tail tile by tile (64 x 16 samples, coalesced) with
// the automaton's state in scalars and JUMPS: a segment opens at the next in-range sample and ends at the
// (error + 1 - err)-th out-of-range sample behind it, both found on the tile's 16-bit lane masks (next set bit; k-th set
// bit by a popcount scan).  It stops as soon as the first merged segment can no longer change -- usually after a few
// tiles, where the lane-per-read kernel waits for the slowest of 64 reads.
// (jnn_chunks was tried first for this: its chunks degenerate, 11.3 ms against the lane kernel's 5.8 ms on 50 000 reads.)
// first set bit at tile-local position >= cur of the 1024-bit mask held as 16 bits per lane (-1: none)
__device__ __forceinline__ int mask_next(uint32_t m16, int cur) {
    const int lo = cur - lane_id() * SS_SPL;
    const uint32_t m = lo <= 0 ? m16 : (lo >= SS_SPL ? 0u : (m16 >> lo) << lo);
    const unsigned long long has = __ballot(m != 0u);
    if (!has) return -1;
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)has) - 1);
    return l * SS_SPL + __builtin_amdgcn_readlane(__ffs((int)m) - 1, l);
}
// last set bit at a tile-local position in [cur, hi) (-1: none)
__device__ __forceinline__ int mask_last(uint32_t m16, int cur, int hi) {
    const int lo = cur - lane_id() * SS_SPL, up = hi - lane_id() * SS_SPL;
    uint32_t m = lo <= 0 ? m16 : (lo >= SS_SPL ? 0u : (m16 >> lo) << lo);
    m = up >= SS_SPL ? m : (up <= 0 ? 0u : m & ((1u << up) - 1u));
    const unsigned long long has = __ballot(m != 0u);
    if (!has) return -1;
    const int l = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)has));
    return l * SS_SPL + __builtin_amdgcn_readlane(31 - __clz((int)m), l);
}

// set bits of the tile mask (16 bits per lane) at tile-local positions >= cur
__device__ __forceinline__ uint32_t mask_from(uint32_t m16, int cur) {
    const int lo = cur - lane_id() * SS_SPL;
    return lo <= 0 ? m16 : (lo >= SS_SPL ? 0u : (m16 >> lo) << lo);
}
// position of the k-th (k >= 1) set bit at a position >= cur, given that there are at least k
__device__ __forceinline__ int mask_select(uint32_t m16, int cur, int k) {
    uint32_t m = mask_from(m16, cur);
    const int c = __popc(m), incl = wave_incl_scan_i(c), excl = incl - c;
    const unsigned long long own = __ballot(excl < k && incl >= k);
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)own) - 1);
    int kk = k - __builtin_amdgcn_readlane(excl, l);  // 1 .. 16, wave-uniform
    for (; kk > 1; --kk) m &= m - 1u;
    return l * SS_SPL + __builtin_amdgcn_readlane(__ffs((int)m) - 1, l);
}

template <typename PRED>
__device__ inline int first_true_i16(PRED pred) {  // smallest v in [-32768, 32767] with pred(v), 32768 if none (pred monotone)
    int lo = -32768, hi = 32768;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pred(mid)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_polya_wave(StatArgs a) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t widx = blockIdx.x * 4 + wv;
    if (widx >= a.b.n_reads) return;
    const uint32_t r = a.order ? a.order[widx] : widx;
    const Region g = get_region(REG_TAIL, a.b, a.prefix, r);
    int px = -1, py = -1;
    if (g.len > 0) {
        const Scale sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
        const float mid = a.prefix[r].adapt_mean + 30.0f;
        const float top = mid + 20.0f, bot = mid - 20.0f;
        const JnnP pp = jnn_polya_params();
        auto f = [&](int v) { return clampf_pa(to_pa((int16_t)v, sc)); };
        int vlo, vhi;  // in range <=> vlo <= raw <= vhi
        if (sc.unit >= 0.0f) {
            vlo = first_true_i16([&](int v) { return f(v) > bot; });
            vhi = first_true_i16([&](int v) { return !(f(v) < top); }) - 1;
        } else {
            vlo = first_true_i16([&](int v) { return f(v) < top; });
            vhi = first_true_i16([&](int v) { return !(f(v) > bot); }) - 1;
        }
        // a NaN anywhere (unit, offset, thresholds) makes every comparison of the reference false: nothing is in range
        if (!(sc.unit == sc.unit) || !(sc.offf == sc.offf) || !(top == top)) { vlo = 1; vhi = 0; }
        WaveRead wr;
        wr.init(a.b, g);
        const int hi_r = vhi + 1, lo_r = vlo - 1;  // in range <=> lo_r < raw < hi_r
        // automaton state (wave-uniform): open, errors so far, trailing tolerated errors, start; first merged segment
        int opn = 0, err = 0, run = 0, start = 0, last_y = 0;
        bool found = false, done = false;
        WaveTile cur_t, nxt_t;
        wr.load(cur_t, 0);
        for (int t = 0; t < wr.ntiles && !done; ++t) {
            if (t + 1 < wr.ntiles) wr.load(nxt_t, t + 1);
            int q_lo, q_hi;
            wr.range(t, 0, q_lo, q_hi);
            uint32_t inm = 0u;
#pragma unroll
            for (int e = 0; e < SS_SPL; ++e) {
                const int iv = (e & 1) ? (int)(int16_t)(cur_t.w[e / 2] >> 16) : (int)(int16_t)(cur_t.w[e / 2] & 0xffffu);
                inm |= ((uint32_t)((iv - hi_r) & (lo_r - iv)) >> 31) << e;
            }
            const int l0 = q_lo - lane * SS_SPL, l1 = q_hi - lane * SS_SPL;
            uint32_t vm = l0 <= 0 ? 0xffffu : (l0 >= SS_SPL ? 0u : (0xffffu >> l0) << l0);
            vm = l1 >= SS_SPL ? vm : (l1 <= 0 ? 0u : vm & ((1u << l1) - 1u));
            inm &= vm;
            const uint32_t outm = ~inm & vm;
            const int jbase = t * SS_TILE - wr.skip;  // sample index (in the tail) of tile-local position 0
            int cur = q_lo;
            for (;;) {
                if (!opn) {
                    const int ps = mask_next(inm, cur);
                    if (ps < 0) break;
                    start = jbase + ps; opn = 1; err = 0; run = 0; cur = ps + 1;
                } else {
                    const int need = pp.error + 1 - err;
                    const int pc = wave_last_i(wave_incl_scan_i(__popc(mask_from(outm, cur))));
                    if (pc < need) {  // the segment outlives the tile
                        err += pc;
                        const int li = mask_last(inm, cur, q_hi);
                        run = li >= 0 ? q_hi - 1 - li : run + (q_hi - cur);
                        break;
                    }
                    const int pe = mask_select(outm, cur, need);
                    const int li = mask_last(inm, cur, pe);
                    const int perr = li >= 0 ? pe - 1 - li : run + (pe - cur);
                    const int i = jbase + pe, end = i - perr;
                    if (i - start >= pp.window) {  // kept (stall_len = 1: the first-segment rule is the same)
                        if (!found) { px = start; py = end; found = true; }
                        else if (start - last_y < pp.seg_dist) py = end;
                        else { done = true; break; }  // the first merged segment is final
                        last_y = end;
                    }
                    opn = 0; err = 0; run = 0; cur = pe + 1;
                }
            }
            cur_t = nxt_t;
        }
    }
    if (lane == 0) {
        a.prefix[r].polya_x = px;
        a.prefix[r].polya_y = py;
    }
}

// jnn_pa (src/jnn.c:295-306) on ONE float array: jnn_core over rm_outlierf(x).  A compatibility entry (the batched
// path never holds pA in memory); every lane of the single wave walks the array, lane 0 stores.
__global__ __launch_bounds__(64) void k_jnn_f32(const float *x, int64_t n, JnnP p, int32_t *seg_x, int32_t *seg_y,
                                                uint32_t cap, uint32_t *n_segs) {
    float top = p.top, bot = p.bot;
    if (p.std_scale > 0.0f) {
        float s = 0.0f;
        for (int64_t j = 0; j < n; ++j) s = s + clampf_pa(x[j]);
        const float mn = s / (float)(int)n;
        float q = 0.0f;
        for (int64_t j = 0; j < n; ++j) {
            const float d = clampf_pa(x[j]) - mn;
            q = q + d * d;
        }
        const float sd = sqrtf(q / (float)(int)n);
        top = mn + sd * p.std_scale;
        bot = mn - sd * p.std_scale;
    }
    JnnAuto A;
    A.init(top, bot, p.corrector, p.seg_dist, p.window, p.stall_len, p.error);
    bool overflow = false;
    const bool writer = lane_id() == 0;
    auto emit = [&](int k, int sx, int sy) {
        if ((uint32_t)k < cap) { if (writer) { seg_x[k] = sx; seg_y[k] = sy; } }
        else overflow = true;
    };
    for (int64_t j = 0; j < n; ++j) A.step((int)j, A.in_mask_f(clampf_pa(x[j])), emit);
    A.finish(emit);
    if (writer) {
        n_segs[0] = (uint32_t)A.nseg;
        n_segs[1] = overflow ? 1u : 0u;
    }
}

// meanf / stdvf / medianf (src/stat.h:17-27, 36-44, 56-63) of ONE float array, for the reference-signature shims:
// the sequential float sums on every lane of the wave (lane 0 stores), the order statistic of rank n/2 by a
// three-level (11 + 11 + 10 bit) radix select on the order-preserving integer image of the floats.
struct TermArr {  // terms kept in registers
    const float (&x)[SS_SPL];
    __device__ __forceinline__ TermArr with(uint32_t) const { return *this; }
    template <int E>
    __device__ __forceinline__ float get() const { return x[E]; }
};

// one seqsum.h chain over a float array: term(x[i]) for i = 0 .. n-1, by one wave (the other waves of the workgroup
// skip it).  Terms can have any sign: the chain is oriented by the sign of its accumulator, tiles holding a term of
// the other sign are added natively.
template <typename F>
__device__ float ss_chain_f32(const float *x, int n, F term) {
    const int lane = lane_id(), q0 = lane * SS_SPL;
    float m = 0.0f, sg = 1.0f;
    const int ntiles = (n + SS_TILE - 1) / SS_TILE;
    const int head = n < SS_HEAD ? n : SS_HEAD;
    for (int t = 0; t < ntiles; ++t) {
        float v[SS_SPL], y[SS_SPL];
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) {
            const int i = t * SS_TILE + q0 + e;
            v[e] = i < n ? term(x[i]) * sg : 0.0f;  // (* +-1: exact)
        }
        if (t == 0) {  // the head natively; its terms are zeroed for the tile chain
#pragma unroll
            for (int e = 0; e < SS_SPL; ++e) y[e] = (q0 + e < head) ? v[e] : 0.0f;
            if (head > 0) m = ss_serial(m, TermArr{y}, 0, (head - 1) / SS_SPL);
#pragma unroll
            for (int e = 0; e < SS_SPL; ++e) v[e] = (q0 + e < head) ? 0.0f : v[e];
        }
        const SsWalk w = ss_walk<true>(m, TermArr{v});
        int sk;
        if (ss_fast<true>(m, w, TermArr{v}, sk)) m = ss_finish<true>(m, TermArr{v}, w, sk);
        if (m < 0.0f) { m = -m; sg = -sg; }
    }
    return m == 0.0f ? 0.0f : m * sg;
}

__global__ __launch_bounds__(256) void k_stat_f32(const float *x, int n, float *out3) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t sel_prefix, sel_rank;
    if (threadIdx.x < 64) {  // wave 0: the sequential float sums through seqsum.h
        const float s = ss_chain_f32(x, n, [](float v) { return v; });
        const float mn = s / n;
        const float q = ss_chain_f32(x, n, [&](float v) { return (v - mn) * (v - mn); });
        if (threadIdx.x == 0) { out3[0] = mn; out3[1] = sqrtf(q / n); }
    }
    auto key = [](float f) -> uint32_t {
        const uint32_t u = __float_as_uint(f);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    uint32_t prefix = 0u, rank = (uint32_t)(n / 2);
    const int shifts[3] = {21, 10, 0};
    const int bitsn[3] = {11, 11, 10};
    uint32_t known = 0u;  // mask of the key bits fixed so far
    for (int lvl = 0; lvl < 3; ++lvl) {
        for (int k = threadIdx.x; k < 2048; k += 256) hist[k] = 0u;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            const uint32_t kk = key(x[j]);
            if ((kk & known) == prefix) atomicAdd(&hist[(kk >> shifts[lvl]) & ((1u << bitsn[lvl]) - 1u)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0u, b = 0u;
            const uint32_t nb = 1u << bitsn[lvl];
            for (; b < nb; ++b) {
                if (acc + hist[b] > rank) break;
                acc += hist[b];
            }
            sel_prefix = prefix | (b << shifts[lvl]);
            sel_rank = rank - acc;
        }
        __syncthreads();
        prefix = sel_prefix;
        rank = sel_rank;
        known |= ((1u << bitsn[lvl]) - 1u) << shifts[lvl];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
        out3[2] = __uint_as_float(u);
    }
}

// find_polya (src/jnn.c:352-374): first segment of jnn_pa on pA[adapt_y..n) with fixed thresholds
// top = (m_a+30)+20, bot = (m_a+30)-20 (src/cfunc.c:191); polyA preset src/jnn.h:52-72.
__global__ __launch_bounds__(64) void k_polya(StatArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[Stream1::LDS_BYTES];
    const uint32_t r = blockIdx.x * 64 + lane_id();
    const bool valid = r < a.b.n_reads;
    Region g = {0, 0};
    Scale sc = {0.0f, 1.0f};
    float m_a = 0.0f;
    if (valid) {
        g = get_region(REG_TAIL, a.b, a.prefix, r);
        sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
        m_a = a.prefix[r].adapt_mean;
    }
    int skip;
    Stream1 rs = make_stream(lds, a.b, g.start, valid && g.len > 0, skip);
    const float mid = m_a + 30.0f;
    JnnAuto A;
    const JnnP pp = {-1.0f, 50, 200, 250, 1.0f, 30, 0.0f, 0.0f};  // JNNV1_R9_POLYA, src/jnn.h:52-61
    A.init(mid + 20.0f, mid - 20.0f, pp.corrector, pp.seg_dist, pp.window, pp.stall_len, pp.error);
    int px = -1, py = -1;
    auto emit = [&](int k, int x, int y) {
        if (k == 0) { px = x; py = y; }
    };
    // (a lane whose first segment is final could stop; the sweep is wave-cooperative, so it just idles)
    sweep_rows_parts<16>(rs, skip, g.len, [&](int64_t j, int16_t v) {
        if (py < 0 || A.nseg < 2) A.step((int)j, A.in_mask_f(clampf_pa(to_pa(v, sc))), emit);
    });
    if (A.nseg == 1 || (A.nseg >= 2 && py < 0)) A.finish(emit);
    if (valid) {
        a.prefix[r].polya_x = px;
        a.prefix[r].polya_y = py;
    }
}

// ---------------------------------------------------------------- find_adaptor / jnnv2 (src/jnn.c:99-188)
// rolling window mean (jnn.c:20-56) of the clamped raw signal, its sequential float mean/std,
// then the below-threshold run finder with merging; first run with lo <= length <= hi.
struct RunFinder {
    int t_lt, t_gt;  // tot < t_lt  <=>  t < bot;   tot >= t_gt  <=>  t > bot   (see roll_threshold)
    int seg_dist, lo, hi;
    int in_run, start, end, nseg, last_x, last_y, ans_x, ans_y, found;
    __device__ void init(int t_lt_, int t_gt_, int seg_dist_, int lo_, int hi_) {
        t_lt = t_lt_; t_gt = t_gt_; seg_dist = seg_dist_; lo = lo_; hi = hi_;
        in_run = 0; start = 0; end = 0; nseg = 0; last_x = 0; last_y = 0; ans_x = 0; ans_y = 0; found = 0;
    }
    __device__ void settle() {  // the last segment can no longer change
        const int len = last_y - last_x;
        if (!found && !(len > hi) && !(len < lo)) { found = 1; ans_x = last_x; ans_y = last_y; }
    }
    __device__ __forceinline__ void step(int j, int tot) {
        const bool below = tot < t_lt, above = tot >= t_gt;
        // selects for the per-sample updates; only the (rare) end of a run branches
        start = (below & !in_run) ? j : start;
        end = (below & (in_run != 0)) ? j : end;
        if (above & (in_run != 0)) {
            if (nseg > 0 && start - last_y < seg_dist) last_y = end;
            else {
                if (nseg > 0) settle();
                last_x = start; last_y = end; ++nseg;
            }
            start = 0; end = 0; in_run = 0;
        } else {
            in_run = below ? 1 : in_run;
        }
    }
    __device__ void finish() { if (nseg > 0) settle(); }
};

constexpr int ADW = 2000;  // jnnv2 window (both presets, src/jnn.h:84-98)

// rolling_window's t_i = tt / w (src/jnn.c:20-56).  tt is a float holding an exact integer (< 2000*1200 < 2^24),
// so it is carried as an int here; the division by the constant 2000 is the correctly rounded three-operation
// form of tstat_math.h (verified exhaustively for every float >= 2^-100 by oracle/verify_math.cpp).
__device__ __forceinline__ float roll_mean(int tot) { return sgk_div_f32<ADW>((float)tot); }

// One rolling-window sweep: calls f(i, tot_i) for i = 0..m-1 (m = n - ADW) in order, tot_i = sum of the clamped
// samples x[i .. i+ADW).  The trailing edge is a second row stream whose base is shifted by 16 samples, so that
// its tiles line up with the leading stream's: trail tile = lead tile - 31 (ADW = 2000 = 31*64 + 16).
constexpr int PART = 16;  // samples handled per (rolled) inner iteration: keeps the unrolled bodies and the
                          // register footprint small (two streams, three sweeps, each in a full and an edge form)
// f(i, tot_i) is called for the window indices i = 0..m-1 only (m = n - ADW, as rolling_window's output length,
// src/jnn.c:20-56); indices are 32-bit (reads are < 2^31 samples, misc.c:20).
template <int K, typename F>
__device__ __forceinline__ void rolling_elems(const uint32_t (&wl)[PART / 2], const uint32_t (&wt)[PART / 2],
                                              int64_t il0, int64_t n, int &tot, F &f) {
    if constexpr (K < PART) {
        const int64_t il = il0 + K;  // lead index
        if (il >= 0 && il < n) {
            const int cl = clampi_raw(RowPrefetch::sample_part<K>(wl));
            if (il < ADW) {
                tot = tot + cl;
                if (il == ADW - 1) f(0, tot);
            } else {
                const int ct = clampi_raw(RowPrefetch::sample_part<K>(wt));
                tot = tot - ct;
                tot = tot + cl;
                if (il < n - 1) f((int)(il - ADW + 1), tot);  // the total after the last sample has no window
            }
        }
        rolling_elems<K + 1>(wl, wt, il0, n, tot, f);
    }
}
// all PART lead indices are in [ADW, n-1): no predicates.  Two samples per packed 16-bit instruction for the outlier
// clamp and the lead - trail difference (|difference| <= 1200 fits int16).
template <int K, typename F>
__device__ __forceinline__ void rolling_full(const uint32_t (&wl)[PART / 2], const uint32_t (&wt)[PART / 2], int i0,
                                             int &tot, F &f) {
    if constexpr (K < PART) {
        static_assert(K % 2 == 0, "pairs");
        const s16x2 d = clamp_raw2(wl[K / 2]) - clamp_raw2(wt[K / 2]);
        tot += (int)d.x;
        f(i0 + K, tot);
        tot += (int)d.y;
        f(i0 + K + 1, tot);
        rolling_full<K + 2>(wl, wt, i0, tot, f);
    }
}
struct NeverStop {
    __device__ bool operator()() const { return false; }
};
// `stop` is asked once per tile (this lane has nothing more to learn); the sweep ends early when every lane says so
template <typename F, typename S = NeverStop>
__device__ inline void sweep_rolling(RowPrefetch &lead, RowPrefetch &trail, int skip, int64_t n, F f, S stop = S()) {
    const int maxq = wave_max_i((int)(n > ADW ? skip + n : 0));
    const int ntiles = (maxq + TILE - 1) / TILE;
    if (ntiles == 0) return;
    constexpr int LAG = 31;  // tiles between the two streams
    int tot = 0;
    const int n32 = (int)n;
    lead.issue(0);
    lead.commit(0);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) lead.issue(t + 1);
        if (t + 1 >= LAG && t + 1 < ntiles) trail.issue(t + 1 - LAG);
#pragma unroll 1
        for (int h = 0; h < TILE / PART; ++h) {
            uint32_t wl[PART / 2], wt[PART / 2];
            lead.row_part<PART>(h, wl);
            if (t >= LAG) trail.row_part<PART>(h, wt);
            else {
#pragma unroll
                for (int k = 0; k < PART / 2; ++k) wt[k] = 0u;
            }
            const int il0 = t * TILE + h * PART - skip;
            // lanes whose read has ended (or is too short, or has not begun) sit the part out; the predicated form
            // is only needed while some lane crosses the start, the first full window or the end of its read
            const bool inside = il0 >= ADW && il0 + PART < n32;
            const bool outside = n32 <= ADW || il0 >= n32 || il0 + PART <= 0;
            if (__all(inside || outside)) {
                if (inside) rolling_full<0>(wl, wt, il0 - ADW + 1, tot, f);
            } else if (!outside) rolling_elems<0>(wl, wt, (int64_t)il0, n, tot, f);
        }
        if (__all(stop())) break;
        if (t + 1 < ntiles) lead.commit(t + 1);
        if (t + 1 >= LAG && t + 1 < ntiles) trail.commit(t + 1 - LAG);
    }
}

// smallest integer total whose rolling mean is >= x (resp. > x): roll_mean is non-decreasing in tot, so
// the run finder's float comparisons t < bot / t > bot (src/jnn.c:139-163) become integer comparisons
// tot < T_lt / tot >= T_gt.  Binary search over [0, 2000*1200].
__device__ inline int roll_threshold(float x, bool strict) {
    if (x != x) return strict ? 0x7fffffff : 0;  // NaN threshold: no t is < or > it
    int lo = 0, hi = ADW * 1200 + 1;  // answer in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float t = roll_mean(mid);
        const bool ok = strict ? (t > x) : (t >= x);   // NaN x: never ok -> hi stays -> nothing is >= / > x
        if (ok) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(64, 2) void k_adaptor(StatArgs a, AdaptP ap) {
    __shared__ __attribute__((aligned(16))) char lds[2 * Stream1::LDS_BYTES];
    const uint32_t r = blockIdx.x * 64 + lane_id();
    const bool valid = r < a.b.n_reads;
    Region g = {0, 0};
    if (valid) g = get_region(REG_WHOLE, a.b, nullptr, r);
    const int64_t n = g.len;
    const bool run = valid && n > ADW;
    int skip;
    Stream1 lead = make_stream(lds, a.b, g.start, run, skip);
    Stream1 trail;
    // trail position = lead position - 2000 = (row base - 16) + (q - 31*64): the same rows, shifted
    trail = lead;
    trail.lds = lds + Stream1::LDS_BYTES;
    trail.set_shift(-2);
    const int64_t m = n - ADW;
    const float mf = (float)(int)m;
    float s = 0.0f;
    sweep_rolling(lead, trail, skip, n, [&](int, int tot) { s = s + roll_mean(tot); });
    const float mn = s / mf;
    float q = 0.0f;
    sweep_rolling(lead, trail, skip, n, [&](int, int tot) {
        const float d = roll_mean(tot) - mn;
        q = q + d * d;
    });
    const float sd = sqrtf(q / mf);
    RunFinder F;
    const float bot = mn - sd * ap.std_scale;
    F.init(roll_threshold(bot, false), roll_threshold(bot, true), ap.seg_dist, ap.lo_thresh, ap.hi_thresh);
    // the answer is the first qualifying segment (the reference breaks out of its segment list, src/jnn.c:154-167),
    // and a segment is final once a later one has started without merging into it: nothing after that changes it
    sweep_rolling(lead, trail, skip, n, [&](int i, int tot) { F.step(i, tot); }, [&]() { return !run || F.found != 0; });
    F.finish();
    if (!valid) return;
    sgk_prefix_rec_t *o = a.prefix + r;
    o->n = (uint32_t)n;
    o->reserved = 0;
    o->polya_x = -1; o->polya_y = -1;
    o->adapt_mean = 0.0f; o->adapt_std = 0.0f; o->adapt_median = 0.0f;
    o->polya_mean = 0.0f; o->polya_std = 0.0f; o->polya_median = 0.0f;
    if (!run) { o->adapt_x = -1; o->adapt_y = -1; }
    else if (F.found) { o->adapt_x = F.ans_x + ADW / 2 - 1; o->adapt_y = F.ans_y + ADW / 2 - 1; }
    else { o->adapt_x = 0; o->adapt_y = 0; }
}

// ---------------------------------------------------------------- find_adaptor / jnnv2, one WAVE per read
// Same arithmetic as k_adaptor, laid out as k_stat_wave: tiles of 64 x SS_SPL window indices; a lane holds the trailing
// samples x[i..i+16) and the leading samples x[i+2000..i+2016) of its 16 indices, forms the 16 differences of the
// clamped values (packed 16-bit), a DPP scan of the lanes' difference sums gives every lane its first rolling total
// (integers: exact in any order), and the sequential float sums of the rolling means (src/jnn.c:106-107) advance through
// seqsum.h.  Three passes: sum of the means; sum of their squared deviations; the run finder (src/jnn.c:126-158), whose
// state only changes where the below / above-threshold flags flip: the wave jumps from flip to flip over 16-bit lane
// masks and stops at the first qualifying segment that can no longer change.
// rolling totals of this lane's 16 window indices.  T0: total of the tile's first index (wave-uniform), advanced to
// the next tile's.  d_lo: tile-local indices below it have no difference (they lie in front of the read).
// (CLAMPED: the tiles hold rm_outlier of the samples already, roll_sweep)
template <bool MASKED, bool CLAMPED = false>
__device__ __forceinline__ void roll_tile(const WaveTile &trail, const WaveTile &lead, int d_lo, int &T0, int (&tot)[SS_SPL]) {
    const int q0 = lane_id() * SS_SPL;
    int run = 0;
#pragma unroll
    for (int k = 0; k < SS_SPL / 2; ++k) {
        const s16x2 d = CLAMPED ? __builtin_bit_cast(s16x2, lead.w[k]) - __builtin_bit_cast(s16x2, trail.w[k])
                                : clamp_raw2(lead.w[k]) - clamp_raw2(trail.w[k]);
        int d0 = (int)d.x, d1 = (int)d.y;
        if (MASKED) {
            d0 = (q0 + 2 * k >= d_lo) ? d0 : 0;
            d1 = (q0 + 2 * k + 1 >= d_lo) ? d1 : 0;
        }
        tot[2 * k] = run;
        run += d0;
        tot[2 * k + 1] = run;
        run += d1;
    }
    const int incl = wave_incl_scan_i(run);
    const int base = T0 + incl - run;
#pragma unroll
    for (int e = 0; e < SS_SPL; ++e) tot[e] += base;
    T0 += wave_last_i(incl);
}

// total of the ADW clamped samples that start at tile-local position lo0 (< SS_TILE) of tile t: the rolling total of
// the window that starts there (tiles t and t + 1 hold all of it)
__device__ __forceinline__ int window_total(const WaveRead &wr, int t, int lo0) {
    const int q0 = lane_id() * SS_SPL;
    int part = 0;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        WaveTile w;
        wr.load(w, t + tt);
        const int lo = lo0 - tt * SS_TILE, hi = lo0 + ADW - tt * SS_TILE;
#pragma unroll
        for (int k = 0; k < SS_SPL / 2; ++k) {
            const s16x2 c = clamp_raw2(w.w[k]);
            part += (q0 + 2 * k >= lo && q0 + 2 * k < hi) ? (int)c.x : 0;
            part += (q0 + 2 * k + 1 >= lo && q0 + 2 * k + 1 < hi) ? (int)c.y : 0;
        }
    }
    return wave_last_i(wave_incl_scan_i(part));
}
// the trailing and the leading tile of window tile t
__device__ __forceinline__ void roll_load(const WaveRead &wr, WaveTile &x, WaveTile &y, int t) {
    const int64_t tile0 = wr.rb + (int64_t)t * SS_TILE;
    wt_load(x, wr.samples, wr.n_total, tile0);
    wt_load(y, wr.samples, wr.n_total, tile0 + ADW);
}
// one tile of a chain over the terms term(rolling total)
template <typename TERM>
__device__ __forceinline__ void roll_chain_tile(float &acc, const WaveRead &wr, int t, const int (&tot)[SS_SPL], TERM term) {
    const int q0 = lane_id() * SS_SPL;
    int q_lo, q_hi;
    float x[SS_SPL];
    if (t == 0) {  // head, natively
        wr.range(0, 0, q_lo, q_hi);
        const int qh = q_lo + wr.head();
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) x[e] = (q0 + e >= q_lo && q0 + e < qh) ? term(tot[e]) : 0.0f;
        if (qh > q_lo) acc = ss_serial(acc, TermArr{x}, q_lo / SS_SPL, (qh - 1) / SS_SPL);
    }
    wr.range(t, t == 0 ? wr.head() : 0, q_lo, q_hi);
    if (wr.interior(t)) {
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) x[e] = term(tot[e]);
    } else {
        const int q0e = q0 + ss_edge_zero();  // (keeps the compares inside this branch, see ss_tile1)
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) x[e] = (q0e + e >= q_lo && q0e + e < q_hi) ? term(tot[e]) : 0.0f;
    }
    const SsWalk w = ss_walk<false>(acc, TermArr{x});
    int sk;
    if (ss_fast<false>(acc, w, TermArr{x}, sk)) acc = ss_finish<false>(acc, TermArr{x}, w, sk);
}

// one sweep over the rolling totals of the windows of wr: f(t, tot) per tile; stops when f returns true.
// `ring` (round 5; nullptr: every trailing tile is loaded from memory): 3 x 2 KB of LDS of this wave's.  The trailing tile
// of window tile t + 1 is what the wave loaded as LEADING tiles t - 1 and t (ADW = 2 000 = 2 x 1 024 - 48 samples: lane l's
// 16 trailing samples are lane l + 3's of leading tile t - 1, the last three lanes' are lanes 0 .. 2's of tile t), so the
// leading tiles go through a ring and the trailing ones come out of it: at 125 000 x 100 000 the second read missed the L2
// for 36 % of its lines (67.9 GB fetched for two passes of 25 GB and a partial third).
constexpr int ROLL_RING_TILES = 3;
constexpr int ROLL_RING_BYTES = ROLL_RING_TILES * SS_TILE * (int)sizeof(int16_t);
static_assert(2 * SS_TILE - ADW == 3 * SS_SPL && ADW > SS_TILE, "the lane shift of the trailing tile");
template <typename F>
__device__ __forceinline__ void roll_sweep(const WaveRead &wr, int first_total, F f, uint4 *ring = nullptr) {
    int T0 = first_total;
    const int lane = lane_id();
    // the tiles are kept clamped (rm_outlier, two samples per instruction): a tile is clamped once, as a leading tile
    auto clamp_tile = [](WaveTile &x) {
#pragma unroll
        for (int k = 0; k < SS_SPL / 2; ++k) x.w[k] = __builtin_bit_cast(uint32_t, clamp_raw2(x.w[k]));
    };
    WaveTile tr, ld, trn, ldn;
    roll_load(wr, tr, ld, 0);
    clamp_tile(tr);
    clamp_tile(ld);
    for (int t = 0; t < wr.ntiles; ++t) {
        const bool more = t + 1 < wr.ntiles;
        const bool from_ring = ring != nullptr && t + 1 >= 2;
        if (ring) {
            uint4 *row = ring + (t % ROLL_RING_TILES) * (SS_TILE / 8) + lane * 2;
            row[0] = make_uint4(ld.w[0], ld.w[1], ld.w[2], ld.w[3]);
            row[1] = make_uint4(ld.w[4], ld.w[5], ld.w[6], ld.w[7]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (more) {
            const int64_t tile0 = wr.rb + (int64_t)(t + 1) * SS_TILE;
            if (!from_ring) wt_load(trn, wr.samples, wr.n_total, tile0);
            wt_load(ldn, wr.samples, wr.n_total, tile0 + ADW);
        }
        int tot[SS_SPL];
        if (t == 0 && wr.skip > 0) roll_tile<true, true>(tr, ld, wr.skip, T0, tot);
        else roll_tile<false, true>(tr, ld, 0, T0, tot);
        if (f(t, tot)) break;
        if (more) {
            clamp_tile(ldn);
            if (!from_ring) clamp_tile(trn);
        }
        if (more && from_ring) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int src = lane + 3;
            const uint4 *row = src < 64 ? ring + ((t + 2) % ROLL_RING_TILES) * (SS_TILE / 8) + src * 2   // tile t - 1
                                        : ring + (t % ROLL_RING_TILES) * (SS_TILE / 8) + (src - 64) * 2;  // tile t
            const uint4 q0 = row[0], q1 = row[1];
            trn.w[0] = q0.x; trn.w[1] = q0.y; trn.w[2] = q0.z; trn.w[3] = q0.w;
            trn.w[4] = q1.x; trn.w[5] = q1.y; trn.w[6] = q1.z; trn.w[7] = q1.w;
        }
        tr = trn; ld = ldn;
    }
    if (ring) {  // the next sweep's rows are written behind this one's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// what find_adaptor leaves in a read's record before anything is found (lane 0)
__device__ __forceinline__ void adaptor_init_rec(sgk_prefix_rec_t *o, int64_t n) {
    o->n = (uint32_t)n;
    o->reserved = 0;
    o->polya_x = -1; o->polya_y = -1;
    o->adapt_mean = 0.0f; o->adapt_std = 0.0f; o->adapt_median = 0.0f;
    o->polya_mean = 0.0f; o->polya_std = 0.0f; o->polya_median = 0.0f;
}
// jnnv2's thresholds from the two sums over the m rolling means (src/jnn.c:106-124) and its run finder (RunFinder above,
// src/jnn.c:126-167) from flip to flip, by one wave; writes adapt_x / adapt_y
__device__ inline void adaptor_find(const WaveRead &wr, int first_total, float s, float q, float mf, const AdaptP &ap,
                                    sgk_prefix_rec_t *o, uint4 *ring = nullptr) {
    const int lane = lane_id(), q0 = lane * SS_SPL;
    const float mn = s / mf;
    const float sd = sqrtf(q / mf);
    const float bot = mn - sd * ap.std_scale;
    const int t_lt = roll_threshold(bot, false), t_gt = roll_threshold(bot, true);
    int in_run = 0, start = 0, end = 0, nseg = 0, last_x = 0, last_y = 0, ans_x = 0, ans_y = 0, found = 0;
    auto settle = [&]() {
        const int len = last_y - last_x;
        if (!found && !(len > ap.hi_thresh) && !(len < ap.lo_thresh)) { found = 1; ans_x = last_x; ans_y = last_y; }
    };
    roll_sweep(wr, first_total, [&](int t, const int (&tot)[SS_SPL]) {
        int q_lo, q_hi;
        wr.range(t, 0, q_lo, q_hi);
        uint32_t bm = 0u, am = 0u;
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) {
            bm |= (uint32_t)((tot[e] - t_lt) >> 31) & (1u << e);       // tot < t_lt
            am |= ~(uint32_t)((tot[e] - t_gt) >> 31) & (1u << e);      // tot >= t_gt
        }
        const int lo = q_lo - q0, hi = q_hi - q0;
        uint32_t vm = lo <= 0 ? 0xffffu : (lo >= SS_SPL ? 0u : (0xffffu >> lo) << lo);
        vm = hi >= SS_SPL ? vm : (hi <= 0 ? 0u : vm & ((1u << hi) - 1u));
        bm &= vm; am &= vm;
        const int jbase = t * SS_TILE - wr.skip;  // window index of tile-local position 0
        int cur = 0;
        for (;;) {
            if (!in_run) {
                const int p = mask_next(bm, cur);
                if (p < 0) break;
                start = jbase + p; in_run = 1; cur = p + 1;
            } else {
                const int pa = mask_next(am, cur);
                const int pb = mask_last(bm, cur, pa < 0 ? SS_TILE : pa);
                if (pb >= 0) end = jbase + pb;
                if (pa < 0) break;
                if (nseg > 0 && start - last_y < ap.seg_dist) last_y = end;
                else {
                    if (nseg > 0) settle();
                    last_x = start; last_y = end; ++nseg;
                }
                start = 0; end = 0; in_run = 0; cur = pa + 1;
                if (found) break;
            }
        }
        return found != 0;
    }, ring);
    if (nseg > 0) settle();
    if (lane == 0) {
        if (found) { o->adapt_x = ans_x + ADW / 2 - 1; o->adapt_y = ans_y + ADW / 2 - 1; }
        else { o->adapt_x = 0; o->adapt_y = 0; }
    }
}

__global__ __launch_bounds__(256) void k_adaptor_wave(StatArgs a, AdaptP ap) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t widx = blockIdx.x * 4 + wv;
    uint32_t r;
    if (a.long_redo) {
        if (!long_redo_read(a, widx, r)) return;
    } else {
        if (widx >= a.b.n_reads) return;
        r = a.order ? a.order[widx] : widx;
    }
    __shared__ uint4 ring_all[4][ROLL_RING_BYTES / 16];
    uint4 *ring = ring_all[wv];
    const Region g = get_region(REG_WHOLE, a.b, nullptr, r);
    const int64_t n = g.len;
    // a long read is k_long_chains' (which runs beside this kernel): its sums, thresholds, run finder and record
    const LongSums *lg = a.long_redo ? nullptr : find_long(a, r, n);
    if (lg && lg->rec_off != LC_NO_REC) return;
    sgk_prefix_rec_t *o = a.prefix + r;
    if (lane == 0) adaptor_init_rec(o, n);
    if (n <= ADW) {  // "Not enough data to trim", src/jnn.c:173-177
        if (lane == 0) { o->adapt_x = -1; o->adapt_y = -1; }
        return;
    }
    const int64_t m = n - ADW;  // number of rolling means
    WaveRead wr;
    wr.init(a.b, Region{g.start, m});

    // total of the first window: clamped samples 0 .. 1999 (tile-local positions skip .. skip + 1999 of tiles 0 and 1)
    const int first_total = window_total(wr, 0, wr.skip);
    const float mf = (float)(int)m;
    float s = 0.0f;
    roll_sweep(wr, first_total, [&](int t, const int (&tot)[SS_SPL]) {
        roll_chain_tile(s, wr, t, tot, [](int v) { return roll_mean(v); });
        return false;
    }, ring);
    const float mn = s / mf;
    float q = 0.0f;
    roll_sweep(wr, first_total, [&](int t, const int (&tot)[SS_SPL]) {
        roll_chain_tile(q, wr, t, tot, [&](int v) { const float d = roll_mean(v) - mn; return d * d; });
        return false;
    }, ring);
    adaptor_find(wr, first_total, s, q, mf, ap, o, ring);
}

// ---------------------------------------------------------------- long reads: the sequential sums on 64 wavefronts
// A wave evaluates a sequential float sum at ~1 000 terms per microsecond; a read of 3 000 000 samples keeps its wave
// busy for milliseconds per sum while the rest of the batch is long done.  k_long_chains gives such a read LC_PARTS
// workgroups of four wavefronts (on LC_PARTS compute units: the sums are bound by vector-instruction issue, one compute
// unit's four SIMDs would not do).  What seqsum.h does with the 16 terms of a lane is done here once more with the 1 024
// terms of a TILE (tools/proto/seqsum_tiles_proto.py is the numpy model, seqsum_segments_proto.py the round-3 sketch it
// grew from):
//
//   level 1, all waves, no dependency between them: a wave owns a contiguous run of tiles.  It PREDICTS the accumulator
//     in front of each tile (sums of the terms in front of its run, pass A below, then tile by tile from its own
//     summaries), takes the binade E of the prediction, and summarises the tile for that binade: T0 / T1, the
//     increment of the accumulator's significand over the tile's 1 024 terms when it enters the tile even / odd (lanes'
//     surrogate walks, parity maps composed across the lanes, once per entering parity).  8 bytes per tile and sum in
//     the workspace.  Tiles the argument does not cover (surrogates left the binade, a negative term, tile 0 with its
//     native head) are marked instead.
//   level 2, one wave per sum, 64 tiles per step: the summaries are composed exactly as ss_fast composes lanes -- parity
//     maps by the segmented xor scan, increments by a sum scan, S + total <= 2^24 certifies that the true sum stayed in
//     the binade.  A tile whose binade was predicted wrongly (E differs from the true accumulator's), in which the sum
//     leaves its binade, or that is marked, is evaluated from the TRUE accumulator with the wave kernels' own tile
//     routine (ss_tile1 / roll_chain_tile): about log2(n / 256) + a few tiles per sum.
//
// Nothing is speculative in the result: a wrong prediction costs a tile evaluation, never a wrong bit.  The workgroups
// of a read meet at barriers on a counter in the workspace (three per stage); everything they exchange is written and
// read with agent-scope atomics (the L2s of the eight XCDs are not coherent for ordinary accesses).  A read's workgroups
// are neighbours in the grid and the grid is small enough to be resident at once, so nobody waits for a workgroup that
// cannot start.  The four (stat), two (jnn, prefix) sums of a read land in LongSums; k_stat_wave / k_jnn_wave /
// k_adaptor_wave pick them up (find_long) and walk the read only for the histogram / pA output, the automaton (already
// 64 chunks wide), the run finder.
constexpr uint32_t LC_VALID = 0x00800000u;

template <int C>
struct Ix {
    static constexpr int v = C;
};
__device__ __forceinline__ double wave_sum_d(double v) { return wave_last_d(wave_incl_scan_d(v)); }
__device__ __forceinline__ int wave_sum_i(int v) { return wave_last_i(wave_incl_scan_i(v)); }
struct LcCtx {
    LongWork *w;
    LongHdr *hdr;
    unsigned long long *rec[2];  // tile records of the two sums: T0 | (E << 24 | LC_VALID | (T1 - T0 + 0x8000) & 0xffff) << 32
    uint32_t phase;              // barriers passed
    int part;                    // this workgroup's index among the read's LC_PARTS
    uint32_t fault;              // StatArgs::long_fault
};
// All workgroups of the read; what they wrote with lc_st before is readable with lc_ld behind it.  Returns false when
// the read is DECLINED: this workgroup waited in vain (the bound -- seconds -- keeps a GPU that does not dispatch a
// grid's workgroups in order, or shares its slots with something that does not end, from hanging) or another one of the
// read did and said so in LongWork::failed.  Every workgroup then leaves the read without writing anything of the
// subtool's output (all of it is written behind a read's LAST barrier: whoever passes that one has seen all LC_PARTS
// arrive, so what it writes is right even if a late workgroup flagged the read meanwhile), and the redo launch of the
// wave kernel (StatArgs::long_redo, behind the join) takes the read on one wavefront -- the path of every read before
// round 4.  The event chain treats a timeout the same way (event_kernels.hip, chain_segment).
// long_fault (tests only, sgk_stat_options_t::debug_fault): 1 | part << 8 | phase << 16: workgroup `part` never arrives
// at barrier `phase` (1-based) and the spin bound is 2^12; 2 | bound << 8: that spin bound, nobody withheld.
__device__ inline bool lc_barrier(LcCtx &cx) {
    __shared__ uint32_t s_fail;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++cx.phase;
    if (threadIdx.x == 0) {
        uint32_t fail = 0u, bound = 1u << 24;
        const uint32_t mode = cx.fault & 0xffu;
        if (mode == 1u) bound = 1u << 12;
        else if (mode == 2u) bound = (cx.fault >> 8) ? (cx.fault >> 8) : 1u;
        if (mode == 1u && (uint32_t)cx.part == ((cx.fault >> 8) & 0xffu) && cx.phase == ((cx.fault >> 16) & 0xffu)) {
            lc_st(&cx.w->failed, 1u);  // (the withheld workgroup: it leaves, the others find out)
            fail = 1u;
        } else {
            atomicAdd(&cx.w->arrive, 1u);
            const uint32_t target = cx.phase * (uint32_t)LC_PARTS;
            uint32_t spins = 0u;
            while (lc_ld(&cx.w->arrive) < target) {
                if (lc_ld(&cx.w->failed)) { fail = 1u; break; }
                __builtin_amdgcn_s_sleep(2);
                if (++spins >= bound) {
                    lc_st(&cx.w->failed, 1u);
                    atomicAdd(&cx.hdr->n_timeout, 1u);
                    fail = 1u;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_fail = fail;
    }
    __syncthreads();
    return s_fail == 0u;  // (the next barrier's leading __syncthreads orders this read before the next write)
}

// level 1: the summary of one tile for the binade of the predicted accumulator mt; tsum: (about) the sum of its terms
template <typename TF>
__device__ __forceinline__ unsigned long long lc_summary(const TF &tf, double mt, bool force_mark, double &tsum) {
    const uint32_t mb = ss_bits(ss_uniform((float)mt));
    const uint32_t ex = (mb >> 23) & 0xffu;
    const bool ok = !(mb >> 31) && ex >= 27u && ex <= 227u && !force_mark;
    const uint32_t b0 = ((ok ? ex : 127u) << 23) | 0x400000u, b1 = b0 + 1u;
    SsWalk w = {ss_float(b0), ss_float(b1), 0u};
    ss_walk_terms<0, true>(w, tf);
    const uint32_t c0 = ss_bits(w.a0), c1 = ss_bits(w.a1);
    const uint32_t bad = (((c0 ^ b0) | (c1 ^ b1)) >> 23) | (w.neg >> 31);
    if (!ok || __any(bad != 0u)) {
        float v = 0.0f;
        ss_native_terms<0>(v, tf.with(ss_opaque_zero()));
        tsum = wave_sum_d((double)v);
        return 0ull;
    }
    const int f0 = (int)(c0 - b0), f1 = (int)(c1 - b1);
    int T0, T1;
    if (__any(f0 != f1)) {  // some lane met a tie: the tile's increment depends on the parity it is entered with
        const int s0 = __builtin_amdgcn_inverse_ballot_w64(ss_parity_in(f0, f1, 0)) ? f1 : f0;
        const int s1 = __builtin_amdgcn_inverse_ballot_w64(ss_parity_in(f0, f1, 1)) ? f1 : f0;
        T0 = wave_sum_i(s0);
        T1 = wave_sum_i(s1);
    } else T0 = T1 = wave_sum_i(f0);
    tsum = (double)T0 * (double)ss_float((ex - 23u) << 23);
    const uint32_t hi = (ex << 24) | LC_VALID | ((uint32_t)(T1 - T0 + 0x8000) & 0xffffu);
    return ((unsigned long long)hi << 32) | (uint32_t)T0;
}

// level 2: the accumulator m taken through tiles [0, nt) (records rec[0 .. nt)); eval(tile, m) evaluates one tile from
// the true accumulator.  One wave.
template <typename EVAL>
__device__ inline float lc_compose(float m, const unsigned long long *rec, int nt, uint32_t &n_true, EVAL eval) {
    const int lane = lane_id();
    for (int g0 = 0; g0 < nt; g0 += 64) {
        const int gn = nt - g0 < 64 ? nt - g0 : 64;
        const unsigned long long rc = lane < gn ? lc_ld(rec + g0 + lane) : 0ull;
        const uint32_t rhi = (uint32_t)(rc >> 32);
        const int t0 = (int)(uint32_t)rc, t1 = t0 + (int)(rhi & 0xffffu) - 0x8000;
        int skip = 0;
        while (skip < gn) {
            m = ss_uniform(m);
            const uint32_t mb = ss_bits(m);
            const uint32_t ex = (mb >> 23) & 0xffu;
            const bool live = lane >= skip && lane < gn;
            // (a record carries a binade in 27 .. 227 or is marked: a negative, tiny, huge or non-finite m matches none)
            const bool okl = live && !(mb >> 31) && (rhi & LC_VALID) && (rhi >> 24) == ex;
            const unsigned long long badm = __ballot(live && !okl);
            const int fb = badm ? (int)__builtin_amdgcn_readfirstlane(__ffsll((long long)badm) - 1) : gn;
            int fail = fb;
            if (fb > skip) {
                const int S = (int)((mb & 0x7fffffu) | 0x800000u);
                const bool in = lane >= skip && lane < fb;
                const int f0 = in ? t0 : 0, f1 = in ? t1 : 0;  // (other lanes: the identity map)
                int f = f0;
                if (__any(f0 != f1)) f = __builtin_amdgcn_inverse_ballot_w64(ss_parity_in(f0, f1, S)) ? f1 : f0;
                // a tile's increment is below 2^23 + 2^10, 64 of them overflow no int; the comparison is done in 64 bits
                const long long incl = (long long)wave_incl_scan_i(f);
                const unsigned long long cm = __ballot(in && (long long)S + incl > (1ll << 24));
                if (cm) fail = (int)__builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
                if (fail > skip) {
                    const int tot = __builtin_amdgcn_readlane((int)incl, fail - 1);
                    m = (float)(S + tot) * ss_float((ex - 23u) << 23);
                }
            }
            if (fail >= gn) break;
            m = eval(g0 + fail, m);
            ++n_true;
            skip = fail + 1;
        }
    }
    return m;
}

// One stage (one or two sums over the same tiles) of a long read.  SRC supplies the tiles:
//   NCH                      sums per stage
//   seek(t) / ahead(t, te) / next()   streaming: position at tile t; issue the loads of tile t + 1 (< te); step
//   terms(t, f)              calls f(Ix<c>, term functor of sum c) for the current tile, c = 0 .. NCH - 1; the functors
//                            mask what lies outside the region and carry the sum's orientation
//   eval(c, t, m)            sum c's tile t from the true (oriented) accumulator m
//   flip(c) / negated(c)     from now on sum c runs on the negated terms / does it?
//   pass_a(t) / pass_b(t) / end_b()   what else the subtool does with the current tile in either pass, and once per
//                            workgroup behind pass B (stat: pA output; window histogram)
// Returns the SIGNED sums in out[]; false: the read is declined (lc_barrier), out[] means nothing.  Every wave of the
// read's LC_PARTS workgroups calls it (barriers inside).
template <typename SRC>
__device__ inline bool lc_stage(SRC &src, LcCtx &cx, int ntiles, float (&out)[2], uint32_t &n_true_out) {
    constexpr int N = SRC::NCH;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const int gw = cx.part * LC_WG_WAVES + wv;  // this wave among the read's LC_WAVES
    const int per = (ntiles + LC_WAVES - 1) / LC_WAVES;
    const int ta = gw * per < ntiles ? gw * per : ntiles, te = ta + per < ntiles ? ta + per : ntiles;
    // ---- pass A: the sum of the terms of this wave's tiles (a double per lane; the prediction needs no more)
    double acc[N];
#pragma unroll
    for (int c = 0; c < N; ++c) acc[c] = 0.0;
    if (ta < te) {
        src.seek(ta);
        for (int t = ta; t < te; ++t) {
            src.ahead(t, te);
            src.terms(t, [&](auto ix, const auto &tf) {
                constexpr int c = decltype(ix)::v;
                float v = 0.0f;
                ss_native_terms<0>(v, tf);
                acc[c] += (double)v;
            });
            src.pass_a(t);
            src.next();
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const double tot = wave_sum_d(acc[c]);
        if (lane == 0) lc_st(&cx.w->seg_tot[gw][c], (unsigned long long)__double_as_longlong(tot));
    }
    if (!lc_barrier(cx)) return false;
    // the sum is oriented by the sign of the read's total (as the wave kernels orient it by the sign of the
    // accumulator): non-negative terms are what the summaries cover
    double mt[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const double v = __longlong_as_double((long long)lc_ld(&cx.w->seg_tot[lane][c]));  // (LC_WAVES == 64 lanes)
        const double before = wave_sum_d(lane < gw ? v : 0.0), all = wave_sum_d(v);
        if (all < 0.0) { src.flip(c); mt[c] = -before; }
        else mt[c] = before;
    }
    // ---- pass B: the tiles' summaries
    if (ta < te) {
        src.seek(ta);
        for (int t = ta; t < te; ++t) {
            src.ahead(t, te);
            src.terms(t, [&](auto ix, const auto &tf) {
                constexpr int c = decltype(ix)::v;
                double ts;
                const unsigned long long rc = lc_summary(tf, mt[c], t == 0, ts);
                if (lane == 0) lc_st(cx.rec[c] + t, rc);
                mt[c] += ts;
            });
            src.pass_b(t);
            src.next();
        }
    }
    src.end_b();
    if (!lc_barrier(cx)) return false;
    // ---- level 2: wave c of the read's first workgroup composes sum c
    if (cx.part == 0 && wv < N) {
        uint32_t n_true = 0u;
        const float m = lc_compose(0.0f, cx.rec[wv], ntiles, n_true, [&](int t, float mm) { return src.eval(wv, t, mm); });
        if (lane == 0) {
            lc_st(reinterpret_cast<uint32_t *>(&cx.w->m[wv]), ss_bits(m));
            atomicAdd(&cx.w->n_true, n_true);
        }
    }
    if (!lc_barrier(cx)) return false;
#pragma unroll
    for (int c = 0; c < N; ++c) out[c] = ss_signed(ss_float(lc_ld(reinterpret_cast<const uint32_t *>(&cx.w->m[c]))), src.negated(c));
    n_true_out = lc_ld(&cx.w->n_true);
    return true;
}

// ---- the tile sources
struct SrcTiles {  // streaming of the raw tiles of a region
    WaveRead wr;
    WaveTile cur, nxt;
    __device__ __forceinline__ void seek(int t) { wr.load(cur, t); }
    __device__ __forceinline__ void ahead(int t, int te) { if (t + 1 < te) wr.load(nxt, t + 1); }
    __device__ __forceinline__ void next() { cur = nxt; }
    __device__ __forceinline__ void pass_a(int) {}
    __device__ __forceinline__ void pass_b(int) {}
    __device__ __forceinline__ void end_b() {}
    template <typename F>
    __device__ __forceinline__ void bases(int t, F f) const {  // f(TermBase) for the current tile
        const int q0 = lane_id() * SS_SPL;
        int q_lo, q_hi;
        wr.range(t, 0, q_lo, q_hi);
        if (wr.interior(t)) f(TermBase<true>{cur, q0, q_lo, q_hi, 0u});
        else f(TermBase<false>{cur, q0, q_lo, q_hi, 0u});
    }
};
struct SrcStatSums : SrcTiles {  // stat, stage 1: raw and pA (src/stat.h:17-33)
    static constexpr int NCH = 2;
    Scale sc;
    int sraw;   // -1: the raw chain runs negated
    float sg;   // the pA chain's orientation times the sign of the unit (as in k_stat_wave)
    float *pa_dst;  // fused stat + pa: the pA array at the region's base (or null)
    __device__ void init(const sgk_batch_t &b, const Region &g, const Scale &s, float *pa_out) {
        wr.init(b, g); sc = s; sraw = 0; sg = s.unit < 0.0f ? -1.0f : 1.0f;
        pa_dst = pa_out ? pa_out + wr.rb : nullptr;
    }
    __device__ __forceinline__ void pass_a(int t) { if (pa_dst) pa_write_tile(wr, t, sc, pa_dst); }
    __device__ __forceinline__ void flip(int c) { if (c == 0) sraw = ~sraw; else sg = -sg; }
    __device__ __forceinline__ bool negated(int c) const { return c == 0 ? sraw != 0 : sg < 0.0f; }
    template <typename F>
    __device__ __forceinline__ void terms(int t, F f) const {
        const Scale so = {sc.offf, sc.unit * sg};
        bases(t, [&](auto b) {
            f(Ix<0>{}, TermRaw<decltype(b)::interior>{b, sraw});
            f(Ix<1>{}, TermPa<decltype(b)::interior>{b, so});
        });
    }
    __device__ __attribute__((noinline)) float eval(int c, int t, float m) const {
        WaveTile x;
        wr.load(x, t);
        const Scale so = {sc.offf, sc.unit * sg};
        if (c == 0) ss_tile1<true>(m, wr, x, t, [&](auto b) { return TermRaw<decltype(b)::interior>{b, sraw}; });
        else ss_tile1<true>(m, wr, x, t, [&](auto b) { return TermPa<decltype(b)::interior>{b, so}; });
        return m;
    }
};
struct SrcStatDevs : SrcTiles {  // stat, stage 2: squared deviations (src/stat.h:36-54)
    static constexpr int NCH = 2;
    Scale sc;
    float mraw, mpa;
    int lo;            // window histogram: first raw value
    uint32_t *hist;    // this workgroup's (LDS, zeroed)
    uint32_t *ghist;   // the read's (workspace, zeroed): the workgroups add theirs
    __device__ __forceinline__ void pass_b(int t) {
        int q_lo, q_hi;
        wr.range(t, 0, q_lo, q_hi);
        if (wr.interior(t)) hist_tile<true>(cur, q_lo, q_hi, lo, hist);
        else hist_tile<false>(cur, q_lo, q_hi, lo, hist);
    }
    __device__ __forceinline__ void end_b() {
        __syncthreads();
        for (int i = (int)threadIdx.x; i < WH_BINS; i += LC_WG_WAVES * 64) {
            const uint32_t h = hist[i];
            if (h) atomicAdd(&ghist[i], h);
        }
    }
    __device__ __forceinline__ void flip(int) {}
    __device__ __forceinline__ bool negated(int) const { return false; }
    template <typename F>
    __device__ __forceinline__ void terms(int t, F f) const {
        bases(t, [&](auto b) {
            f(Ix<0>{}, TermDevRaw<decltype(b)::interior>{b, mraw});
            f(Ix<1>{}, TermDevPa<decltype(b)::interior>{b, sc, mpa});
        });
    }
    __device__ __attribute__((noinline)) float eval(int c, int t, float m) const {
        WaveTile x;
        wr.load(x, t);
        if (c == 0) ss_tile1<false>(m, wr, x, t, [&](auto b) { return TermDevRaw<decltype(b)::interior>{b, mraw}; });
        else ss_tile1<false>(m, wr, x, t, [&](auto b) { return TermDevPa<decltype(b)::interior>{b, sc, mpa}; });
        return m;
    }
};
template <bool DEV>
struct SrcClamp : SrcTiles {  // jnn: rm_outlier(raw), then its squared deviations (src/jnn.c:195-199)
    static constexpr int NCH = 1;
    float mean;
    __device__ __forceinline__ void flip(int) {}
    __device__ __forceinline__ bool negated(int) const { return false; }
    template <typename B>
    __device__ __forceinline__ auto term(B b) const {
        if constexpr (DEV) return TermDevClamp<B::interior>{b, mean};
        else return TermClamp<B::interior>{b};
    }
    template <typename F>
    __device__ __forceinline__ void terms(int t, F f) const {
        bases(t, [&](auto b) { f(Ix<0>{}, term(b)); });
    }
    __device__ __attribute__((noinline)) float eval(int, int t, float m) const {
        WaveTile x;
        wr.load(x, t);
        ss_tile1<false>(m, wr, x, t, [&](auto b) { return term(b); });
        return m;
    }
};
template <bool DEV>
struct SrcRoll {  // jnnv2: the rolling means of ADW clamped samples, then their squared deviations (src/jnn.c:106-124)
    static constexpr int NCH = 1;
    WaveRead wr;  // region: the windows' first samples
    float mean;
    int T0;
    WaveTile tr, ld, trn, ldn;
    __device__ __forceinline__ void flip(int) {}
    __device__ __forceinline__ bool negated(int) const { return false; }
    __device__ __forceinline__ float term(int v) const {
        if constexpr (DEV) { const float d = roll_mean(v) - mean; return d * d; }
        else return roll_mean(v);
    }
    __device__ __forceinline__ void seek(int t) {
        T0 = window_total(wr, t, t == 0 ? wr.skip : 0);
        roll_load(wr, tr, ld, t);
    }
    __device__ __forceinline__ void ahead(int t, int te) { if (t + 1 < te) roll_load(wr, trn, ldn, t + 1); }
    __device__ __forceinline__ void next() { tr = trn; ld = ldn; }
    __device__ __forceinline__ void pass_a(int) {}
    __device__ __forceinline__ void pass_b(int) {}
    __device__ __forceinline__ void end_b() {}
    __device__ __forceinline__ void totals(const WaveTile &a, const WaveTile &b, int t, int &T, int (&tot)[SS_SPL]) const {
        if (t == 0 && wr.skip > 0) roll_tile<true>(a, b, wr.skip, T, tot);
        else roll_tile<false>(a, b, 0, T, tot);
    }
    template <typename F>
    __device__ __forceinline__ void terms(int t, F f) {
        int tot[SS_SPL];
        totals(tr, ld, t, T0, tot);
        const int q0 = lane_id() * SS_SPL;
        int q_lo, q_hi;
        wr.range(t, 0, q_lo, q_hi);
        float x[SS_SPL];
#pragma unroll
        for (int e = 0; e < SS_SPL; ++e) x[e] = (q0 + e >= q_lo && q0 + e < q_hi) ? term(tot[e]) : 0.0f;
        f(Ix<0>{}, TermArr{x});
    }
    __device__ __attribute__((noinline)) float eval(int, int t, float m) const {
        int T = window_total(wr, t, t == 0 ? wr.skip : 0);
        WaveTile a, b;
        roll_load(wr, a, b, t);
        int tot[SS_SPL];
        totals(a, b, t, T, tot);
        roll_chain_tile(m, wr, t, tot, [&](int v) { return term(v); });
        return m;
    }
};

enum { LC_STAT = 0, LC_JNN = 1, LC_ADAPT = 2 };
// lists the reads of long_min samples or more (any order) and gives each its tile records
__global__ __launch_bounds__(256) void k_long_list(StatArgs a) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.b.n_reads) return;
    const uint32_t len = a.b.lengths[r];
    if (len < a.long_min) return;
    const uint32_t i = atomicAdd(&a.long_hdr->n_long, 1u);
    if (i >= LC_CAP) return;
    const uint32_t need = (len + 7u) / SS_TILE + 2u;  // tiles of the read from an 8-sample boundary in front of it
    const uint32_t off = atomicAdd(&a.long_hdr->pool_used, need);
    a.long_list[i] = r;
    LongSums *o = a.longs + i;
    o->read = r;
    o->valid = 0u;
    o->rec_off = off + need <= a.long_pool_tiles ? off : LC_NO_REC;  // (no room: the read runs on one wave as before)
    a.long_work[i].arrive = 0u;
    a.long_work[i].n_true = 0u;
    a.long_work[i].failed = 0u;
}
// A batch whose own threshold still lists more reads than two rounds of the long kernel's grid take is a batch of
// similar, long reads: one wave per read balances that by itself (1 000 reads of 500 000 samples: stat 1.4 ms) and
// the long kernel, built for a few outliers, does not (64 reads at a time).  The list is dropped.
__global__ void k_long_limit(LongHdr *hdr, uint32_t limit) {
    if (threadIdx.x == 0 && hdr->n_long > limit) hdr->n_long = 0u;
}
template <int KIND>
__global__ __launch_bounds__(LC_WG_WAVES * 64) void k_long_chains(StatArgs a, JnnP p, AdaptP ap) {
    __shared__ uint32_t hist[KIND == LC_STAT ? WH_BINS : 1];
#ifndef SGK_LC_NOPRIO
    // these waves are the batch's critical path and share their SIMDs with the wave kernel's: they issue first
    __builtin_amdgcn_s_setprio(3);
#endif
    const uint32_t nl = a.long_hdr->n_long, n_long = nl < LC_CAP ? nl : LC_CAP;
    const uint32_t groups = gridDim.x / LC_PARTS;
    for (uint32_t i = blockIdx.x / LC_PARTS; i < n_long; i += groups) {
        LongSums *o = a.longs + i;
        if (o->rec_off == LC_NO_REC) continue;
        const uint32_t r = a.long_list[i];
        const Region g = get_region(REG_WHOLE, a.b, nullptr, r);
        LcCtx cx;
        cx.w = a.long_work + i;
        cx.hdr = a.long_hdr;
        cx.rec[0] = a.long_pool + o->rec_off;
        cx.rec[1] = a.long_pool + a.long_pool_tiles + o->rec_off;
        cx.phase = 0u;
        cx.part = (int)(blockIdx.x % LC_PARTS);
        cx.fault = a.long_fault;
        float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
        uint32_t tiles = 0u, n_true = 0u, done = 1u;  // done: 1 the sums, 2 the subtool's whole output for this read
        if (KIND == LC_STAT) {
            const Scale sc = make_scale(a.b.digitisation[r], a.b.offset[r], a.b.range[r]);
            const float nf = (float)(int)g.len;
            uint32_t *ghist = a.long_hist + (size_t)i * WH_BINS;
            for (int b = (int)threadIdx.x; b < WH_BINS / LC_PARTS; b += LC_WG_WAVES * 64) lc_st(&ghist[cx.part * (WH_BINS / LC_PARTS) + b], 0u);
            for (int b = (int)threadIdx.x; b < WH_BINS; b += LC_WG_WAVES * 64) hist[b] = 0u;
            SrcStatSums src1;
            src1.init(a.b, g, sc, a.pa_out);
            if (!lc_stage(src1, cx, src1.wr.ntiles, s1, n_true)) continue;  // declined: the redo launch has the read
            SrcStatDevs src2;
            src2.wr = src1.wr; src2.sc = sc; src2.mraw = s1[0] / nf; src2.mpa = s1[1] / nf;
            src2.lo = hist_window_lo(src2.mraw); src2.hist = hist; src2.ghist = ghist;
            if (!lc_stage(src2, cx, src2.wr.ntiles, s2, n_true)) continue;
            tiles = 4u * (uint32_t)src1.wr.ntiles;
            // the record: the read's histogram through this workgroup's LDS (everybody is behind the stage's last barrier)
            if (cx.part == 0 && threadIdx.x < 64) {
                constexpr int PER = WH_BINS / 64;
                const int lane = lane_id();
#pragma unroll
                for (int b = 0; b < PER; ++b) hist[lane * PER + b] = lc_ld(&ghist[lane * PER + b]);
                stat_finish<REG_WHOLE>(a, r, g, sc, src2.lo, hist, src2.mraw, src2.mpa, sqrtf(s2[0] / nf), sqrtf(s2[1] / nf));
            }
            __syncthreads();
            done = 2u;
        } else if (KIND == LC_JNN) {
            // (fixed thresholds: launch_jnn does not come here; slots too small for the chunks: k_jnn_wave keeps the read)
            if (p.std_scale > 0.0f && jnn_long_cap(a, r, (g.start & 7) + g.len) >= 4u) {
                const float nf = (float)(int)g.len;
                SrcClamp<false> src1;
                src1.wr.init(a.b, g); src1.mean = 0.0f;
                if (!lc_stage(src1, cx, src1.wr.ntiles, s1, n_true)) continue;
                SrcClamp<true> src2;
                src2.wr = src1.wr; src2.mean = s1[0] / nf;
                if (!lc_stage(src2, cx, src2.wr.ntiles, s2, n_true)) continue;
                tiles = 2u * (uint32_t)src1.wr.ntiles;
                // ---- the automaton on all waves: 64 chunks per wave (jnn_chunks), every chunk stages its first
                // candidate and its strong segments in its part of the upper half of the read's slots; one wave merges
                // them, 64 chunks per round
                const float mn = s1[0] / nf, band = sqrtf(s2[0] / nf) * p.std_scale;
                const JnnThr th = jnn_thresholds(mn + band, mn - band, p);
                const WaveRead &wr = src1.wr;
                const int C = jnn_long_chunks(wr.skip + g.len);
                const uint64_t slot0 = a.seg_slots[r], cap = a.seg_slots[r + 1] - slot0;
                const uint32_t half = (uint32_t)(cap / 2), capL = (uint32_t)((cap - half) / (uint32_t)C);
                {
                    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
                    const int gw = cx.part * LC_WG_WAVES + wv;
                    int32_t *up_x = a.seg_x + slot0 + half, *up_y = a.seg_y + slot0 + half;
                    if (gw * 64 < C) {
                        const int gc = gw * 64 + lane;
                        uint32_t *sx_ = reinterpret_cast<uint32_t *>(up_x + (uint64_t)(gc < C ? gc : 0) * capL);
                        uint32_t *sy_ = reinterpret_cast<uint32_t *>(up_y + (uint64_t)(gc < C ? gc : 0) * capL);
                        int fx = 0, fy = 0, fstrong = 0, has_first = 0;
                        uint32_t cnt = 0u;
                        auto candidate = [&](int sx, int sy, int c) {
                            const int strong = c >= p.window ? 1 : 0;
                            if (!has_first) { has_first = 1; fx = sx; fy = sy; fstrong = strong; }
                            else if (strong) {
                                if (cnt < capL - 2u) { lc_st(sx_ + 2 + cnt, (uint32_t)sx); lc_st(sy_ + 2 + cnt, (uint32_t)sy); }
                                ++cnt;
                            }
                        };
                        jnn_chunks(wr, g.len, th.hi_r, th.lo_r, p.error, th.keep_min, candidate, C, gw * 64);
                        if (gc < C) {  // the chunk's header: its first candidate, how many strong segments follow
                            lc_st(sx_, (uint32_t)fx); lc_st(sx_ + 1, (uint32_t)fy);
                            lc_st(sy_, (uint32_t)(has_first | (fstrong << 1))); lc_st(sy_ + 1, cnt);
                        }
                    }
                    if (!lc_barrier(cx)) continue;
                    if (cx.part == 0 && wv == 0) {
                        JnnCarry cy = {false, false, false, 0, 0u};
                        for (int j = 0; j < C; j += 64) {
                            const int gc = j + lane;
                            const int32_t *sx_ = up_x + (uint64_t)(gc < C ? gc : 0) * capL, *sy_ = up_y + (uint64_t)(gc < C ? gc : 0) * capL;
                            int fx = 0, fy = 0, fl = 0;
                            uint32_t cnt = 0u;
                            if (gc < C) { fx = jnn_ld<true>(sx_); fy = jnn_ld<true>(sx_ + 1); fl = jnn_ld<true>(sy_); cnt = (uint32_t)jnn_ld<true>(sy_ + 1); }
                            jnn_merge_round<true>(cy, fl & 1, fx, fy, (fl >> 1) & 1, cnt, capL - 2u, sx_ + 2, sy_ + 2, p.seg_dist,
                                                  a.seg_x + slot0, a.seg_y + slot0, half);
                        }
                        const uint32_t total = jnn_merge_flush(cy, a.seg_y + slot0, half);
                        if (lane == 0) a.n_segs[r] = total;  // (JNN_REDO_MARK: the lane-per-read kernel takes the read)
                    }
                    done = 2u;
                }
            }
        } else {
            if (g.len > ADW) {  // (always: long_min is far above the window)
                const int64_t m = g.len - ADW;
                const float mf = (float)(int)m;
                SrcRoll<false> src1;
                src1.wr.init(a.b, Region{g.start, m}); src1.mean = 0.0f;
                if (!lc_stage(src1, cx, src1.wr.ntiles, s1, n_true)) continue;
                SrcRoll<true> src2;
                src2.wr = src1.wr; src2.mean = s1[0] / mf;
                if (!lc_stage(src2, cx, src2.wr.ntiles, s2, n_true)) continue;
                tiles = 2u * (uint32_t)src1.wr.ntiles;
                // thresholds, run finder (it stops at the first adaptor candidate) and the record: one wave
                if (cx.part == 0 && threadIdx.x < 64) {
                    sgk_prefix_rec_t *rec = a.prefix + r;
                    if (threadIdx.x == 0) adaptor_init_rec(rec, g.len);
                    adaptor_find(src1.wr, window_total(src1.wr, 0, src1.wr.skip), s1[0], s2[0], mf, ap, rec);
                }
                done = 2u;
            }
        }
        if (cx.part == 0 && threadIdx.x == 0) {
            o->s1[0] = s1[0]; o->s1[1] = s1[1];
            o->s2[0] = s2[0]; o->s2[1] = s2[1];
            o->valid = done;
            atomicAdd(&a.long_hdr->n_tiles, tiles);
            atomicAdd(&a.long_hdr->n_true, n_true);
        }
    }
}

// ---------------------------------------------------------------- dispatch order of the wave-per-read kernels
// A wave-per-read kernel cannot finish before its longest read has: reads are handed to the waves longest first
// (workgroups start in index order), by a counting sort of the read lengths into 128 buckets (4 per octave).
// (per-workgroup LDS histograms first: a batch of equal-length reads would otherwise send every atomic to one word)
__global__ __launch_bounds__(256) void k_order_count(const uint32_t *lengths, uint32_t n, uint32_t *hist) {
    __shared__ uint32_t h[128];
    if (threadIdx.x < 128) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) atomicAdd(&h[len_bucket(lengths[r])], 1u);
    __syncthreads();
    if (threadIdx.x < 128 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_order_scan(uint32_t *hist /* 128 counts -> cursors, longest bucket first */) {
    if (threadIdx.x == 0) {
        uint32_t acc = 0u;
        for (int b = 127; b >= 0; --b) { const uint32_t c = hist[b]; hist[b] = acc; acc += c; }
    }
}
__global__ __launch_bounds__(256) void k_order_fill(const uint32_t *lengths, uint32_t n, uint32_t *cursor, uint32_t *order) {
    __shared__ uint32_t h[128], base[128];
    if (threadIdx.x < 128) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    uint32_t b = 0u, local = 0u;
    if (r < n) { b = len_bucket(lengths[r]); local = atomicAdd(&h[b], 1u); }
    __syncthreads();
    if (threadIdx.x < 128 && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (r < n) order[base[b] + local] = r;
}
size_t order_workspace_bytes(uint32_t n_reads) { return 64 + ((size_t)n_reads * 4 + 128 * 4 + 63) / 64 * 64; }
int launch_order(const uint32_t *lengths, uint32_t nr, uint32_t *order, uint32_t *hist, hipStream_t st) {
    SGK_HIP_TRY(hipMemsetAsync(hist, 0, 128 * 4, st));
    hipLaunchKernelGGL(k_order_count, dim3((nr + 255) / 256), dim3(256), 0, st, lengths, nr, hist);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(64), 0, st, hist);
    hipLaunchKernelGGL(k_order_fill, dim3((nr + 255) / 256), dim3(256), 0, st, lengths, nr, hist, order);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}
// the tile records: 8 bytes per tile and sum for the long reads of the batch, at most LC_POOL_TILES tiles
static uint32_t long_pool_tiles(uint64_t n_samples, uint32_t max_read_len) {
    if (max_read_len < LC_LONG_MIN_FLOOR) return 0u;
    const uint64_t most = n_samples / SS_TILE + 2ull * (n_samples / LC_LONG_MIN_FLOOR < LC_CAP ? n_samples / LC_LONG_MIN_FLOOR : LC_CAP) + 2ull;
    return (uint32_t)(most < LC_POOL_TILES ? most : LC_POOL_TILES);
}
size_t long_workspace_bytes(uint64_t n_samples, uint32_t max_read_len) {
    return sizeof(LongHdr) + (size_t)LC_CAP * (4 + sizeof(LongSums) + sizeof(LongWork) + LC_HIST_BINS * 4) +
           (size_t)long_pool_tiles(n_samples, max_read_len) * 16;
}
uint32_t long_threshold(uint64_t n_samples, int32_t opt_long_min, LongRule rule) {
    uint64_t lm64 = opt_long_min > 0 ? (uint64_t)opt_long_min : n_samples / rule.div;
    if (opt_long_min <= 0) {
        uint64_t fl = rule.floor_div ? n_samples / rule.floor_div : rule.floor_lo;
        fl = fl < rule.floor_lo ? rule.floor_lo : (fl > LC_LONG_MIN ? LC_LONG_MIN : fl);
        if (lm64 < fl) lm64 = fl;
    }
    if (lm64 < LC_LONG_MIN_FLOOR) lm64 = LC_LONG_MIN_FLOOR;
    return lm64 > 0xffffffffull ? 0xffffffffu : (uint32_t)lm64;
}
int prepare_long(StatArgs &a, void *ws, size_t ws_bytes, int32_t opt_long_min, LongRule auto_div, hipStream_t st) {
    a.long_hdr = nullptr;
    a.long_list = nullptr;
    a.longs = nullptr;
    a.long_work = nullptr;
    a.long_pool = nullptr;
    a.long_hist = nullptr;
    a.long_pool_tiles = 0u;
    a.long_min = 0u;
    // By default a read is long when one wavefront would still be busy with it after the rest of the batch is done:
    // a wave takes 2 - 4 ns per sample, the full GPU ~1.3 ps, and the batch's longest reads are dispatched first, so a
    // read of more than n_samples / 2048 samples (jnn, whose wave is slower on a long read: / 3072) decides when the
    // kernel ends (and one of less than 131 072 - 262 144 samples, by the size of the batch, costs less than the long
    // path's barriers: stat_args.h, LongRule).  Measured on 20 000
    // log-normal reads (1 081 of 262 144 samples or more): with all of those on the long path stat takes 6.8 ms
    // instead of 3.9 -- the wave kernels balance them.
    const uint32_t lm = long_threshold(a.b.n_samples, opt_long_min, auto_div);
    const size_t off = order_workspace_bytes(a.b.n_reads);
    if (!ws || ws_bytes < off + long_workspace_bytes(0, 0) || (reinterpret_cast<uintptr_t>(ws) & 7u)) return SGK_OK;
    char *base = static_cast<char *>(ws) + off;
    const uint32_t pool = long_pool_tiles(a.b.n_samples, a.b.max_read_len);
    if (opt_long_min < 0 || a.b.max_read_len < lm || pool == 0u || ws_bytes < off + long_workspace_bytes(a.b.n_samples, a.b.max_read_len)) {
        SGK_HIP_TRY(hipMemsetAsync(base, 0, sizeof(LongHdr), st));  // no long read in this call: sgk_stat_long_status says so
        return SGK_OK;
    }
    a.long_hdr = reinterpret_cast<LongHdr *>(base);
    base += sizeof(LongHdr);
    a.longs = reinterpret_cast<LongSums *>(base);
    base += (size_t)LC_CAP * sizeof(LongSums);
    a.long_work = reinterpret_cast<LongWork *>(base);
    base += (size_t)LC_CAP * sizeof(LongWork);
    a.long_pool = reinterpret_cast<unsigned long long *>(base);
    base += (size_t)pool * 16;
    a.long_hist = reinterpret_cast<uint32_t *>(base);
    base += (size_t)LC_CAP * LC_HIST_BINS * 4;
    a.long_list = reinterpret_cast<uint32_t *>(base);
    a.long_pool_tiles = pool;
    a.long_min = lm;
    SGK_HIP_TRY(hipMemsetAsync(a.long_hdr, 0, sizeof(LongHdr), st));
    hipLaunchKernelGGL(k_long_list, dim3((a.b.n_reads + 255) / 256), dim3(256), 0, st, a);
    if (opt_long_min == 0) hipLaunchKernelGGL(k_long_limit, dim3(1), dim3(64), 0, st, a.long_hdr, LC_AUTO_MAX_READS);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}
// workgroups of k_long_chains: LC_PARTS per long read the batch can hold, all resident at once (at most 64 reads at a
// time: 1 024 workgroups of 256 threads; further long reads follow in the same workgroups)
static uint32_t long_grid(const StatArgs &a) {
    const uint64_t most = a.b.n_samples / a.long_min;
    return (uint32_t)(most < 1 ? 1 : (most > 64 ? 64 : most)) * LC_PARTS;
}
int prepare_order(StatArgs &a, void *ws, size_t ws_bytes, hipStream_t st) {
    a.order = nullptr;
    const uint32_t nr = a.b.n_reads;
    if (!ws || nr < ORDER_MIN_READS || ws_bytes < order_workspace_bytes(nr) || (reinterpret_cast<uintptr_t>(ws) & 3u)) return SGK_OK;
    // (a batch of near-equal lengths -- the longest read at most 1.25 x the mean -- is taken in batch order)
    if ((uint64_t)a.b.max_read_len * nr <= a.b.n_samples + a.b.n_samples / 4) return SGK_OK;
    uint32_t *order = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + 64), *hist = order + nr;
    const int rc = launch_order(a.b.lengths, nr, order, hist, st);
    if (rc != SGK_OK) return rc;
    a.order = order;
    return SGK_OK;
}

// ---------------------------------------------------------------- launchers
// Two implementations (sgk_stat_options_t::kernels): the lane-per-read kernels of round 1 (1; kept as an independent
// second implementation: tests compare the two, tools/bench_subtools.py times both) and the wave-per-read kernels (2).
// By default (0) a batch takes the wave kernels unless it is a LARGE batch of SHORT reads of SIMILAR length (stat:
// >= 49 152 reads of at most 32 768 samples or >= 16 384 of at most 16 384; jnn: >= 65 536 reads of at most 12 288; the
// longest at most 1.5 x the mean; prefix: see launch_prefix): there the lane-per-read kernels have 64 reads per wavefront, nothing to gain from intra-read parallelism and
// no per-read costs (native heads, binade crossings, chunk start-up), and are up to 2 x faster (400 000 x 5 000 samples:
// jnn 3.7 ms against 7.7 ms); everywhere else -- ragged, small or long-read batches -- the wave kernels win by 1.5 - 40 x.
// Per subtool (profiles/r04_z_subtools_wave_vs_lane_short_reads.txt, 2e9 samples per batch): stat's lane kernels win up to
// 32 768 samples per read (2.67 against 3.05 ms; at 65 536 the wave kernel wins), jnn's and prefix' only up to ~12 000
// (8 192: 3.5 / 7.0 against 5.4 / 7.6 ms; 16 384: 4.0 / 7.3 against 3.5 / 5.6).  stat's also win on smaller batches of
// short reads (20 000 x 5 000: 0.28 against 0.49 ms; 40 000 x 10 000: 0.61 against 1.13; 20 000 x 20 000: a tie), jnn's and
// prefix' need the 65 536 reads (40 000 x 5 000: 0.74 / 1.48 against 0.85 / 0.85).
bool stat_lane_per_read(int tool, int kernels, uint32_t n_reads, uint64_t n_samples, uint32_t max_read_len) {
    if (kernels == 1) return true;
    if (kernels == 2) return false;
    if (tool != 4 && (uint64_t)max_read_len * n_reads > n_samples + n_samples / 2) return false;  // not of similar length
    for (int k = 0; k < N_LANE_RULES; ++k) {
        const LaneRule &q = LANE_RULES[k];
        if (q.tool == tool && n_reads >= q.min_reads && max_read_len <= lane_rule_max_len(q, n_reads)) return true;
    }
    return false;   // (prefix' finders, tool 2: the wave kernels win at every shape)
}
static bool lane_per_read(int tool, const StatArgs &a) {
    return stat_lane_per_read(tool, a.kernels, a.b.n_reads, a.b.n_samples, a.b.max_read_len);
}

#define SGK_LAUNCH(name, kern, grid, block, ...)                                   \
    do {                                                                           \
        ProfScope ps_(name, st);                                                   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, __VA_ARGS__);     \
    } while (0)

// The wave kernel (wave_launch(stream)) beside k_long_chains<KIND>.  The long reads' few workgroups go to the caller's
// stream and the wave kernel to a side stream that first waits for the fork event: launched the other way round the long
// workgroups found every slot taken by the wave kernel's -- whose first workgroups hold the batch's longest reads -- and
// started 2 ms late.  The side stream joins when the returned guard goes out of scope (or at guard.join()).
// Behind the join the wave kernel is launched once more over the long list (wave_launch(stream, redo args): LC_CAP waves)
// for the reads k_long_chains declined -- a barrier of theirs timed out, lc_barrier; usually none, the launch costs a few
// microseconds: no read's result depends on the long path having worked.
template <int KIND, typename WL>
static int launch_beside_long(const StatArgs &a, const JnnP &p, const AdaptP &ap, const char *name, hipStream_t st,
                              WL wave_launch) {
    if (!a.longs) {
        wave_launch(st, a, (a.b.n_reads + 3) / 4);
        SGK_HIP_TRY(hipGetLastError());
        return SGK_OK;
    }
    {
        SideFork side;  // (joins at the end of this block)
        const bool forked = side.open(0, st);
        {
            ProfScope ps_(name, st);
            hipLaunchKernelGGL((k_long_chains<KIND>), dim3(long_grid(a)), dim3(LC_WG_WAVES * 64), 0, st, a, p, ap);
        }
        SGK_HIP_TRY(hipGetLastError());
        wave_launch(forked ? side.stream() : st, a, (a.b.n_reads + 3) / 4);
        SGK_HIP_TRY(hipGetLastError());
    }
    StatArgs redo = a;
    redo.long_redo = 1u;
    redo.order = nullptr;
    wave_launch(st, redo, LC_CAP / 4);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_stat(const StatArgs &a, hipStream_t st) {
    const uint32_t nr = a.b.n_reads;
    if (nr == 0) return SGK_OK;
    if (lane_per_read(a.pa_out ? 3 : 0, a)) {
        if (a.pa_out) {
            SGK_LAUNCH("k_moments", (k_moments<REG_WHOLE>), (nr + 63) / 64, 64, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median_pa", (k_median<REG_WHOLE, true>), nr, 256, a);
        } else if (nr >= STAT_MOMENTS_MEDIAN_MIN_READS) {
            // the medians come out of the moments' second pass; k_median only for the reads it flagged.  (With fewer reads
            // k_moments has too few wavefronts -- 64 reads each -- to hide what the counting adds, and k_median, a
            // workgroup per read, fills the GPU: 61 035 x 32 768 fused 2.70, apart 2.33 ms; 100 000 x 20 000 2.00 / 2.24.)
            SGK_LAUNCH("k_moments_median", (k_moments<REG_WHOLE, true>), (nr + 63) / 64, 64, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median_flagged", (k_median<REG_WHOLE, false, true>), nr, 256, a);
        } else {
            SGK_LAUNCH("k_moments", (k_moments<REG_WHOLE>), (nr + 63) / 64, 64, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median", (k_median<REG_WHOLE, false>), nr, 256, a);
        }
        SGK_HIP_TRY(hipGetLastError());
        return SGK_OK;
    }
    // the long reads' workgroups run beside the wave kernel (which skips those reads) when a side stream is to be had
    {
        const int rc = launch_beside_long<LC_STAT>(a, JnnP{}, AdaptP{}, "k_long_chains_stat", st, [&](hipStream_t st, const StatArgs &aw, uint32_t grid) {
            if (aw.pa_out) SGK_LAUNCH(aw.long_redo ? "k_stat_wave_pa_redo" : "k_stat_wave_pa", (k_stat_wave<REG_WHOLE, true>), grid, 256, aw);
            else SGK_LAUNCH(aw.long_redo ? "k_stat_wave_redo" : "k_stat_wave", (k_stat_wave<REG_WHOLE, false>), grid, 256, aw);
        });
        if (rc != SGK_OK) return rc;
    }
    SGK_LAUNCH("k_median_flagged", (k_median<REG_WHOLE, false, true>), nr, 256, a);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_jnn(const StatArgs &a, const JnnP &p, hipStream_t st) {
    const uint32_t nr = a.b.n_reads;
    if (nr == 0) return SGK_OK;
    SGK_HIP_TRY(hipMemsetAsync(a.err_count, 0, 4, st));
    const bool wave_ok = p.error >= 0 && p.error < p.corrector && p.error <= 31 && p.window >= 128;
    if (lane_per_read(1, a) || !wave_ok) SGK_LAUNCH("k_jnn", k_jnn, (nr + 63) / 64, 64, a, p);
    else {
        StatArgs aw = a;
        if (!(p.std_scale > 0.0f)) aw.longs = nullptr;  // (fixed thresholds: no sums, k_jnn_wave does every read)
        {
            const int rc = launch_beside_long<LC_JNN>(aw, p, AdaptP{}, "k_long_chains_jnn", st, [&](hipStream_t st, const StatArgs &ax, uint32_t grid) {
                SGK_LAUNCH(ax.long_redo ? "k_jnn_wave_redo" : "k_jnn_wave", k_jnn_wave, grid, 256, ax, p);
            });
            if (rc != SGK_OK) return rc;
        }
        StatArgs redo = a;
        redo.jnn_redo = 1u;  // the reads the wave kernel gave up on (none, usually: its wavefronts return at once)
        SGK_LAUNCH("k_jnn_redo", k_jnn, (nr + 63) / 64, 64, redo, p);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_adaptor(const StatArgs &a, const AdaptP &p, hipStream_t st) {
    const uint32_t nr = a.b.n_reads;
    if (nr == 0) return SGK_OK;
    if (lane_per_read(2, a)) SGK_LAUNCH("k_adaptor", k_adaptor, (nr + 63) / 64, 64, a, p);
    else {
        const int rc = launch_beside_long<LC_ADAPT>(a, JnnP{}, p, "k_long_chains_adapt", st, [&](hipStream_t st, const StatArgs &ax, uint32_t grid) {
            SGK_LAUNCH(ax.long_redo ? "k_adaptor_wave_redo" : "k_adaptor_wave", k_adaptor_wave, grid, 256, ax, p);
        });
        if (rc != SGK_OK) return rc;
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_stat_f32(const float *x, int n, float *out3, hipStream_t st) {
    SGK_LAUNCH("k_stat_f32", k_stat_f32, 1, 256, x, n, out3);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_jnn_f32(const float *x, int64_t n, const JnnP &p, int32_t *seg_x, int32_t *seg_y, uint32_t cap,
                   uint32_t *n_segs, hipStream_t st) {
    SGK_LAUNCH("k_jnn_f32", k_jnn_f32, 1, 64, x, n, p, seg_x, seg_y, cap, n_segs);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int launch_prefix(const StatArgs &a, int rna, int pore, hipStream_t st) {
    const uint32_t nr = a.b.n_reads;
    if (nr == 0) return SGK_OK;
    const uint32_t gw = (nr + 63) / 64;
    // The adaptor and polyA finders: one read per wavefront unless the caller forces the lane kernels (k_adaptor_wave 4.3
    // against k_adaptor 5.5 ms on 400 000 x 5 000, 13.4 against 29 on 125 000 x 100 000; k_polya_wave 0.2 against 1.2 - 5.9).
    // The statistics of the regions they find are a few thousand samples per read whatever the read's length: with enough
    // reads to fill the lanes (64 per wavefront) the lane kernels do them in 1.0 ms where k_stat_wave takes 2.4 (400 000 x
    // 5 000), 0.9 + 1.2 against 1.1 + 1.4 (50 000 x 100 000 RNA, adaptor + polyA), a tie at 125 000 x 100 000.
    const bool lanes = lane_per_read(2, a);
    const bool lane_regions = lanes || lane_per_read(4, a);
    if (lanes) SGK_LAUNCH("k_adaptor", k_adaptor, gw, 64, a, adaptor_preset(pore));
    else {
        // (joined inside: the kernels behind read every read's adapt_x / adapt_y)
        const AdaptP ap = adaptor_preset(pore);
        const int rc = launch_beside_long<LC_ADAPT>(a, JnnP{}, ap, "k_long_chains_adapt", st, [&](hipStream_t st, const StatArgs &ax, uint32_t grid) {
            SGK_LAUNCH(ax.long_redo ? "k_adaptor_wave_redo" : "k_adaptor_wave", k_adaptor_wave, grid, 256, ax, ap);
        });
        if (rc != SGK_OK) return rc;
    }
    SGK_HIP_TRY(hipGetLastError());
    if (lane_regions && !lanes) {  // (the medians out of the moments' second pass, k_median for the regions it flags)
        SGK_LAUNCH("k_moments_median_adapt", (k_moments<REG_ADAPT, true>), gw, 64, a);
        SGK_HIP_TRY(hipGetLastError());
        SGK_LAUNCH("k_median_adapt_flagged", (k_median<REG_ADAPT, false, true>), nr, 256, a);
    } else if (lane_regions) {
        SGK_LAUNCH("k_moments_adapt", (k_moments<REG_ADAPT>), gw, 64, a);
        SGK_HIP_TRY(hipGetLastError());
        SGK_LAUNCH("k_median_adapt", (k_median<REG_ADAPT>), nr, 256, a);
    } else {
        SGK_LAUNCH("k_stat_wave_adapt", (k_stat_wave<REG_ADAPT, false>), (nr + 3) / 4, 256, a);
        SGK_HIP_TRY(hipGetLastError());
        SGK_LAUNCH("k_median_adapt_flagged", (k_median<REG_ADAPT, false, true>), nr, 256, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    if (rna) {
        if (lanes) SGK_LAUNCH("k_polya", k_polya, gw, 64, a);
        else SGK_LAUNCH("k_polya_wave", k_polya_wave, (nr + 3) / 4, 256, a);
        SGK_HIP_TRY(hipGetLastError());
        if (lane_regions && !lanes) {
            SGK_LAUNCH("k_moments_median_polya", (k_moments<REG_POLYA, true>), gw, 64, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median_polya_flagged", (k_median<REG_POLYA, false, true>), nr, 256, a);
        } else if (lane_regions) {
            SGK_LAUNCH("k_moments_polya", (k_moments<REG_POLYA>), gw, 64, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median_polya", (k_median<REG_POLYA>), nr, 256, a);
        } else {
            SGK_LAUNCH("k_stat_wave_polya", (k_stat_wave<REG_POLYA, false>), (nr + 3) / 4, 256, a);
            SGK_HIP_TRY(hipGetLastError());
            SGK_LAUNCH("k_median_polya_flagged", (k_median<REG_POLYA, false, true>), nr, 256, a);
        }
        SGK_HIP_TRY(hipGetLastError());
    }
    return SGK_OK;
}

}  // namespace sgk
