// row_stream.h -- LDS staging of 64 sample rows (one row per lane of a wavefront).
//
// A "row" is a contiguous run of samples in global memory (a chunk of a read, a whole read, or
// a region of a read).  The wave copies one 64-sample tile of every row per call: each lane
// moves one 16-byte vector, a row segment of 128 bytes (int16) is fetched by 8 adjacent lanes, so
// global traffic is whole cache lines even though consumption is lane-per-row.  In LDS the row
// stride is padded by one dword, which makes the lane-per-row reads bank-conflict free.
//
// All lanes of the wave must call load_tile() together (it contains workgroup barriers; the
// kernels using it run one wave per workgroup).
#pragma once
#include "sgk_common.h"

namespace sgk {

constexpr int TILE = 64;  // samples per row tile

template <typename T, int SLOTS>
struct RowStream {
    static constexpr int ROW_BYTES = SLOTS * TILE * (int)sizeof(T) + 4;
    static constexpr int LDS_BYTES = 64 * ROW_BYTES;
    static constexpr int PER_VEC = 16 / (int)sizeof(T);  // samples per 16-byte vector
    static constexpr int VECS = TILE / PER_VEC;          // vectors per row tile
    static constexpr int ROWS_PER_IT = 64 / VECS;

    char *lds;        // this wave's region (LDS_BYTES)
    const T *base;    // common base pointer (global)
    int64_t lo, hi;   // loads are legal for base-relative sample index in [lo, hi)
    int64_t rb;       // this lane's row: base-relative index of row sample 0
    bool base_al;     // base pointer 16-byte aligned

    __device__ __attribute__((noinline)) void load_tile(int tile, unsigned long long rowmask) {
        const int l = lane_id();
        const int v = l % VECS;
        __syncthreads();  // readers of the slot being replaced are done
        for (int it = 0; it < VECS; ++it) {
            const int row = it * ROWS_PER_IT + l / VECS;
            const int64_t rbr = (int64_t)__shfl((long long)rb, row, 64);
            if ((rowmask >> row) & 1ull) {
                const int64_t p0 = rbr + (int64_t)tile * TILE + (int64_t)v * PER_VEC;
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (base_al && (p0 & (PER_VEC - 1)) == 0 && p0 >= lo && p0 + PER_VEC <= hi) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(base + p0);
                    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
                } else {
                    T tmp[PER_VEC];
#pragma unroll
                    for (int k = 0; k < PER_VEC; ++k) {
                        const int64_t p = p0 + k;
                        tmp[k] = (p >= lo && p < hi) ? base[p] : (T)0;
                    }
                    __builtin_memcpy(w, tmp, 16);
                }
                uint32_t *dst = reinterpret_cast<uint32_t *>(
                    lds + row * ROW_BYTES + (tile % SLOTS) * TILE * (int)sizeof(T) + v * 16);
                dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
            }
        }
        __syncthreads();
    }
    // sample at row index q of this lane's row (its tile must be resident)
    __device__ T get(int64_t q) const {
        const int off = lane_id() * ROW_BYTES + ((int)((q >> 6) % SLOTS) * TILE + (int)(q & 63)) * (int)sizeof(T);
        return *reinterpret_cast<const T *>(lds + off);
    }
};

}  // namespace sgk

namespace sgk {

// Prefetching variant for int16 rows, used by the lane-per-read kernels (stat / jnn / prefix):
//   issue(t)   every lane starts its eight 16-byte global loads of tile t (no wait),
//   commit(t)  waits for them and writes them into the wave's LDS tile,
//   row(w)     copies this lane's 64-sample row of the resident tile from LDS into 32 registers.
// A sweep issues tile t+1, copies tile t into registers, consumes it (pure arithmetic, no memory op on
// the serial float chain), then commits t+1 over the same LDS tile: one tile per stream is resident
// (8.4 KB per wave), so LDS never limits the number of waves per CU.  Rows must start on a multiple of 8
// samples (callers align the row base down and skip the leading samples) and the buffer base must be
// 16-byte aligned.  The bases of the eight rows a lane fetches pieces of are kept as 32-bit offsets
// (units of 8 samples) from the wave's first row.
struct RowPrefetch {
    static constexpr int ROW_BYTES = TILE * 2 + 4;
    static constexpr int LDS_BYTES = 64 * ROW_BYTES;
    char *lds;
    const int16_t *base;
    int64_t hi;   // readable samples in [0, hi), hi a multiple of 8
    int64_t rb;   // this lane's row base (multiple of 8)
    int64_t rb0;  // wave-uniform: the smallest row base of the wave
    int shift8;   // added to every row base, in units of 8 samples (a second stream over the same rows)
    uint4 pf[8];
    uint32_t rb_of[8];  // (row base - rb0) / 8 of the eight rows this lane fetches pieces of
    uint32_t rb_max;    // wave-uniform: the largest of them over the wave
    int t_lo, t_hi;     // tiles t_lo..t_hi lie inside the buffer for every row of the wave: no clamping needed
    unsigned long long rowmask;

    __device__ void init(char *lds_, const int16_t *base_, int64_t hi_, int64_t rb_, unsigned long long rowmask_) {
        lds = lds_; base = base_; hi = hi_; rb = rb_; rowmask = rowmask_;
        const int l = lane_id();
        const bool wanted = (rowmask_ >> l) & 1ull;
        long long m = wanted ? (long long)rb_ : 0x7fffffffffffffffll;  // rows nobody wants do not move the base
        long long mx = wanted ? (long long)rb_ : -0x7fffffffffffffffll;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const long long o = __shfl_xor(m, d, 64), o2 = __shfl_xor(mx, d, 64);
            m = o < m ? o : m;
            mx = o2 > mx ? o2 : mx;
        }
        rb0 = (m == 0x7fffffffffffffffll) ? 0 : m;
        rb_max = (m == 0x7fffffffffffffffll) ? 0u : (uint32_t)((mx - rb0) >> 3);
        const uint32_t mine = wanted ? (uint32_t)((rb_ - rb0) >> 3) : 0u;
#pragma unroll
        for (int it = 0; it < 8; ++it) rb_of[it] = (uint32_t)__shfl((int)mine, it * 8 + l / 8, 64);
        set_shift(0);
    }
    // a second stream over the same rows, displaced by s8 * 8 samples
    __device__ void set_shift(int s8) {
        shift8 = s8;
        // piece position = rb0 + (rb_of + shift8) * 8 + tile * 64 + v * 8 must lie in [0, hi - 8] for v = 0..7
        const int64_t first = rb0 + (int64_t)shift8 * 8, last = rb0 + ((int64_t)rb_max + shift8) * 8 + 56;
        const int64_t lo64 = first >= 0 ? 0 : (-first + TILE - 1) / TILE;
        int64_t hi64 = (hi - 8 - last) >= 0 ? (hi - 8 - last) / TILE : -1;
        // the fast path addresses pieces with 32-bit byte offsets from rb0
        const int64_t lim = ((int64_t)1 << 31) / (TILE * 2) - (((int64_t)rb_max + 8) * 16) / (TILE * 2) - 2;
        if (hi64 > lim) hi64 = lim;
        t_lo = (int)(lo64 > 0x7fffffff ? 0x7fffffff : lo64);
        t_hi = (int)hi64;
    }
    __device__ __forceinline__ bool interior(int tile) const { return tile >= t_lo && tile <= t_hi; }
    __device__ __forceinline__ int64_t piece_pos(int it, int tile) const {
        return rb0 + (((int64_t)rb_of[it] + shift8) << 3) + (int64_t)tile * TILE + (lane_id() & 7) * 8;
    }
    __device__ __forceinline__ void issue(int tile) {
        if (interior(tile)) {  // wave-uniform: plain 32-bit offset arithmetic, no clamping
            const char *wave_base = reinterpret_cast<const char *>(base + rb0);
            const int toff = (tile * TILE + shift8 * 8 + (lane_id() & 7) * 8) * 2;
#pragma unroll
            for (int it = 0; it < 8; ++it)
                pf[it] = *reinterpret_cast<const uint4 *>(wave_base + (int)(rb_of[it] * 16u) + toff);
            return;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int64_t p0 = piece_pos(it, tile);
            p0 = p0 > hi - 8 ? hi - 8 : p0;
            p0 = p0 < 0 ? 0 : p0;
            pf[it] = *reinterpret_cast<const uint4 *>(base + p0);
        }
    }
    __device__ __forceinline__ void commit(int tile) {
        const int l = lane_id();
        const int v = l & 7;
        const bool inside = interior(tile);
        __syncthreads();  // every lane has copied the previous tile into registers
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + l / 8;
            uint4 q = pf[it];
            if (!inside) {
                const int64_t p0 = piece_pos(it, tile);
                if (p0 < 0 || p0 > hi - 8) q = make_uint4(0u, 0u, 0u, 0u);  // outside the buffer: zeros
            }
            uint32_t *dst = reinterpret_cast<uint32_t *>(lds + row * ROW_BYTES + v * 16);
            dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void row(uint32_t (&w)[32]) const {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lds + lane_id() * ROW_BYTES);
#pragma unroll
        for (int k = 0; k < 32; ++k) w[k] = src[k];
    }
    template <int K>
    static __device__ __forceinline__ int16_t sample(const uint32_t (&w)[32]) {
        return (K & 1) ? (int16_t)(w[K / 2] >> 16) : (int16_t)(w[K / 2] & 0xffffu);
    }
    // a part of the row (N samples, N/2 registers, part index h) for kernels that hold two streams at once
    template <int N>
    __device__ __forceinline__ void row_part(int h, uint32_t (&w)[N / 2]) const {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lds + lane_id() * ROW_BYTES) + h * (N / 2);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) w[k] = src[k];
    }
    template <int K, int NW>
    static __device__ __forceinline__ int16_t sample_part(const uint32_t (&w)[NW]) {
        return (K & 1) ? (int16_t)(w[K / 2] >> 16) : (int16_t)(w[K / 2] & 0xffffu);
    }
};

}  // namespace sgk
