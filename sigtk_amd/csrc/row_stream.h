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
