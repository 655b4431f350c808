// misc_kernels.hip -- pA conversion (src/misc.c:15-32) and the synthetic read generator.
#include "sgk_common.h"
#include "synth.h"

namespace sgk {

// ---------------------------------------------------------------- pa
// One workgroup (256 threads) per 8192-sample slab of a read; 8 samples (16 B in, 32 B out) per
// lane per step: purely streaming, 6 B/sample.
#ifndef SGK_PA_SLAB
#define SGK_PA_SLAB 8192
#endif
#ifndef SGK_PA_UNROLL
#define SGK_PA_UNROLL 4
#endif
constexpr int PA_SLAB = SGK_PA_SLAB;
constexpr int PA_UNROLL = SGK_PA_UNROLL;

// A launch holds at most SLAB_GRID_MAX workgroups (gridDim.x * 256 threads must stay below 2^32, and a ragged batch
// with one very long read has n_reads * slabs_per_read far beyond that): every kernel of this shape strides over the
// (read, slab) pairs.
constexpr uint32_t SLAB_GRID_MAX = 1u << 22;

__global__ __launch_bounds__(256) void k_pa(const int16_t *samples, const uint64_t *offsets,
                                            const uint32_t *lengths, const double *dig,
                                            const double *off, const double *rng, uint32_t n_reads,
                                            uint32_t slabs_per_read, float *out) {
    // slabs_per_read comes from the batch's MEAN read length (launch_pa): a read longer than that many slabs is
    // finished by its workgroups in further rounds, so that one very long read does not make the grid n_reads x its slabs
    const uint64_t total = (uint64_t)n_reads * slabs_per_read;
    for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {
      const uint32_t r = (uint32_t)(w / slabs_per_read);
      const uint64_t o0 = offsets[r];
      const uint64_t n = lengths[r];
      const Scale sc = make_scale(dig[r], off[r], rng[r]);
      for (uint64_t slab = w % slabs_per_read; slab * PA_SLAB < n; slab += slabs_per_read) {
        const uint64_t b = slab * PA_SLAB;
        const uint64_t e = (b + PA_SLAB < n) ? b + PA_SLAB : n;
        const int16_t *src = samples + o0;
        float *dst = out + o0;
        const bool vec = ((reinterpret_cast<uintptr_t>(src + b) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(dst + b) & 15u) == 0);
        if (vec) {
            // four 16-byte loads in flight per lane before the first store (the loads of a plain loop wait behind the
            // previous iteration's stores: the compiler cannot know that src and dst do not alias).  (Round 5: 8-byte
            // loads and one 16-byte store per lane, every store instruction a contiguous KB, and eight loads in flight
            // instead of four were tried at 125 000 x 100 000: 13.4 - 13.9 ms either way, 75 GB at 5.5 TB/s is what the
            // memory system gives this 1 : 2 mix of reads and writes.)
            constexpr int UN = PA_UNROLL;
            for (uint64_t p0 = b + (uint64_t)threadIdx.x * 8; p0 < e; p0 += (uint64_t)UN * 256 * 8) {
                uint4 q[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint64_t p = p0 + (uint64_t)u * 256 * 8;
                    q[u] = (p + 8 <= e) ? *reinterpret_cast<const uint4 *>(src + p) : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const uint64_t p = p0 + (uint64_t)u * 256 * 8;
                    if (p + 8 <= e) {
                        int16_t s[8];
                        __builtin_memcpy(s, &q[u], 16);
                        float4 a, c;
                        a.x = to_pa(s[0], sc); a.y = to_pa(s[1], sc); a.z = to_pa(s[2], sc); a.w = to_pa(s[3], sc);
                        c.x = to_pa(s[4], sc); c.y = to_pa(s[5], sc); c.z = to_pa(s[6], sc); c.w = to_pa(s[7], sc);
                        *reinterpret_cast<float4 *>(dst + p) = a;
                        *reinterpret_cast<float4 *>(dst + p + 4) = c;
                    } else if (p < e) {
                        for (uint64_t k = p; k < e; ++k) dst[k] = to_pa(src[k], sc);
                    }
                }
            }
        } else {
            for (uint64_t p = b + threadIdx.x; p < e; p += 256) dst[p] = to_pa(src[p], sc);
        }
      }
    }
}

int launch_pa(const sgk_batch_t *b, float *out, hipStream_t st) {
    if (b->n_reads == 0 || b->max_read_len == 0) return SGK_OK;
    // workgroups per read: what a read of 1.25 x the mean length needs, at most what the longest needs
    const uint64_t mean = b->n_samples / b->n_reads;
    const uint64_t want = (mean + mean / 4 + PA_SLAB - 1) / PA_SLAB + 1, most = (b->max_read_len + PA_SLAB - 1) / PA_SLAB;
    const uint32_t spr = (uint32_t)(want < most ? want : most);
    const uint64_t blocks = (uint64_t)b->n_reads * spr;
    const uint32_t grid = blocks < SLAB_GRID_MAX ? (uint32_t)blocks : SLAB_GRID_MAX;
    {
        ProfScope ps("k_pa", st);
        hipLaunchKernelGGL(k_pa, dim3(grid), dim3(256), 0, st, b->samples, b->offsets, b->lengths, b->digitisation,
                           b->offset, b->range, b->n_reads, spr, out);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

// ---------------------------------------------------------------- synthetic reads
__global__ __launch_bounds__(256) void k_synth(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths,
                                               double *dig, double *off,
                                               double *rng, uint32_t n_reads, uint32_t slabs_per_read,
                                               uint64_t first_read, uint64_t seed, int kind) {
    const uint64_t total = (uint64_t)n_reads * slabs_per_read;
    for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const uint32_t r = (uint32_t)(w / slabs_per_read);
        const uint32_t slab = (uint32_t)(w % slabs_per_read);
        const uint64_t o0 = offsets[r];
        const int64_t n = (int64_t)lengths[r];
        const sgk_synth_read_t R = sgk_synth_read_init(seed, first_read + r, n, kind);
        if (slab == 0 && threadIdx.x == 0) {
            dig[r] = SGK_SYNTH_DIGITISATION;
            off[r] = (double)R.offset;
            rng[r] = SGK_SYNTH_RANGE;
        }
        const int64_t b = (int64_t)slab * PA_SLAB;
        const int64_t e = (b + PA_SLAB < n) ? b + PA_SLAB : n;
        for (int64_t i = b + threadIdx.x; i < e; i += 256) samples[o0 + i] = sgk_synth_sample(R, i);
    }
}

int launch_synth(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, double *dig, double *off,
                 double *rng, uint32_t n_reads,
                 uint32_t max_read_len, uint64_t first_read, uint64_t seed, int kind, hipStream_t st) {
    if (n_reads == 0) return SGK_OK;
    const uint32_t spr = (max_read_len + PA_SLAB - 1) / PA_SLAB;
    const uint64_t blocks = (uint64_t)n_reads * (spr ? spr : 1);
    const uint32_t grid = blocks < SLAB_GRID_MAX ? (uint32_t)blocks : SLAB_GRID_MAX;
    hipLaunchKernelGGL(k_synth, dim3(grid), dim3(256), 0, st, samples, offsets, lengths, dig, off, rng, n_reads,
                       spr ? spr : 1, first_read, seed, kind);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

}  // namespace sgk
