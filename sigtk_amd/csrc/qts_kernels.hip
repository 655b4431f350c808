// qts_kernels.hip -- `sigtk qts` on the device: the per-sample quantisers of src/qts.c:27-43, :126-142
// and the svb-zd ENCODER that turns the quantised signal back into the blob a BLOW5 record stores
// (slow5lib/src/slow5_press.c:1063-1089: u32 count, streamvbyte keys + data of zigzag(delta), prev = 0).
//
// streamvbyte's encoding is canonical (every value takes the fewest bytes: code = number of bytes - 1), so the
// blobs produced here are byte-identical to the ones slow5lib writes; tests compare against the blobs of the
// reference's bundled file.
//
// One wavefront per read, tiles of 64 lanes x 16 samples, two passes:
//   k_svbzd_size    zigzag-delta byte lengths -> blob length of every read (the caller lays the blobs out)
//   k_svbzd_encode  key bytes straight to global memory (one dword per lane), data bytes staged in LDS at the
//                   lane's scanned byte offset and flushed as whole aligned dwords; the < 4 bytes left over are
//                   carried into the next tile.
#include "sgk_common.h"

namespace sgk {

constexpr int ENC_VPL = 16;
constexpr int ENC_TILE = 64 * ENC_VPL;
constexpr int ENC_STAGE = ENC_TILE * 4 + 8;  // worst-case data bytes of a tile + carried bytes (generic u32 values)

struct EncArgs {
    const int16_t *samples;
    const uint64_t *offsets;
    const uint32_t *lengths;
    uint32_t n_reads;
    uint8_t *blobs;               // encode only
    const uint64_t *blob_offsets; // encode only
    uint32_t *blob_lengths;       // size: out; encode: in
};

// this lane's 16 samples of the tile starting at v0 (zeros beyond the read) and the sample before them
__device__ __forceinline__ void enc_load(const int16_t *x, uint32_t n, uint32_t v0, int l, int (&s)[ENC_VPL], int &nval,
                                         int &prev_tile_last) {
    const uint32_t vl = v0 + (uint32_t)l * ENC_VPL;
    nval = vl >= n ? 0 : (n - vl >= (uint32_t)ENC_VPL ? ENC_VPL : (int)(n - vl));
    if (nval == ENC_VPL && ((reinterpret_cast<uintptr_t>(x + vl) & 15u) == 0)) {
        const uint4 *p = reinterpret_cast<const uint4 *>(x + vl);
        const uint4 a = p[0], b = p[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s[2 * k] = (int)(int16_t)(w[k] & 0xffffu);
            s[2 * k + 1] = (int)(int16_t)(w[k] >> 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < ENC_VPL; ++k) s[k] = (k < nval) ? (int)x[vl + k] : 0;
    }
    (void)prev_tile_last;
}

// zigzag(delta) of the lane's values and their byte lengths; returns the lane's data byte count
__device__ __forceinline__ int enc_codes(const int (&s)[ENC_VPL], int nval, int prev, uint32_t (&zz)[ENC_VPL], uint32_t &kw) {
    int nbytes = 0;
    kw = 0;
#pragma unroll
    for (int k = 0; k < ENC_VPL; ++k) {
        const int d = s[k] - prev;                                  // 17-bit for int16 input
        const uint32_t z = (uint32_t)((d + d) ^ (d >> 31));         // _zigzag_encode_32, streamvbyte_zigzag.c:11-13
        const uint32_t code = (z > 0xFFu) + (z > 0xFFFFu) + (z > 0xFFFFFFu);  // streamvbyte_encode.c:14-25
        zz[k] = z;
        if (k < nval) {
            kw |= code << (2 * k);
            nbytes += (int)code + 1;
        }
        prev = s[k];
    }
    return nbytes;
}

__global__ __launch_bounds__(64) void k_svbzd_size(EncArgs a) {
    const uint32_t r = blockIdx.x;
    const int l = lane_id();
    const uint32_t n = a.lengths[r];
    const int16_t *x = a.samples + a.offsets[r];
    uint32_t total = 0;
    int carry_prev = 0;  // last sample of the previous tile (0 before the read: prev = 0)
    for (uint32_t v0 = 0; v0 < n; v0 += ENC_TILE) {
        int s[ENC_VPL], nval, dummy = 0;
        enc_load(x, n, v0, l, s, nval, dummy);
        int up = __shfl_up(s[ENC_VPL - 1], 1, 64);
        const int prev = l == 0 ? carry_prev : up;
        uint32_t zz[ENC_VPL], kw;
        const int nb = enc_codes(s, nval, prev, zz, kw);
        total += (uint32_t)wave_last_i(wave_incl_scan_i(nb));
        carry_prev = __builtin_amdgcn_readlane(s[ENC_VPL - 1], 63);
    }
    if (l == 0) a.blob_lengths[r] = 4u + (n + 3u) / 4u + total;
}

__global__ __launch_bounds__(64) void k_svbzd_encode(EncArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[ENC_STAGE];
    const uint32_t r = blockIdx.x;
    const int l = lane_id();
    const uint32_t n = a.lengths[r];
    const int16_t *x = a.samples + a.offsets[r];
    uint8_t *blob = a.blobs + a.blob_offsets[r];
    const uint32_t nkeys = (n + 3u) / 4u;
    if (l < 4) blob[l] = (uint8_t)(n >> (8 * l));  // count word
    uint8_t *keys = blob + 4;
    uint8_t *data = keys + nkeys;
    // stage[0] always corresponds to a 4-byte aligned global address `gptr`; the first `carry` bytes of the stage
    // are already accounted for (tile 0: they belong to the key bytes in front of the data and are not stored)
    const int a0 = (int)(reinterpret_cast<uintptr_t>(data) & 3u);
    uint8_t *gptr = data - a0;
    int carry = a0;
    bool first = true;
    int carry_prev = 0;
    for (uint32_t v0 = 0; v0 < n; v0 += ENC_TILE) {
        int s[ENC_VPL], nval, dummy = 0;
        enc_load(x, n, v0, l, s, nval, dummy);
        const int up = __shfl_up(s[ENC_VPL - 1], 1, 64);
        const int prev = l == 0 ? carry_prev : up;
        uint32_t zz[ENC_VPL], kw;
        const int nb = enc_codes(s, nval, prev, zz, kw);
        carry_prev = __builtin_amdgcn_readlane(s[ENC_VPL - 1], 63);
        // key bytes of this lane: values v0 + 16 l .. -> key bytes (v0 + 16 l) / 4 .. + 3
        if (nval > 0) {
            uint8_t *kp = keys + (v0 / 4u) + (uint32_t)l * 4u;
            const int nk = (nval + 3) / 4;
            if (nk == 4 && (reinterpret_cast<uintptr_t>(kp) & 3u) == 0) *reinterpret_cast<uint32_t *>(kp) = kw;
            else
                for (int j = 0; j < nk; ++j) kp[j] = (uint8_t)(kw >> (8 * j));
        }
        const int incl = wave_incl_scan_i(nb);
        const int tile_bytes = wave_last_i(incl);
        __syncthreads();  // the previous flush is complete
        int pos = carry + incl - nb;
#pragma unroll
        for (int k = 0; k < ENC_VPL; ++k) {
            if (k < nval) {
                const uint32_t z = zz[k];
                const int len = (int)((kw >> (2 * k)) & 3u) + 1;
                stage[pos] = (uint8_t)z;
                if (len > 1) stage[pos + 1] = (uint8_t)(z >> 8);
                if (len > 2) stage[pos + 2] = (uint8_t)(z >> 16);
                if (len > 3) stage[pos + 3] = (uint8_t)(z >> 24);
                pos += len;
            }
        }
        __syncthreads();
        const int avail = carry + tile_bytes;
        const int ndw = avail >> 2;
        for (int w = l; w < ndw; w += 64) {
            if (first && w == 0 && a0 != 0) {
                for (int j = a0; j < 4; ++j) gptr[j] = stage[j];  // the leading bytes of this dword are key bytes
            } else {
                reinterpret_cast<uint32_t *>(gptr)[w] = reinterpret_cast<const uint32_t *>(stage)[w];
            }
        }
        __syncthreads();
        const int rem = avail & 3;
        uint32_t tailw = 0;
        if (l == 0 && rem) tailw = reinterpret_cast<const uint32_t *>(stage)[ndw];
        __syncthreads();
        if (l == 0 && rem) reinterpret_cast<uint32_t *>(stage)[0] = tailw;
        // (when ndw == 0 on the first tile the a0 foreign bytes are still in front: keep `first`)
        if (ndw > 0) first = false;
        gptr += (size_t)ndw * 4;
        carry = rem;
    }
    __syncthreads();
    // bytes that never filled a dword
    const int lo = (first ? a0 : 0);
    if (l >= lo && l < carry) gptr[l] = stage[l];
}

// ---------------------------------------------------------------- quantisers (src/qts.c:27-43, 126-142)
__device__ inline int16_t qts_apply(int16_t v, int b, int method) {
    const int x = (int)v;
    if (method == 0) return (int16_t)((x >> b) << b);                        // floor: (raw >> b) << b
    if (method == 2) return (int16_t)(x | ((1 << b) - 1));                    // fill-ones
    const int mask = (1 << b) - 1;                                            // round_to_power_of_2 (qts.c:27-43)
    const int lsb = x & mask;
    const int thr = 1 << (b - 1);
    const int base = x & ~mask;
    return (int16_t)(lsb < thr ? base : base + (1 << b));
}

__global__ __launch_bounds__(256) void k_qts(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths,
                                             uint32_t n_reads, uint32_t slabs_per_read, int bits, int method) {
    const uint64_t total = (uint64_t)n_reads * slabs_per_read;
    for (uint64_t w = blockIdx.x; w < total; w += gridDim.x) {  // bounded grid, see sgk_qts
        const uint32_t r = (uint32_t)(w / slabs_per_read);
        const uint32_t slab = (uint32_t)(w % slabs_per_read);
        const uint64_t n = lengths[r];
        const uint64_t b0 = (uint64_t)slab * 8192u;
        if (b0 >= n) continue;
        const uint64_t e = b0 + 8192u < n ? b0 + 8192u : n;
        int16_t *x = samples + offsets[r];
        for (uint64_t p = b0 + threadIdx.x; p < e; p += 256) x[p] = qts_apply(x[p], bits, method);
    }
}

}  // namespace sgk

using namespace sgk;

extern "C" {

int sgk_qts(int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads, uint32_t max_read_len,
            int bits, int method, void *stream) {
    if (n_reads == 0) return SGK_OK;
    if (!samples || !offsets || !lengths) return SGK_ERR_ARG;
    if (bits < 1 || bits > 15 || method < 0 || method > 2) return SGK_ERR_ARG;
    const uint32_t spr = (max_read_len + 8191u) / 8192u;
    if (spr == 0) return SGK_OK;
    const uint64_t blocks = (uint64_t)n_reads * spr;
    // gridDim.x * 256 threads must stay below 2^32: a bounded grid strides over the (read, slab) pairs
    const uint32_t grid = blocks < (1u << 22) ? (uint32_t)blocks : (1u << 22);
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("k_qts", st);
        hipLaunchKernelGGL(k_qts, dim3(grid), dim3(256), 0, st, samples, offsets, lengths, n_reads, spr, bits, method);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int sgk_svbzd_size(const int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                   uint32_t *blob_lengths, void *stream) {
    if (n_reads == 0) return SGK_OK;
    if (!samples || !offsets || !lengths || !blob_lengths) return SGK_ERR_ARG;
    EncArgs a = {samples, offsets, lengths, n_reads, nullptr, nullptr, blob_lengths};
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("k_svbzd_size", st);
        hipLaunchKernelGGL(k_svbzd_size, dim3(n_reads), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

int sgk_svbzd_encode(const int16_t *samples, const uint64_t *offsets, const uint32_t *lengths, uint32_t n_reads,
                     uint8_t *blobs, const uint64_t *blob_offsets, const uint32_t *blob_lengths, void *stream) {
    if (n_reads == 0) return SGK_OK;
    if (!samples || !offsets || !lengths || !blobs || !blob_offsets || !blob_lengths) return SGK_ERR_ARG;
    EncArgs a = {samples, offsets, lengths, n_reads, blobs, blob_offsets, const_cast<uint32_t *>(blob_lengths)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        ProfScope ps("k_svbzd_encode", st);
        hipLaunchKernelGGL(k_svbzd_encode, dim3(n_reads), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

}  // extern "C"
