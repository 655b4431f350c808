// svb_kernels.hip -- GPU decode of BLOW5 "svb-zd" signal blobs.
//
// Format (slow5lib/src/slow5_press.c:1091-1146, streamvbyte_decode.c:59-105, streamvbyte_zigzag.c:34-40):
//   u32 count | ceil(count/4) key bytes (2-bit length codes, value j of a key byte at bits 2*(j%4))
//   | data bytes (code c -> c+1 little-endian bytes) ; value = zigzag(delta), prev = 0 ; out = (int16)prefix sum.
// The reference decodes this serially on the CPU (byte cursor + running sum).  Here one wavefront
// decodes one read: per tile of 64 lanes x 16 values each lane derives its 16 byte lengths from its four
// key bytes, a wave scan of the per-lane byte counts gives every lane its data cursor, the tile's data
// bytes are staged through LDS (coalesced dword loads), values are extracted with byte-aligned funnel
// shifts, zigzag-decoded, and a second (lane-local + wave) scan of the deltas yields the samples, which
// each lane stores as 32 contiguous bytes.  Integer arithmetic only: bit-exact by construction.
#include "sgk_common.h"

namespace sgk {

constexpr int SVB_VPL = 16;                 // values per lane per tile
constexpr int SVB_TILE = 64 * SVB_VPL;      // values per tile
constexpr int SVB_STAGE = SVB_TILE * 4 + 16;  // worst-case data bytes of a tile (+ alignment lead-in)

struct SvbArgs {
    const uint8_t *blobs;
    const uint64_t *blob_offsets;  // n_reads
    const uint32_t *blob_lengths;  // n_reads (bytes, including the 4-byte count)
    int16_t *samples;
    const uint64_t *offsets;       // n_reads (sample index of each read in `samples`)
    const uint32_t *lengths;       // n_reads (expected sample counts)
    uint32_t *status;              // n_reads: 0 ok, 1 count mismatch, 2 truncated / inconsistent blob
    uint32_t n_reads;
};

__global__ __launch_bounds__(64) void k_svbzd_decode(SvbArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[SVB_STAGE];
    const uint32_t r = blockIdx.x;
    const int l = lane_id();
    const uint8_t *blob = a.blobs + a.blob_offsets[r];
    const uint32_t blen = a.blob_lengths[r];
    uint32_t count = 0;
    if (blen >= 4) count = (uint32_t)blob[0] | ((uint32_t)blob[1] << 8) | ((uint32_t)blob[2] << 16) | ((uint32_t)blob[3] << 24);
    const uint32_t nkeys = (count + 3) / 4;
    uint32_t st = 0;
    if (blen < 4 || count != a.lengths[r]) st = 1;
    if ((uint64_t)4 + nkeys > blen) st = 2;
    if (st) {
        if (l == 0) a.status[r] = st;
        return;
    }
    const uint8_t *keys = blob + 4;
    const uint8_t *data = keys + nkeys;
    const uint32_t ndata = blen - 4 - nkeys;
    int16_t *out = a.samples + a.offsets[r];
    uint32_t cursor = 0;  // data bytes consumed so far (wave-uniform)
    int32_t prev = 0;     // running sum of deltas (wave-uniform)
    for (uint32_t v0 = 0; v0 < count; v0 += SVB_TILE) {
        const uint32_t vl = v0 + (uint32_t)l * SVB_VPL;  // first value of this lane
        // four key bytes = 16 two-bit codes of this lane (values beyond count have code 0 and are ignored)
        uint32_t kw = 0;
        if (vl < count) {
            const uint32_t kb = vl / 4;  // key byte index (vl is a multiple of 16)
            if (kb + 4 <= nkeys && (reinterpret_cast<uintptr_t>(keys + kb) & 3u) == 0) {
                kw = *reinterpret_cast<const uint32_t *>(keys + kb);  // the lane's four key bytes in one load
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kb + j < nkeys) kw |= (uint32_t)keys[kb + j] << (8 * j);
            }
        }
        const int nval = vl >= count ? 0 : (count - vl >= SVB_VPL ? SVB_VPL : (int)(count - vl));
        // byte count of this lane: nval + sum of its codes
        uint32_t kmask = kw;
        if (nval < SVB_VPL) kmask &= (nval == 0) ? 0u : ((1u << (2 * nval)) - 1u);
        const uint32_t codesum = __popc(kmask & 0x55555555u) + 2u * __popc(kmask & 0xAAAAAAAAu);
        const int nbytes = nval + (int)codesum;
        const int incl = wave_incl_scan_i(nbytes);
        const int lane_off = incl - nbytes;
        const int tile_bytes = wave_last_i(incl);
        if ((uint64_t)cursor + (uint32_t)tile_bytes > ndata) {  // blob shorter than its keys promise
            if (l == 0) a.status[r] = 2;
            return;
        }
        // stage the tile's data bytes in LDS with aligned dword loads (the byte range is contiguous;
        // `shift` bytes of lead-in keep the global loads 4-byte aligned)
        const uint8_t *src = data + cursor;
        const int shift = (int)(reinterpret_cast<uintptr_t>(src) & 3u);
        const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src - shift);
        const int ndw = (tile_bytes + shift + 3) >> 2;
        __syncthreads();
        for (int b = l; b < ndw; b += 64) reinterpret_cast<uint32_t *>(stage)[b] = src32[b];
        __syncthreads();
        // extract, zigzag-decode, lane-local prefix sum
        int32_t d[SVB_VPL];
        int pos = lane_off + shift;
        int32_t run = 0;
#pragma unroll
        for (int k = 0; k < SVB_VPL; ++k) {
            const int len = (int)((kw >> (2 * k)) & 3u) + 1;
            uint32_t v = 0;
            if (k < nval) {
                // the 4 bytes at stage[pos] through two aligned dword reads and a byte-granular funnel shift
                const uint32_t *sp = reinterpret_cast<const uint32_t *>(stage) + (pos >> 2);
                const uint32_t w = __builtin_amdgcn_alignbyte(sp[1], sp[0], (uint32_t)(pos & 3));
                v = w & (0xffffffffu >> (32 - 8 * len));
                pos += len;
            }
            const int32_t delta = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);
            run += delta;  // int32 wrap-around, as the reference's `prev += val`
            d[k] = run;
        }
        const int incl2 = wave_incl_scan_i(run);
        const int32_t base = prev + (incl2 - run);
        // store 16 samples (32 bytes) per lane
        if (nval == SVB_VPL && ((reinterpret_cast<uintptr_t>(out + vl) & 15u) == 0)) {
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                w[k] = ((uint32_t)(uint16_t)(int16_t)(base + d[2 * k])) | ((uint32_t)(uint16_t)(int16_t)(base + d[2 * k + 1]) << 16);
            uint4 *dst = reinterpret_cast<uint4 *>(out + vl);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        } else {
#pragma unroll
            for (int k = 0; k < SVB_VPL; ++k)
                if (k < nval) out[vl + k] = (int16_t)(base + d[k]);
        }
        prev += wave_last_i(incl2);
        cursor += (uint32_t)tile_bytes;
    }
    if (l == 0) a.status[r] = (cursor == ndata) ? 0u : 2u;
}

int launch_svbzd(const SvbArgs &a, hipStream_t st) {
    if (a.n_reads == 0) return SGK_OK;
    {
        ProfScope ps("k_svbzd_decode", st);
        hipLaunchKernelGGL(k_svbzd_decode, dim3(a.n_reads), dim3(64), 0, st, a);
    }
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

}  // namespace sgk

extern "C" int sgk_svbzd_decode(const uint8_t *blobs, const uint64_t *blob_offsets, const uint32_t *blob_lengths,
                                uint32_t n_reads, int16_t *samples, const uint64_t *offsets,
                                const uint32_t *lengths, uint32_t *status, void *stream) {
    if (n_reads == 0) return SGK_OK;
    if (!blobs || !blob_offsets || !blob_lengths || !samples || !offsets || !lengths || !status) return SGK_ERR_ARG;
    sgk::SvbArgs a = {blobs, blob_offsets, blob_lengths, samples, offsets, lengths, status, n_reads};
    return sgk::launch_svbzd(a, static_cast<hipStream_t>(stream));
}
