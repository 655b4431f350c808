// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ for the access shapes of k_event
// (VERDICT r02 #6): the guide (MI355X_MICROARCH.md, HBM) calibrates only 16 B/lane coalesced streaming reads (reported
// at exactly 1/2).  Each kernel below reads a buffer of known size exactly once:
//   k_coalesced<16|32|64>  lane l of a wave reads bytes [B*l, B*l+B) of the wave's contiguous B*64-byte piece
//                          (64: the builder's tile loads; 16: the guide's case)
//   k_lane_stream<32>      every lane walks ITS OWN contiguous region, 32 bytes per step: a wave-instruction touches
//                          64 different lines, each line is consumed over 4 steps (the detector's sample prefetch)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib   (and a second pass
// with --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum); the program prints the true byte count of every kernel.
// tools/pmc_kernels.py summarises the counters; profiles/archive/r03_fetch_calibration.json keeps the factors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int B>
__global__ __launch_bounds__(256) void k_coalesced(const uint4 *buf, uint64_t n16, uint32_t *sink) {
    constexpr int V = B / 16;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t p = wave * 64 * V; p + 64 * V <= n16; p += nwaves * 64 * V) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint4 v = buf[p + lane * V + k];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int B>
__global__ __launch_bounds__(64) void k_lane_stream(const uint4 *buf, uint64_t n16, uint32_t *sink) {
    constexpr int V = B / 16;
    const uint64_t lanes = (uint64_t)gridDim.x * 64, me = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t per = n16 / lanes / V * V;   // 16-byte units per lane
    const uint4 *mine = buf + me * per;
    uint32_t acc = 0;
    for (uint64_t p = 0; p < per; p += V) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint4 v = mine[p + k];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const uint64_t bytes = 2ull << 30;   // 2 GiB: well past the 256 MiB Infinity Cache
    uint4 *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    const uint64_t n16 = bytes / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_coalesced<16>), dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
        hipLaunchKernelGGL((k_coalesced<32>), dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
        hipLaunchKernelGGL((k_coalesced<64>), dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
        // 3072 waves of 64 lanes, each lane its own region of bytes / 196608 (about 10.9 KB; k_event: 3.1 KB per lane)
        hipLaunchKernelGGL((k_lane_stream<32>), dim3(3072), dim3(64), 0, 0, buf, n16, sink);
        hipLaunchKernelGGL((k_lane_stream<16>), dim3(3072), dim3(64), 0, 0, buf, n16, sink);
    }
    CHECK(hipDeviceSynchronize());
    const uint64_t lanes = 3072ull * 64;
    printf("{\"buffer_bytes\": %llu, \"true_bytes\": {\"k_coalesced<16>\": %llu, \"k_coalesced<32>\": %llu, \"k_coalesced<64>\": %llu, "
           "\"k_lane_stream<32>\": %llu, \"k_lane_stream<16>\": %llu}}\n",
           (unsigned long long)bytes, (unsigned long long)(n16 / 64 * 64 * 16), (unsigned long long)(n16 / 128 * 128 * 16),
           (unsigned long long)(n16 / 256 * 256 * 16), (unsigned long long)((n16 / lanes / 2 * 2) * lanes * 16),
           (unsigned long long)((n16 / lanes) * lanes * 16));
    return 0;
}
