// rsq_check.hip -- exhaustive accuracy of v_rsq_f32 and v_rcp_f32 on this GPU: maximum relative error over every
// positive normal float, against 1/sqrt and 1/x evaluated in double.  sgk_tail_f32 (tstat_math.h) assumes a
// relative error of v_rsq_f32 of at most 2^-22; sgk_refined_rcp starts from v_rcp_f32 (1 ulp).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/rsq_check tools/rsq_check.hip ; run: ./tools/rsq_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void k(double *max_rsq, double *max_rcp, unsigned long long *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    double m1 = 0.0, m2 = 0.0;
    for (uint64_t u = 0x00800000ull + tid; u < 0x7f800000ull; u += stride) {
        const float x = __uint_as_float((uint32_t)u);
        const float r = __builtin_amdgcn_rsqf(x);
        const float c = __builtin_amdgcn_rcpf(x);
        const double er = fabs((double)r * sqrt((double)x) - 1.0);
        // the reciprocal is only used for event lengths (integers in [1, 2^24)): check it on [2^-64, 2^64)
        const double ec = (u >= 0x1f800000ull && u < 0x5f800000ull) ? fabs((double)c * (double)x - 1.0) : 0.0;
        m1 = er > m1 ? er : m1;
        m2 = ec > m2 ? ec : m2;
        if (er > 2.384185791015625e-07) atomicAdd(bad, 1ull);  // 2^-22
    }
    // reduce through global atomics on the bit patterns (non-negative doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long *>(max_rsq), (unsigned long long)__double_as_longlong(m1));
    atomicMax(reinterpret_cast<unsigned long long *>(max_rcp), (unsigned long long)__double_as_longlong(m2));
}

int main() {
    double *d; unsigned long long *b;
    hipMalloc(&d, 16); hipMalloc(&b, 8);
    hipMemset(d, 0, 16); hipMemset(b, 0, 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, d + 1, b);
    double h[2]; unsigned long long hb;
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(&hb, b, 8, hipMemcpyDeviceToHost);
    printf("v_rsq_f32: max relative error %.6e = 2^%.2f (%.3f ulp of 2^-24), inputs beyond 2^-22: %llu\n", h[0], log2(h[0]), h[0] / 5.9604644775390625e-08, hb);
    printf("v_rcp_f32: max relative error %.6e = 2^%.2f (%.3f ulp of 2^-24)\n", h[1], log2(h[1]), h[1] / 5.9604644775390625e-08);
    return (hb == 0 && h[0] < 2.384185791015625e-07) ? 0 : 1;
}
