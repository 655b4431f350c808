// sgk_common.h -- internal helpers shared by the HIP translation units of libsigtk_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <float.h>

#include "../../include/sigtk_gpu.h"

namespace sgk {

// ---- host-side error plumbing -------------------------------------------------------
void set_hip_error(hipError_t e, const char *what, const char *file, int line);

#define SGK_HIP_TRY(call)                                                 \
    do {                                                                  \
        hipError_t _e = (call);                                           \
        if (_e != hipSuccess) {                                           \
            ::sgk::set_hip_error(_e, #call, __FILE__, __LINE__);          \
            return SGK_ERR_HIP;                                           \
        }                                                                 \
    } while (0)

// ---- per-kernel timing (sgk_profile_*) ----------------------------------------------
// RAII: records a hipEvent pair around one kernel launch when profiling is enabled.
struct ProfScope {
    ProfScope(const char *name, hipStream_t s);
    ~ProfScope();
    const char *name;
    hipStream_t stream;
    int slot;
};

static inline uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

// ---- device helpers -----------------------------------------------------------------
#if defined(__HIPCC__)

constexpr int WAVE = 64;

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

__device__ inline double shfl_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}
__device__ inline double shfl_up_d(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}

// inclusive wave scan (sum) of a double; exact whenever every partial sum is representable
__device__ inline double wave_incl_scan_d(double v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = shfl_up_d(v, d);
        if (l >= d) v = v + o;
    }
    return v;
}
__device__ inline int wave_incl_scan_i(int v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (l >= d) v += o;
    }
    return v;
}
__device__ inline float wave_min_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ inline float wave_max_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// pA conversion, src/misc.c:26-28: (float)raw + offset, then * unit; never contracted
// (the library is compiled with -ffp-contract=off).
struct Scale {
    float offf;
    float unit;
};
__device__ inline Scale make_scale(double digitisation, double offset, double range) {
    Scale s;
    const float rangef = (float)range;
    const float digf = (float)digitisation;
    s.offf = (float)offset;
    s.unit = rangef / digf;
    return s;
}
__device__ inline float to_pa(int16_t raw, const Scale &s) {
    const float shifted = (float)raw + s.offf;
    return shifted * s.unit;
}
__device__ inline float to_pa(float pa, const Scale &) { return pa; }

#endif  // __HIPCC__

}  // namespace sgk
