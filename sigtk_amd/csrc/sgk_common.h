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

// Inclusive wave64 scans with DPP (row_shr 1/2/4/8 inside 16-lane rows, then row_bcast15 / row_bcast31
// across rows): register-to-register, no LDS crossbar round trips as with __shfl_up / ds_bpermute.
#define SGK_DPP_ROW_SHR(n) (0x110 + (n))
#define SGK_DPP_ROW_BCAST15 0x142
#define SGK_DPP_ROW_BCAST31 0x143
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_i(int v) {  // lanes without a source read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_d(double v) {  // lanes without a source read +0.0
    const int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ inline int wave_incl_scan_i(int v) {
    v += dpp_i<SGK_DPP_ROW_SHR(1), 0xf>(v);
    v += dpp_i<SGK_DPP_ROW_SHR(2), 0xf>(v);
    v += dpp_i<SGK_DPP_ROW_SHR(4), 0xf>(v);
    v += dpp_i<SGK_DPP_ROW_SHR(8), 0xf>(v);
    v += dpp_i<SGK_DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_i<SGK_DPP_ROW_BCAST31, 0xc>(v);
    return v;
}
// sum scan of a double; exact whenever every partial sum is representable (any order then agrees)
__device__ inline double wave_incl_scan_d(double v) {
    v = v + dpp_d<SGK_DPP_ROW_SHR(1), 0xf>(v);
    v = v + dpp_d<SGK_DPP_ROW_SHR(2), 0xf>(v);
    v = v + dpp_d<SGK_DPP_ROW_SHR(4), 0xf>(v);
    v = v + dpp_d<SGK_DPP_ROW_SHR(8), 0xf>(v);
    v = v + dpp_d<SGK_DPP_ROW_BCAST15, 0xa>(v);
    v = v + dpp_d<SGK_DPP_ROW_BCAST31, 0xc>(v);
    return v;
}
__device__ inline int wave_last_i(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ inline double wave_last_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// value of lane l-1 (wave shift right by one lane, DPP wave_shr:1); lane 0 receives `first`
__device__ inline int wave_shr1_i(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}
__device__ inline double wave_shr1_d(double v, double first) {
    const int lo = wave_shr1_i(__double2loint(v), __double2loint(first));
    const int hi = wave_shr1_i(__double2hiint(v), __double2hiint(first));
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_d(double v, int lane) {  // lane must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
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

// length class of a read for the dispatch order (launch_order: counting sort into 128 classes, 4 per octave)
__device__ inline uint32_t len_bucket(uint32_t n) {
    if (n < 4u) return n;
    const uint32_t e = 31u - (uint32_t)__clz((int)n);
    return 4u * e + ((n >> (e - 2u)) & 3u);  // <= 127
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
// two samples at once: the add and the multiply map onto v_pk_add_f32 / v_pk_mul_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 to_pa2(int16_t r0, int16_t r1, const Scale &s) {
    const f32x2 raw = {(float)r0, (float)r1};
    const f32x2 off = {s.offf, s.offf};
    const f32x2 unit = {s.unit, s.unit};
    const f32x2 shifted = raw + off;
    return shifted * unit;
}
__device__ inline f32x2 to_pa2(float p0, float p1, const Scale &) { return f32x2{p0, p1}; }

#endif  // __HIPCC__

}  // namespace sgk
