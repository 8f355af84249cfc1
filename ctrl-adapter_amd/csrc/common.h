// Common device/host helpers for libctrlhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

// dtype codes of the C ABI (include/ctrl_hip.h)
enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

// ---- error plumbing (no C++ exceptions cross the ABI) ----
void ctrl_set_error(const std::string& s);
#define CTRL_FAIL(msg)                                                            \
    do {                                                                          \
        ctrl_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
        return 1;                                                                 \
    } while (0)
#define CTRL_CHECK(cond, msg) do { if (!(cond)) CTRL_FAIL(msg); } while (0)
#define HIP_TRY(expr)                                                             \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) CTRL_FAIL(std::string(#expr) + " -> " + hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// ---- per-device singletons (one rank per GPU is the norm, but a process may drive several devices) ----
constexpr int kMaxDevices = 16;
inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < kMaxDevices) ? d : 0; }
// 4 KiB of zeros on the current device: source of the LDS-DMA loads that implement zero padding
const void* device_zero_page();
// range check of the fp16 activations (runtime.cpp): on?, and the per-device flag word the epilogues raise
bool range_check_on();
int* range_flag();

// ---- per-kernel-class event profiler (used by bench.py's roofline leg) ----
void prof_before(const char* tag, const char* kern_expr, long grid_threads, hipStream_t s);
void prof_after(hipStream_t s);
extern bool g_prof_on;
extern double g_prof_flops, g_prof_bytes;     // algorithmic work of the NEXT launch (set by the op, consumed by prof_before)
#define PROF_WORK(flops, bytes) do { if (g_prof_on) { g_prof_flops = (double)(flops); g_prof_bytes = (double)(bytes); } } while (0)
void prof_detail(const char* fmt, ...);       // optional per-launch shape note (dumped when CTRL_PROF_DUMP=<file> is set)
void prof_symbol(const char* fmt, ...);       // kernel symbol with its template arguments, as rocprofv3 prints it (default: the launch expression)

#define LAUNCH(tag, kern, grid, block, shmem, stream, ...)                        \
    do {                                                                          \
        if (g_prof_on) { const dim3 g_ = (grid), b_ = (block);                    \
            prof_before(tag, #kern, (long)g_.x * g_.y * g_.z * b_.x * b_.y * b_.z, stream); } \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);        \
        if (g_prof_on) prof_after(stream);                                        \
        hipError_t le_ = hipGetLastError();                                       \
        if (le_ != hipSuccess) CTRL_FAIL(std::string("launch ") + tag + ": " + hipGetErrorString(le_)); \
    } while (0)

#ifdef __HIPCC__
// ---- device helpers ----
__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    unsigned r = u + 0x7FFFu + ((u >> 16) & 1u);   // RNE (NaN not expected on this path)
    return (u16)(r >> 16);
}
__device__ __forceinline__ float load_as_f32(const void* p, size_t i, int dt) {
    if (dt == DT_F32) return ((const float*)p)[i];
    if (dt == DT_F16) return (float)((const half_t*)p)[i];
    return bf16_to_f32(((const u16*)p)[i]);
}
__device__ __forceinline__ void store_from_f32(void* p, size_t i, int dt, float v) {
    if (dt == DT_F32) ((float*)p)[i] = v;
    else if (dt == DT_F16) ((half_t*)p)[i] = (half_t)v;
    else ((u16*)p)[i] = f32_to_bf16(v);
}
// v_rcp_f32 (1 ulp) instead of an IEEE division (~10 VALU): SiLU sits in GEMM epilogues and the GroupNorm apply pass
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU with erf from Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below fp16 resolution).
//   gelu(x) = x/2 * (1 + erf(x/sqrt2)),  erf(z) = sign(z) * (1 - P(t) e^{-z^2}),  t = 1/(1 + p|z|)
//           = max(x, 0) - |x| * (P(t)/2) * e^{-x^2/2}            (x/2 + |x|/2 = max(x,0): no sign select, no 1 +/- erf)
// ~11 VALU issues instead of libm erff's ~50, which dominated the GEGLU GEMM epilogue
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
    const float hp = t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f +
                     t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
    return fmaf(-ax * hp, e, fmaxf(x, 0.f));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
