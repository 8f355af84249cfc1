// Issue-rate probe for gfx950: v_exp_f32 vs v_fma_f32 vs v_mfma_f32_32x32x16_f16, alone and mixed, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; prints instructions per SIMD per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
    h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
    f16v acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        }
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        }
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int r = 0; r < (MODE == 5 ? 2 : 4); ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        }
        if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* out) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3;
    printf("%-34s waves/SIMD %d : %.3f ms", name, waves_per_simd, ms);
    const double per_simd_iters = (double)iters * waves_per_simd;
    if (MODE == 0) printf("  exp/SIMD/us %.1f", per_simd_iters * 8 / us);
    if (MODE == 1) printf("  fma/SIMD/us %.1f", per_simd_iters * 8 / us);
    if (MODE == 2) printf("  mfma/SIMD/us %.1f", per_simd_iters * 2 / us);
    if (MODE == 3) printf("  exp/SIMD/us %.1f  mfma/SIMD/us %.1f", per_simd_iters * 8 / us, per_simd_iters * 2 / us);
    if (MODE == 4) printf("  exp/SIMD/us %.1f  fma/SIMD/us %.1f", per_simd_iters * 8 / us, per_simd_iters * 32 / us);
    if (MODE == 5) printf("  fma/SIMD/us %.1f  mfma/SIMD/us %.1f", per_simd_iters * 16 / us, per_simd_iters * 2 / us);
    if (MODE == 6) printf("  fma/SIMD/us %.1f  mfma/SIMD/us %.1f", per_simd_iters * 32 / us, per_simd_iters * 2 / us);
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("v_exp_f32 x8", w, out);
        run<1>("v_fma_f32 x8", w, out);
        run<2>("mfma 32x32x16 x2", w, out);
        run<3>("exp x8 + mfma x2 (32 exp-cyc vs 64)", w, out);
        run<4>("exp x8 + fma x32", w, out);
        run<5>("fma x16 + mfma x2 (64 vs 64 cyc)", w, out);
        run<6>("fma x32 + mfma x2 (128 vs 64 cyc)", w, out);
    }
    return 0;
}
