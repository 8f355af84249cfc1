// Issue-mix probe for gfx950: the instruction mix of ONE wave-tile of the head_dim-64 flash-attention loop (csrc/attention_d64.hip,
// steady state of the deferred-rescale path: 16 v_mfma_f32_32x32x16_f16, 32 v_exp_f32, 16 v_cvt_pk_f16_f32, 15 v_max3_f32,
// 8 v_max_f32, 16 v_mov_b32, 16 ds_read_b128 -- counted in the device ISA) with NO data dependencies between the instructions, no
// barrier and no global loads: what the SIMD can issue when nothing but the issue port and the two pipes limits it.  Every
// instruction is its own `asm volatile`, so the order below IS the issue order of a wave.
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/attn_mix tools/ubench/attn_mix.hip
//   run:    tools/ubench/attn_mix [out.txt]        (prints time per wave-tile relative to the 16 MFMAs alone = attainable
//           fraction of the MFMA peak for this mix)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define CVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define MAX3(d, x, y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define MAX2(d, x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(x))
#define MOV(d, x) asm volatile("v_mov_b32 %0, %1" : "+v"(d) : "v"(x))
#define LDS(d, addr) asm volatile("ds_read_b128 %0, %1" : "+v"(d) : "v"(addr))
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)")

// MODE 0: the 16 MFMAs alone            1: the 99 VALU alone              2: interleaved evenly (6-7 VALU behind every MFMA)
//      3: = 2 + the 16 LDS reads        4: blocked like the product loop: [8 MFMA + max / mov] [8 MFMA + exp / cvt]
//      5: 32 exp + 16 MFMA interleaved   6: the 67 non-exp VALU + 16 MFMA interleaved
//      7: = 2 with only 16 exp (half of the exponentials gone)
//      8: = 3 with the VALU of each MFMA gap issued BEFORE the MFMA's LDS read (LDS in the MFMA shadow)
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(float* out, int iters) {
    __shared__ float4 lds[1024];
    lds[threadIdx.x] = float4{1.f, 2.f, 3.f, 4.f};
    lds[threadIdx.x + 512] = float4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    // (destinations are re-used round robin: 128 registers per wave, four waves per SIMD, like the product kernel)
    float x[16], y[8], m[8], t[4];      // x: exponentials in place; y: read-only inputs of everything else
    unsigned pk[4];
    f4v ld[2];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (float)(threadIdx.x + i);
    for (int i = 0; i < 8; ++i) { m[i] = 0.f; y[i] = 0.002f * (float)(threadIdx.x + i); }
    for (int i = 0; i < 4; ++i) { t[i] = 0.f; pk[i] = 0; }
    for (int i = 0; i < 2; ++i) ld[i] = f4v{0.f, 0.f, 0.f, 0.f};
    h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
    f16v acc[4] = {{0}, {0}, {0}, {0}};
    f4v acc16[8];
    for (int k = 0; k < 8; ++k) acc16[k] = f4v{0.f, 0.f, 0.f, 0.f};
    if (MODE == 9 || MODE == 11 || MODE == 13) {       // pseudo-random operands (what real activations look like to the multipliers)
        unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(h >> 16) & 4095) - 2048) * (_Float16)0.0007f;
            h = h * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(h >> 16) & 4095) - 2048) * (_Float16)0.0007f;
        }
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 16; ++i) { h = h * 1664525u + 1013904223u; acc[k][i] = (float)((int)(h >> 16) & 1023) * 0.001f; }
    }
    if (MODE == 10) { for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; } }
    const unsigned addr = (threadIdx.x & 511) * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 12 || MODE == 13) {      // the same FLOPs through the 16x16x32 instruction (the GEMM family's): 32 of them
#pragma unroll
            for (int k = 0; k < 32; ++k) MFMA16(acc16[k & 7]);
        } else if (MODE == 0 || MODE == 9 || MODE == 10) {
#pragma unroll
            for (int k = 0; k < 16; ++k) MFMA(acc[k & 3]);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                EXP(x[(2 * k) & 15]); EXP(x[(2 * k + 1) & 15]); CVT(pk[k & 3], y[((2 * k) & 15) & 7], y[((2 * k + 1) & 15) & 7]); MOV(t[k & 3], y[(k) & 7]);
                if (k < 15) MAX3(m[k & 7], y[(k) & 7], y[((k + 5) & 15) & 7]);
                if (k & 1) MAX2(m[(k >> 1) & 7], y[(k) & 7]);
            }
        } else if (MODE == 2 || MODE == 3 || MODE == 7 || MODE == 8 || MODE == 11) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (MODE == 3) { LDS(ld[k & 1], addr); }
                if (MODE != 8) MFMA(acc[k & 3]);
                EXP(x[(2 * k) & 15]);
                if (MODE != 7) EXP(x[(2 * k + 1) & 15]);
                CVT(pk[k & 3], y[((2 * k) & 15) & 7], y[((2 * k + 1) & 15) & 7]); MOV(t[k & 3], y[(k) & 7]);
                if (k < 15) MAX3(m[k & 7], y[(k) & 7], y[((k + 5) & 15) & 7]);
                if (k & 1) MAX2(m[(k >> 1) & 7], y[(k) & 7]);
                if (MODE == 8) { MFMA(acc[k & 3]); LDS(ld[k & 1], addr); }
                if ((MODE == 3 || MODE == 8) && (k & 3) == 3) LGKM0();
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                MFMA(acc[k & 3]);
                MOV(t[(2 * k) & 3], y[(k) & 7]); MOV(t[(2 * k + 1) & 3], y[((k + 8) & 15) & 7]);
                MAX3(m[k], y[(k) & 7], y[((k + 5) & 15) & 7]);
                if (k < 7) MAX3(m[k], y[((k + 8) & 15) & 7], y[((k + 9) & 15) & 7]);
                MAX2(m[k], y[((k + 1) & 15) & 7]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                MFMA(acc[k & 3]);
                EXP(x[(4 * k) & 15]); EXP(x[(4 * k + 1) & 15]); EXP(x[(4 * k + 2) & 15]); EXP(x[(4 * k + 3) & 15]);
                CVT(pk[(2 * k) & 3], y[((4 * k) & 15) & 7], y[((4 * k + 1) & 15) & 7]); CVT(pk[(2 * k + 1) & 3], y[((4 * k + 2) & 15) & 7], y[((4 * k + 3) & 15) & 7]);
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { MFMA(acc[k & 3]); EXP(x[(2 * k) & 15]); EXP(x[(2 * k + 1) & 15]); }
        } else if (MODE == 6) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                MFMA(acc[k & 3]);
                CVT(pk[k & 3], y[((2 * k) & 15) & 7], y[((2 * k + 1) & 15) & 7]); MOV(t[k & 3], y[(k) & 7]);
                if (k < 15) MAX3(m[k & 7], y[(k) & 7], y[((k + 5) & 15) & 7]);
                if (k & 1) MAX2(m[(k >> 1) & 7], y[(k) & 7]);
                if (k < 12) MOV(t[(k + 3) & 3], y[((k + 1) & 15) & 7]);
            }
        }
    }
    // keep everything live without arithmetic the optimiser could move into the loop
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(x[i]));
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(y[i]));
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(m[i]));
    for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(t[i])); asm volatile("" ::"v"(pk[i])); }
    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(ld[i]));
    for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(acc16[k]));
    float s = 0;
    for (int k = 0; k < 4; ++k) {
        asm volatile("" : "+v"(acc[k]));
        s += acc[k][0] + acc[k][15];
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static FILE* g_out = nullptr;
static int g_iters = 4000;
template <int MODE>
float run(const char* name, int wg_per_cu, float* out, float base) {
    const int iters = g_iters;
    const int blocks = 256 * wg_per_cu;      // 512 threads = 8 waves = 2 per SIMD; wg_per_cu workgroups per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 512>>>(out, 50);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        probe<MODE><<<blocks, 512>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 2 * wg_per_cu waves, each `iters` wave-tiles
    const double ns_per_tile = best * 1e6 / ((double)iters * 2 * wg_per_cu);
    char buf[256];
    snprintf(buf, sizeof buf, "%-64s waves/SIMD %d : %8.3f ms  %7.1f ns per wave-tile per SIMD%s", name, 2 * wg_per_cu, best, ns_per_tile,
             base > 0 ? "" : "\n");
    fputs(buf, stdout); if (g_out) fputs(buf, g_out);
    if (base > 0) {
        snprintf(buf, sizeof buf, "   MFMA-only time / this = %.3f\n", base / best);
        fputs(buf, stdout); if (g_out) fputs(buf, g_out);
    }
    return best;
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    float* out; hipMalloc(&out, (size_t)256 * 4 * 512 * sizeof(float));
    if (argc > 2) {      // data dependence of the matrix rate: long launches (clock / power management settles), 4 waves per SIMD
        g_iters = atoi(argv[2]);
        const float base = run<0>("0: 16 MFMA alone, operands = 1.0, accumulators from 0", 2, out, 0.f);
        run<10>("10: 16 MFMA alone, operands = 0", 2, out, base);
        run<9>("9: 16 MFMA alone, pseudo-random operands and accumulators", 2, out, base);
        run<2>("2: 16 MFMA + 99 VALU interleaved, operands = 1.0", 2, out, base);
        run<11>("11: 16 MFMA + 99 VALU interleaved, pseudo-random operands", 2, out, base);
        run<0>("0: again", 2, out, base);
        run<12>("12: 32 MFMA 16x16x32 alone (same FLOPs), operands = 1.0", 2, out, base);
        run<13>("13: 32 MFMA 16x16x32 alone, pseudo-random operands", 2, out, base);
        // how long a launch has to be for the clock to come down: the random-operand stream at 1/10 and 1/100 of the length
        for (int div : {10, 100}) {
            g_iters = atoi(argv[2]) / div;
            const float b2 = run<0>(div == 10 ? "0: operands = 1.0, launch 1/10 as long" : "0: operands = 1.0, launch 1/100 as long", 2, out, 0.f);
            run<9>(div == 10 ? "9: pseudo-random, launch 1/10 as long" : "9: pseudo-random, launch 1/100 as long", 2, out, b2);
        }
        if (g_out) fclose(g_out);
        return 0;
    }
    for (int w : {1, 2}) {
        const float base = run<0>("0: 16 MFMA 32x32x16 alone", w, out, 0.f);
        run<1>("1: 99 VALU alone (32 exp, 16 cvt_pk, 15 max3, 8 max, 16 mov ...)", w, out, base);
        run<2>("2: 16 MFMA + 99 VALU, interleaved evenly", w, out, base);
        run<3>("3: = 2 + 16 ds_read_b128 in front of the MFMAs", w, out, base);
        run<8>("8: = 3 with the LDS read behind the MFMA", w, out, base);
        run<4>("4: blocked like the product loop [8 MFMA+max/mov][8 MFMA+exp/cvt]", w, out, base);
        run<5>("5: 16 MFMA + 32 exp", w, out, base);
        run<6>("6: 16 MFMA + 67 non-exp VALU", w, out, base);
        run<7>("7: = 2 with 16 of the 32 exp removed", w, out, base);
    }
    if (g_out) fclose(g_out);
    return 0;
}
