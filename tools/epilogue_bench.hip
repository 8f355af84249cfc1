// Epilogue micro-benchmark for the implicit GEMM's row outputs (round-3 groundwork, not part of the product build).
//
// DESIGN.md section 3 ("Where a short-K GEMM launch spends its time"): at K <= 512 a 256x256 tile spends more time handing
// its 128 KB of fp16 results to memory than multiplying, and the cost is latency (an LDS round trip and a drained vmcnt per
// 16-row slab), not bandwidth (the chip fills memory at 6.4 TB/s).  This program times the candidate schemes on the
// GEMM's own geometry -- 8 waves, 128x64 accumulators per wave in the swapped-operand fragment layout (a lane owns row
// lane & 15 and columns (lane >> 4) * 4 .. + 3 of every 16x16 fragment), a dummy MFMA k-loop of adjustable length in front
// -- and checks every scheme writes the same bytes:
//
//   slab      the product's scheme: per 16-row slab  ds_write x4 -> wait -> (ds_read x2, cvt, 16-B store) x2 -> wait
//   slab_nw   the same without the two explicit waits (LDS ops of one wave execute in order; the compiler still waits
//             for read DATA)
//   direct    no LDS: every lane stores its 4 halves (8 B) per fragment straight from registers, 16 rows x 32 B per store
//   tile16    the whole 128x64 wave tile converted to fp16 and staged at once (16 KB per wave = the dead 4-stage ring),
//             ONE wait, then 16 x (ds_read_b128 + 16-B store) back to back
//   tile16_nt tile16 with non-temporal stores
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/epilogue_bench.hip -o tools/bin/epilogue_bench
//   run:    tools/bin/epilogue_bench [out.txt]          (GPU box; ~2 s)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef half_t h8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

enum { V_SLAB = 0, V_SLAB_NW = 1, V_DIRECT = 2, V_TILE16 = 3, V_TILE16_NT = 4, NVAR = 5 };
static const char* kNames[NVAR] = {"slab", "slab_nw", "direct", "tile16", "tile16_nt"};

constexpr int BM = 256, BN = 256, WAVES_M = 2, WAVES_N = 4, WM = 128, WN = 64, MI = 8, NI = 4;

template <int V>
__global__ __launch_bounds__(512, 1) void epi_kernel(half_t* __restrict__ out, int M, int N, int kiters, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x - tile_m * ntn;
    const int row0 = tile_m * BM + wm * WM, col0 = tile_n * BN + wn * WN;
    const int erow = lane & 15, ecol = (lane >> 4) * 4;

    // accumulators: a deterministic function of the global (row, column) so every scheme must write identical bytes
    f4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row0 + mi * 16 + erow, c = col0 + ni * 16 + ecol + i;
                acc[mi][ni][i] = (float)((r * 31 + c * 17) & 1023) * (1.0f / 256.0f) - 2.0f;
            }
    // dummy k-loop: kiters x 32 MFMAs per wave on zero operands (keeps the values, occupies the matrix pipe like a k-loop)
    h8 za = {0, 0, 0, 0, 0, 0, 0, 0}, zb = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("" : "+v"(za), "+v"(zb));
    for (int k = 0; k < kiters; ++k) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zb, za, acc[mi][ni], 0, 0, 0);
    }

    if constexpr (V == V_DIRECT) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = row0 + mi * 16 + erow;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const f4 x = acc[mi][ni];
                const h4 p = {(half_t)x[0], (half_t)x[1], (half_t)x[2], (half_t)x[3]};
                if (row < M) *(h4*)(out + (size_t)row * N + col0 + ni * 16 + ecol) = p;
            }
        }
    } else if constexpr (V == V_SLAB || V == V_SLAB_NW) {
        constexpr int SLD = WN + 4;
        float* stg = (float*)smem + wave * (16 * SLD);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) *(f4*)(stg + erow * SLD + ni * 16 + ecol) = acc[mi][ni];
            if constexpr (V == V_SLAB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                const int row = row0 + mi * 16 + r;
                const f4 v0 = *(const f4*)(stg + r * SLD + c8 * 8), v1 = *(const f4*)(stg + r * SLD + c8 * 8 + 4);
                const h8 p = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3], (half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                if (row < M) *(h8*)(out + (size_t)row * N + col0 + c8 * 8) = p;
            }
            if constexpr (V == V_SLAB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        constexpr int SLH = WN + 8;                                   // halves per staged row (144 B: rows 4 banks apart)
        half_t* stg = (half_t*)smem + wave * (WM * SLH);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const f4 x = acc[mi][ni];
                const h4 p = {(half_t)x[0], (half_t)x[1], (half_t)x[2], (half_t)x[3]};
                *(h4*)(stg + (mi * 16 + erow) * SLH + ni * 16 + ecol) = p;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        h8 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = lane + 64 * j, r = idx >> 3, c8 = idx & 7;
            v[j] = *(const h8*)(stg + r * SLH + c8 * 8);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = lane + 64 * j, r = idx >> 3, c8 = idx & 7;
            const int row = row0 + r;
            if (row < M) {
                h8* dst = (h8*)(out + (size_t)row * N + col0 + c8 * 8);
                if constexpr (V == V_TILE16_NT) __builtin_nontemporal_store(v[j], dst);
                else *dst = v[j];
            }
        }
    }
}

static FILE* g_out = nullptr;
static void say(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fputs(buf, stdout);
    fflush(stdout);
    if (g_out) { fputs(buf, g_out); fflush(g_out); }
}

static uint64_t checksum(const void* d, size_t bytes) {
    std::vector<uint64_t> h(bytes / 8);
    CK(hipMemcpy(h.data(), d, bytes / 8 * 8, hipMemcpyDeviceToHost));
    uint64_t a = 0x243F6A8885A308D3ull;
    for (uint64_t x : h) a = (a ^ x) * 0x100000001B3ull + (a >> 29);
    return a;
}

template <int V>
static void launch(half_t* out, int M, int N, int kiters, hipStream_t st) {
    constexpr size_t smem = (V == V_TILE16 || V == V_TILE16_NT) ? (size_t)8 * WM * (WN + 8) * 2 : (size_t)8 * 16 * (WN + 4) * 4;
    static bool done = false;
    if (!done) {
        CK(hipFuncSetAttribute((const void*)epi_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        done = true;
    }
    const int ntm = (M + BM - 1) / BM, ntn = N / BN;
    hipLaunchKernelGGL(epi_kernel<V>, dim3(ntm * ntn), dim3(512), smem, st, out, M, N, kiters, ntn);
}

static void launch_v(int v, half_t* out, int M, int N, int kiters, hipStream_t st) {
    switch (v) {
        case V_SLAB: launch<V_SLAB>(out, M, N, kiters, st); break;
        case V_SLAB_NW: launch<V_SLAB_NW>(out, M, N, kiters, st); break;
        case V_DIRECT: launch<V_DIRECT>(out, M, N, kiters, st); break;
        case V_TILE16: launch<V_TILE16>(out, M, N, kiters, st); break;
        default: launch<V_TILE16_NT>(out, M, N, kiters, st); break;
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    const int M = 131072, N = 2048;
    const size_t bytes = (size_t)M * N * 2;
    half_t* out = nullptr;
    CK(hipMalloc((void**)&out, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    say("epilogue schemes, M %d N %d fp16 rows out (%zu MB), 256x256 tiles, 8 waves; kiters = dummy k-loop length (32 MFMAs per wave each)\n", M, N, bytes >> 20);
    uint64_t sums[NVAR];
    for (int v = 0; v < NVAR; ++v) {
        CK(hipMemsetAsync(out, 0xff, bytes, st));
        launch_v(v, out, M - 40, N, 0, st);                       // ragged M: the masked rows must stay 0xff in every scheme
        CK(hipStreamSynchronize(st));
        sums[v] = checksum(out, bytes);
    }
    for (int v = 0; v < NVAR; ++v) say("  %-10s checksum %016llx%s\n", kNames[v], (unsigned long long)sums[v], sums[v] == sums[0] ? "" : "   <-- DIFFERS");
    for (int kiters : {0, 8, 16, 32, 64}) {
        say("kiters %d\n", kiters);
        for (int round = 0; round < 2; ++round)
            for (int v = 0; v < NVAR; ++v) {
                std::vector<float> t;
                for (int i = 0; i < 2; ++i) launch_v(v, out, M, N, kiters, st);
                for (int i = 0; i < 7; ++i) {
                    CK(hipEventRecord(e0, st));
                    launch_v(v, out, M, N, kiters, st);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms);
                }
                std::sort(t.begin(), t.end());
                if (round == 1) say("  %-10s %.4f ms   %.2f TB/s of results\n", kNames[v], t[3], bytes / t[3] * 1e-9);
            }
    }
    say("\ndone\n");
    return 0;
}
