// A/B harness for the head_dim-64 flash-attention variants (csrc/attention_d64.hip), torch-free so it starts in
// milliseconds on a fresh GPU box: times ctrl_op_flash_attn through the C-ABI under every ctrl_attn_set_variant(v) at the
// shapes of the SDXL adapter's spatial self-attention (model/adapter_spatial_temporal.py:271: B8 h5 L16384, B8 h10 L4096,
// B8 h5 L4096), compares every variant with variant 0 (the round-2 kernel) element by element, and checks variant 0
// itself against a host double-precision softmax(QK^T)V on a sample of queries.
//
//   build:  hipcc -O2 -std=c++17 -Iinclude tools/attn_bench.cpp -o tools/bin/attn_bench -Lctrl-adapter_amd -lctrlhip \
//                 -Wl,-rpath,'$ORIGIN/../../ctrl-adapter_amd'
//   run:    tools/bin/attn_bench [out.txt [v1,v2,... [nshapes]]]   (on the GPU box, from the repo root; nshapes = 1: only the
//           L = 16384 shape, e.g. under rocprofv3 --pmc -- tools/attn_pmc.sh)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ctrl_hip.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

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

static float h2f(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024), (int)e - 25);
    return s ? -v : v;
}
static uint16_t f2h(float f) {      // round to nearest even, no denormal / overflow care beyond clamping (test data only)
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000;
    int e = (int)((x >> 23) & 255) - 127 + 15;
    uint32_t m = x & 0x7fffff;
    if (e <= 0) return (uint16_t)s;
    if (e >= 31) return (uint16_t)(s | 0x7bff);
    uint32_t r = (m >> 13) | ((uint32_t)e << 10);
    const uint32_t rem = m & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) ++r;
    return (uint16_t)(s | r);
}

// approximately normal fp16 values (sum of 4 uniforms), scaled
static void fill_normal(std::vector<uint16_t>& v, uint64_t seed, float scale) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = 0; i < v.size(); ++i) {
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            acc += (float)((s >> 11) & 0xffffff) / 16777216.0f - 0.5f;
        }
        v[i] = f2h(acc * 1.7320508f * scale);      // variance of the sum of 4 U(-.5,.5) = 1/3
    }
}

struct Shape { int B, heads, L; };

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    std::vector<int> variants;
    if (argc > 2) {
        for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok));
    } else {
        for (int v = 0; v <= 14; ++v) variants.push_back(v);
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const Shape all_shapes[] = {{8, 5, 16384}, {8, 10, 4096}, {8, 5, 4096}};
    const int nshapes = argc > 3 ? std::max(1, std::min(3, atoi(argv[3]))) : 3;
    const std::vector<Shape> shapes(all_shapes, all_shapes + nshapes);
    const float kscale = 1.4426950408889634f / 8.0f;     // softmax_scale * log2(e) for head_dim 64, folded into K
    for (const Shape& sh : shapes) {
        const int C = sh.heads * 64, L = sh.L, B = sh.B;
        const size_t nq = (size_t)B * L * C;
        // Q ~ N(0,1), K ~ N(0,1) * kscale: scores in the exp2 domain have a standard deviation of ~1.4, maxima around 6
        std::vector<uint16_t> hq(nq), hk(nq), hv(nq);
        fill_normal(hq, 1, 1.0f);
        fill_normal(hk, 2, kscale);
        fill_normal(hv, 3, 1.0f);       // V^T layout [B][C][L]
        void *dq, *dk, *dv, *d0, *d1;
        CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nq * 2)); CK(hipMalloc(&dv, nq * 2));
        CK(hipMalloc(&d0, nq * 2)); CK(hipMalloc(&d1, nq * 2));
        CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dk, hk.data(), nq * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dv, hv.data(), nq * 2, hipMemcpyHostToDevice));
        ctrl_attn_desc d = {};
        d.Q = dq; d.ldq = C; d.K = dk; d.ldk = C; d.Vt = dv; d.Lkpad = L; d.kvB = B; d.ldo = C;
        d.B = B; d.heads = sh.heads; d.D = 64; d.Lq = L; d.Lk = L; d.scale = 0.125f; d.k_prescaled = 1;
        const double flops = 4.0 * B * sh.heads * (double)L * L * 64;
        say("\nB%d h%d D64 L%d  (%.3f TFLOP per launch)\n", B, sh.heads, L, flops / 1e12);
        std::vector<uint16_t> ref(nq), got(nq);
        bool have_ref = false;
        for (int v : variants) {
            if (ctrl_attn_set_variant(v)) { say("  variant %2d: %s\n", v, ctrl_last_error()); continue; }
            d.O = (v == 0 || !have_ref) ? d0 : d1;
            CK(hipMemsetAsync(d.O, 0xff, nq * 2, st));
            if (ctrl_op_flash_attn(&d, st)) { say("  variant %2d: %s\n", v, ctrl_last_error()); continue; }
            if (hipStreamSynchronize(st) != hipSuccess) { say("  variant %2d: launch failed: %s\n", v, hipGetErrorString(hipGetLastError())); return 3; }
            for (int i = 0; i < 2; ++i) ctrl_op_flash_attn(&d, st);
            const int reps = L >= 16384 ? 6 : 20;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) ctrl_op_flash_attn(&d, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            if (!have_ref) {
                CK(hipMemcpy(ref.data(), d0, nq * 2, hipMemcpyDeviceToHost));
                have_ref = true;
                // host check of the reference variant on a sample of queries (double precision)
                double worst = 0.0, refmax = 0.0;
                const int samples[][3] = {{0, 0, 0}, {0, 0, 1}, {B - 1, sh.heads - 1, L - 1}, {B / 2, sh.heads / 2, L / 2 + 37}, {1, 1, 255}, {2, 0, 256}, {3, 2, 8191}};
                std::vector<double> sc(L);
                for (auto& sm : samples) {
                    const int b = sm[0], h = sm[1], q = sm[2] % L;
                    double mx = -1e300;
                    for (int k = 0; k < L; ++k) {
                        double s = 0;
                        for (int e = 0; e < 64; ++e)
                            s += (double)h2f(hq[((size_t)b * L + q) * C + h * 64 + e]) * (double)h2f(hk[((size_t)b * L + k) * C + h * 64 + e]);
                        sc[k] = s;
                        mx = std::max(mx, s);
                    }
                    double den = 0;
                    for (int k = 0; k < L; ++k) { sc[k] = exp2(sc[k] - mx); den += sc[k]; }
                    for (int e = 0; e < 64; ++e) {
                        double o = 0;
                        for (int k = 0; k < L; ++k) o += sc[k] * (double)h2f(hv[((size_t)b * C + h * 64 + e) * L + k]);
                        o /= den;
                        const double g = h2f(ref[((size_t)b * L + q) * C + h * 64 + e]);
                        worst = std::max(worst, fabs(g - o));
                        refmax = std::max(refmax, fabs(o));
                    }
                }
                say("  variant %2d vs host double softmax(QK^T)V on 7 sampled queries: max |err| %.3e (max |ref| %.3e)\n", v, worst, refmax);
                say("  variant %2d: %.4f ms  %7.1f TFLOP/s\n", v, ms, flops / (ms * 1e-3) / 1e12);
                continue;
            }
            CK(hipMemcpy(got.data(), d.O, nq * 2, hipMemcpyDeviceToHost));
            double maxd = 0.0, maxr = 0.0;
            size_t nbad = 0;
            for (size_t i = 0; i < nq; ++i) {
                const float a = h2f(got[i]), r = h2f(ref[i]);
                if (!(fabsf(a) < 1e30f)) { ++nbad; continue; }
                maxd = std::max(maxd, (double)fabsf(a - r));
                maxr = std::max(maxr, (double)fabsf(r));
            }
            say("  variant %2d: %.4f ms  %7.1f TFLOP/s   max |diff vs first variant| %.3e (max |ref| %.3e)%s\n", v, ms,
                flops / (ms * 1e-3) / 1e12, maxd, maxr, nbad ? "  NON-FINITE OUTPUTS" : "");
        }
        CK(hipFree(dq)); CK(hipFree(dk)); CK(hipFree(dv)); CK(hipFree(d0)); CK(hipFree(d1));
    }
    say("\ndone\n");
    return 0;
}
