// Fused GEGLU feed-forward (csrc/ffn.hip) against the two-launch form (GEGLU GEMM + K = 2048 GEMM of csrc/igemm.hip) through the
// C-ABI, torch-free so it starts in milliseconds on a fresh GPU box: same packed weights, same inputs; prints the median launch
// times, the algorithmic TFLOP/s (2 M (512 x 4096 + 2048 x 512)) and the largest difference between the two results.
//
//   build:  hipcc -O2 -std=c++17 -Iinclude tools/ffn_bench.cpp -o tools/bin/ffn_bench -Lctrl-adapter_amd -lctrlhip -Wl,-rpath,'$ORIGIN/../../ctrl-adapter_amd'
//   run:    tools/bin/ffn_bench [out.txt]        (on the GPU box, from the repo root)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

static uint16_t f2h(float f) {      // round-to-nearest-even fp32 -> fp16 (normal range is all this tool needs)
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    int e = (int)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e <= 0) return (uint16_t)s;
    if (e >= 31) return (uint16_t)(s | 0x7c00u);
    uint32_t h = s | ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)h;
}
static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static float urand() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.0f * 2.f - 1.f; }
static void* dev_half(size_t n, float scale) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2h(urand() * scale);
    void* d = nullptr;
    CK(hipMalloc(&d, n * 2));
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static void* dev_float(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = urand() * scale;
    void* d = nullptr;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    // argv[2] (optional): value of CTRL_FF_FUSED for the fused launches -- "jitter", or an ablation build ("abl_nodma", "abl_nogeglu",
    // "abl_noread": WRONG results by construction; what each build does not take is what that part costs)
    if (argc > 2) { ctrl_policy_set("CTRL_FF_FUSED", argv[2]); }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int D = 512, H = 2048;
    // weights in the library's packs: W1 = GEGLU-interleaved [4096][512] (what the values are does not matter for timing / for the
    // comparison of the two forms: both read the same pack), W2 [512][2048], W2p = ctrl_op_ffn_pack_w2(W2)
    void* W1 = dev_half((size_t)2 * H * D, 0.06f);
    void* b1 = dev_float(2 * H, 0.1f);
    void* W2 = dev_half((size_t)D * H, 0.06f);
    void* b2 = dev_float(D, 0.1f);
    void* W2p = nullptr;
    CK(hipMalloc(&W2p, (size_t)D * H * 2));
    if (ctrl_op_ffn_pack_w2(W2, W2p, D, H, st) != 0) { say("pack failed: %s\n", ctrl_last_error()); return 3; }
    say("fused GEGLU feed-forward%s%s vs the two-launch form (fp32 residual stream in, fp32 master + fp16 mirror out)\n", argc > 2 ? " build " : "", argc > 2 ? argv[2] : "");
    for (int M : {131072, 32768, 8192, 131072 + 40}) {
        void* X = dev_half((size_t)M * D, 1.0f);
        void* res = dev_float((size_t)M * D, 1.0f);
        void *hid = nullptr, *o1 = nullptr, *o2 = nullptr, *m1 = nullptr, *m2 = nullptr;
        CK(hipMalloc(&hid, (size_t)M * H * 2));
        CK(hipMalloc(&o1, (size_t)M * D * 4));
        CK(hipMalloc(&o2, (size_t)M * D * 4));
        CK(hipMalloc(&m1, (size_t)M * D * 2));
        CK(hipMalloc(&m2, (size_t)M * D * 2));
        auto out_desc = [&](void* out, void* mir) {
            ctrl_igemm_desc d;
            memset(&d, 0, sizeof d);
            d.A = hid; d.lda = H; d.mode = 0; d.Cin = H; d.taps = 1;
            d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
            d.W = W2; d.M = M; d.Nout = D; d.Ktot = H; d.rows_per_img = 1; d.scale = 1.f;
            d.bias = (const float*)b2; d.res = res; d.ldres = D; d.res_f32 = 1;
            d.out16 = mir; d.ld16 = D;
            d.nseg = 1;
            d.seg[0].out = out; d.seg[0].ld = D; d.seg[0].ncols = D; d.seg[0].dtype = CTRL_F32; d.seg[0].L = 1;
            return d;
        };
        ctrl_igemm_desc g1;
        memset(&g1, 0, sizeof g1);
        g1.A = X; g1.lda = D; g1.mode = 0; g1.Cin = D; g1.taps = 1;
        g1.Hin = g1.Win = g1.Hout = g1.Wout = g1.stride = g1.up = 1;
        g1.W = W1; g1.M = M; g1.Nout = 2 * H; g1.Ktot = D; g1.rows_per_img = 1; g1.scale = 1.f; g1.geglu = 1;
        g1.bias = (const float*)b1;
        g1.nseg = 1;
        g1.seg[0].out = hid; g1.seg[0].ld = H; g1.seg[0].ncols = H; g1.seg[0].dtype = CTRL_F16; g1.seg[0].L = 1;
        ctrl_igemm_desc g2 = out_desc(o2, m2);
        ctrl_ffn_desc f;
        memset(&f, 0, sizeof f);
        f.X = X; f.ldx = D; f.W1 = W1; f.b1 = (const float*)b1; f.W2p = W2p; f.out = out_desc(o1, m1);
        auto run_fused = [&]() { return ctrl_op_ffn(&f, st); };
        auto run_two = [&]() { int rc = ctrl_op_igemm(&g1, st); return rc ? rc : ctrl_op_igemm(&g2, st); };
        if (run_fused() != 0) { say("M %d: fused launch failed: %s\n", M, ctrl_last_error()); return 3; }
        if (run_two() != 0) { say("M %d: two-launch form failed: %s\n", M, ctrl_last_error()); return 3; }
        CK(hipStreamSynchronize(st));
        // difference of the results
        std::vector<float> h1((size_t)M * D), h2((size_t)M * D);
        CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
        double dmax = 0, amax = 0;
        size_t bad = 0;
        for (size_t i = 0; i < h1.size(); ++i) {
            if (!(std::fabs(h1[i]) < 1e30f)) ++bad;
            dmax = std::max(dmax, (double)std::fabs(h1[i] - h2[i]));
            amax = std::max(amax, (double)std::fabs(h2[i]));
        }
        auto timeit = [&](auto&& fn) {
            std::vector<float> t;
            for (int i = 0; i < 3; ++i) fn();
            for (int i = 0; i < 9; ++i) {
                CK(hipEventRecord(e0, st));
                fn();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            return t[t.size() / 2];
        };
        // interleaved rounds (cdna_hip_programming.md rule 24)
        float tf = 1e9f, tt = 1e9f;
        for (int r = 0; r < 3; ++r) { tf = std::min(tf, timeit(run_fused)); tt = std::min(tt, timeit(run_two)); }
        const double fl = 2.0 * M * ((double)D * 2 * H + (double)H * D);
        say(" M %6d  fused %.4f ms (%.0f TFLOP/s)   two launches %.4f ms (%.0f TFLOP/s)   ratio %.3f   max |diff| %.3e of %.3e%s\n", M, tf,
            fl / tf * 1e-9, tt, fl / tt * 1e-9, tf / tt, dmax, amax, bad ? "   NON-FINITE VALUES IN THE FUSED RESULT" : "");
        hipFree(X); hipFree(res); hipFree(hid); hipFree(o1); hipFree(o2); hipFree(m1); hipFree(m2);
    }
    say("done\n");
    return 0;
}
