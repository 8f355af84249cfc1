// Tile-walk-order experiment for the implicit GEMM (csrc/tile_order.h), torch-free so it starts in milliseconds on a fresh
// GPU box: times ctrl_op_igemm through the C-ABI for the path's wide-N token GEMMs under each order and checks that the
// results are bit-identical between orders.
//
//   build:  hipcc -O2 -std=c++17 -Iinclude tools/gemm_order_bench.cpp -o tools/bin/gemm_order_bench \
//                 -Lctrl-adapter_amd -lctrlhip -Wl,-rpath,'$ORIGIN/../../ctrl-adapter_amd'
//   run:    tools/bin/gemm_order_bench [out.txt [ksweep|stores|check]]        (on the GPU box, from the repo root)
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

// random fp16 bit patterns with magnitudes in [2^-6, 1): sign | exponent 9..14 | mantissa
// (GEMM_BENCH_ZERO=1: all-zero operands instead -- the data dependence of the matrix rate, i.e. how far a kernel is power-limited)
static void fill_half(std::vector<uint16_t>& v, uint64_t seed) {
    if (const char* z = getenv("GEMM_BENCH_ZERO")) if (z[0] == '1') { std::fill(v.begin(), v.end(), (uint16_t)0); return; }
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = 0; i < v.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t r = (uint32_t)(s >> 20);
        v[i] = (uint16_t)(((r & 1) << 15) | ((9 + (r >> 1) % 6) << 10) | ((r >> 8) & 0x3ff));
    }
}

static void* dev_half(size_t n, uint64_t seed) {
    std::vector<uint16_t> h(n);
    fill_half(h, seed);
    void* d = nullptr;
    CK(hipMalloc(&d, n * 2));
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}

static uint64_t checksum(const void* d, size_t bytes) {
    std::vector<uint64_t> h(bytes / 8);
    CK(hipMemcpy(h.data(), d, bytes / 8 * 8, hipMemcpyDeviceToHost));
    uint64_t a = 0x243F6A8885A308D3ull;
    for (uint64_t x : h) a = (a ^ x) * 0x100000001B3ull + (a >> 29);
    return a;
}

struct Shape { const char* name; int M, N, K; bool geglu, stream; std::vector<const char*> orders; };

// K sweep: time(K) = fixed (prologue + epilogue + launch) + K * slope (main loop) for a plain fp16-out token GEMM, under the
// default 256x256 tile and the forced two-workgroup 256x128 tile, with real activations and with every row aliased to row 0
// (lda = 0: all A traffic hits in cache) -- separates "waiting for memory" from "issue / LDS / MFMA scheduling".
static int ksweep() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int M = 131072, N = 2048, KMAX = 4096;
    void* A = dev_half((size_t)M * KMAX, 1);
    void* W = dev_half((size_t)N * KMAX, 2);
    void* out = nullptr;
    CK(hipMalloc(&out, (size_t)M * N * 2));
    say("\nK sweep  M%d N%d, fp16 row output, no epilogue terms\n", M, N);
    const char* forces[] = {"", "256x128", "128x256"};
    for (const char* f : forces) {
        ctrl_policy_set("CTRL_IGEMM_FORCE", *f ? f : nullptr);      // (the library reads its environment once: overrides go through the policy table)
        for (int alias = 0; alias < 2; ++alias) {
            say(" tile %s, A rows %s\n", *f ? f : "default (256x256)", alias ? "aliased (lda = 0)" : "distinct");
            for (int K : {256, 512, 1024, 2048, 4096}) {
                ctrl_igemm_desc d;
                memset(&d, 0, sizeof d);
                d.A = A; d.lda = alias ? 0 : K; d.mode = 0; d.Cin = K; d.taps = 1;
                d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
                d.W = W; d.M = M; d.Nout = N; d.Ktot = K; d.rows_per_img = 1; d.scale = 1.f;
                d.nseg = 1;
                d.seg[0].out = out; d.seg[0].ld = N; d.seg[0].ncols = N; d.seg[0].dtype = CTRL_F16; d.seg[0].L = 1;
                std::vector<float> t;
                for (int i = 0; i < 2; ++i)
                    if (ctrl_op_igemm(&d, st) != 0) { say("  launch failed: %s\n", ctrl_last_error()); return 3; }
                for (int i = 0; i < 7; ++i) {
                    CK(hipEventRecord(e0, st));
                    ctrl_op_igemm(&d, st);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms);
                }
                std::sort(t.begin(), t.end());
                say("   K %4d  median %.4f ms  %.0f TFLOP/s\n", K, t[t.size() / 2], 2.0 * M * N * K / t[t.size() / 2] * 1e-9);
            }
        }
    }
    ctrl_policy_set("CTRL_IGEMM_FORCE", nullptr);
    say("\ndone\n");
    return 0;
}

// Where the fixed cost of a short-K launch goes: the same plain GEMM with the output rows aliased (ld = 0: every store hits
// one 4 KB row, nothing reaches HBM), the activation rows aliased (lda = 0), both, and the chip's plain fill / copy rates.
static int stores() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int M = 131072, N = 2048;
    void* A = dev_half((size_t)M * 512, 1);
    void* W = dev_half((size_t)N * 512, 2);
    void *out = nullptr, *out2 = nullptr;
    const size_t ob = (size_t)M * N * 2;
    CK(hipMalloc(&out, ob));
    CK(hipMalloc(&out2, ob));
    auto timeit = [&](auto&& fn) {
        std::vector<float> t;
        for (int i = 0; i < 2; ++i) fn();
        for (int i = 0; i < 7; ++i) {
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
    float ms = timeit([&] { CK(hipMemsetAsync(out, 0x11, ob, st)); });
    say("fill %zu MB: %.4f ms  %.2f TB/s written\n", ob >> 20, ms, ob / ms * 1e-9);
    ms = timeit([&] { CK(hipMemcpyAsync(out2, out, ob, hipMemcpyDeviceToDevice, st)); });
    say("copy %zu MB: %.4f ms  %.2f TB/s read + %.2f TB/s written\n", ob >> 20, ms, ob / ms * 1e-9, ob / ms * 1e-9);
    for (int K : {256, 512}) {
        for (int variant = 0; variant < 4; ++variant) {
            ctrl_igemm_desc d;
            memset(&d, 0, sizeof d);
            d.A = A; d.lda = (variant & 1) ? 0 : K; d.mode = 0; d.Cin = K; d.taps = 1;
            d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
            d.W = W; d.M = M; d.Nout = N; d.Ktot = K; d.rows_per_img = 1; d.scale = 1.f;
            d.nseg = 1;
            d.seg[0].out = out; d.seg[0].ld = (variant & 2) ? 0 : N; d.seg[0].ncols = N; d.seg[0].dtype = CTRL_F16; d.seg[0].L = 1;
            ms = timeit([&] { ctrl_op_igemm(&d, st); });
            say("K %d  A rows %-8s out rows %-8s %.4f ms  %.0f TFLOP/s\n", K, (variant & 1) ? "aliased" : "distinct",
                (variant & 2) ? "aliased" : "distinct", ms, 2.0 * M * N * K / ms * 1e-9);
        }
    }
    say("\ndone\n");
    return 0;
}

// Numerical check of the GEMM epilogue forms against a host reference (double accumulation) on a sample of rows that covers
// whole tiles at both ends of M plus scattered rows: out = ((A.W^T + bias [+ rowvec]) [GEGLU | SiLU] + residual) * scale.
static float h2f(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    if (e == 0) return (s ? -1.f : 1.f) * ldexpf((float)m, -24);
    return (s ? -1.f : 1.f) * ldexpf((float)(m | 1024), (int)e - 25);
}
static int check() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    struct C { const char* name; int M, N, K; bool geglu; int res; bool rowvec; int unused; };   // res: 0 none, 1 fp32 stream (+ mirror), 2 fp16
    const C cs[] = {
        {"plain + bias, 256x256", 8192, 2048, 64, false, 0, false, 0},
        {"geglu + bias, 256x256", 8192, 4096, 64, true, 0, false, 0},
        {"fp32 stream, 256x256 (residual fetched ahead)", 8192, 2048, 64, false, 1, false, 0},
        {"fp32 stream, ragged M", 8192 - 37, 2048, 64, false, 1, false, 0},
        {"fp32 stream, 128x256 two-workgroup tile", 65536, 512, 64, false, 1, false, 0},
        {"geglu, 256x128 two-workgroup tile", 65536, 512, 64, true, 0, false, 0},
        {"fp16 residual, 256x256", 8192, 2048, 64, false, 2, false, 0},
        {"fp32 stream, 256x320 (N = 640)", 32768, 640, 64, false, 1, false, 0},
        {"rowvec + SiLU, 256x256", 8192, 2048, 64, false, 0, true, 0},
        {"small tile, M 1000 N 512 K 96", 1000, 512, 96, false, 1, false, 0},
    };
    int bad = 0;
    for (const C& c : cs) {
        const int on = c.geglu ? c.N / 2 : c.N;
        std::vector<uint16_t> hA((size_t)c.M * c.K), hW((size_t)c.N * c.K);
        fill_half(hA, 11);
        fill_half(hW, 12);
        std::vector<float> hb(c.N), hres32, hrv;
        std::vector<uint16_t> hres16;
        for (int i = 0; i < c.N; ++i) hb[i] = 0.05f * (float)((i * 37) % 41 - 20);
        void *A, *W, *out, *mirror = nullptr, *res = nullptr;
        float *bias, *rowvec = nullptr;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&W, hW.size() * 2)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc((void**)&bias, c.N * 4)); CK(hipMemcpy(bias, hb.data(), c.N * 4, hipMemcpyHostToDevice));
        const bool f32out = c.res == 1;
        const size_t ob = (size_t)c.M * on * (f32out ? 4 : 2);
        CK(hipMalloc(&out, ob)); CK(hipMemset(out, 0xff, ob));
        if (c.res == 1) {
            hres32.resize((size_t)c.M * on);
            for (size_t i = 0; i < hres32.size(); ++i) hres32[i] = 0.25f * (float)((int)((i * 2654435761u) >> 20 & 1023) - 512) / 64.f;
            CK(hipMalloc(&res, hres32.size() * 4)); CK(hipMemcpy(res, hres32.data(), hres32.size() * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&mirror, (size_t)c.M * on * 2)); CK(hipMemset(mirror, 0xff, (size_t)c.M * on * 2));
        } else if (c.res == 2) {
            hres16.resize((size_t)c.M * on);
            fill_half(hres16, 13);
            CK(hipMalloc(&res, hres16.size() * 2)); CK(hipMemcpy(res, hres16.data(), hres16.size() * 2, hipMemcpyHostToDevice));
        }
        const int rpi = 4096;
        if (c.rowvec) {
            hrv.resize((size_t)((c.M + rpi - 1) / rpi) * c.N);
            for (size_t i = 0; i < hrv.size(); ++i) hrv[i] = 0.1f * (float)((int)(i % 23) - 11);
            CK(hipMalloc((void**)&rowvec, hrv.size() * 4)); CK(hipMemcpy(rowvec, hrv.data(), hrv.size() * 4, hipMemcpyHostToDevice));
        }
        ctrl_igemm_desc d;
        memset(&d, 0, sizeof d);
        d.A = A; d.lda = c.K; d.mode = 0; d.Cin = c.K; d.taps = 1;
        d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
        d.W = W; d.M = c.M; d.Nout = c.N; d.Ktot = c.K;
        d.bias = bias; d.rows_per_img = c.rowvec ? rpi : 1; d.scale = 0.75f; d.geglu = c.geglu;
        if (c.rowvec) { d.rowvec = rowvec; d.rowvec_ld = c.N; d.act = 1; }
        if (c.res) { d.res = res; d.ldres = on; d.res_f32 = c.res == 1; }
        if (c.res == 1) { d.out16 = mirror; d.ld16 = on; }
        d.nseg = 1;
        d.seg[0].out = out; d.seg[0].ld = on; d.seg[0].ncols = on; d.seg[0].dtype = f32out ? CTRL_F32 : CTRL_F16; d.seg[0].L = 1;
        if (ctrl_op_igemm(&d, st) != 0) { say("%s: launch failed: %s\n", c.name, ctrl_last_error()); return 3; }
        CK(hipStreamSynchronize(st));
        std::vector<uint8_t> ho(ob), hm;
        CK(hipMemcpy(ho.data(), out, ob, hipMemcpyDeviceToHost));
        if (mirror) { hm.resize((size_t)c.M * on * 2); CK(hipMemcpy(hm.data(), mirror, hm.size(), hipMemcpyDeviceToHost)); }
        // rows to check: two whole 256-row tiles at the start, the last 300 rows, 200 scattered ones
        std::vector<int> rows;
        for (int r = 0; r < 512 && r < c.M; ++r) rows.push_back(r);
        for (int r = c.M > 300 ? c.M - 300 : 0; r < c.M; ++r) if (r >= 512) rows.push_back(r);
        for (int i = 0; i < 200; ++i) { const int r = (int)(((uint64_t)i * 2654435761u + 977) % (uint64_t)c.M); if (r >= 512 && r < c.M - 300) rows.push_back(r); }
        std::vector<float> fW((size_t)c.N * c.K);
        for (size_t i = 0; i < fW.size(); ++i) fW[i] = h2f(hW[i]);
        double max_ref = 0, max_err = 0, max_merr = 0;
        std::vector<float> fa(c.K);
        std::vector<double> dots(c.N);
        for (int r : rows) {
            for (int k = 0; k < c.K; ++k) fa[k] = h2f(hA[(size_t)r * c.K + k]);
            for (int n = 0; n < c.N; ++n) {
                double acc = 0;
                const float* w = &fW[(size_t)n * c.K];
                for (int k = 0; k < c.K; ++k) acc += (double)fa[k] * (double)w[k];
                dots[n] = acc + hb[n] + (c.rowvec ? hrv[(size_t)(r / rpi) * c.N + n] : 0.f);
            }
            for (int o = 0; o < on; ++o) {
                double x;
                if (c.geglu) {
                    const int j = o / 16, cc = o % 16;
                    const double h = dots[32 * j + cc], g = dots[32 * j + 16 + cc];
                    x = h * 0.5 * g * (1.0 + erf(g / sqrt(2.0)));
                } else {
                    x = dots[o];
                }
                if (c.rowvec) x = x / (1.0 + exp(-x));
                if (c.res == 1) x += hres32[(size_t)r * on + o];
                if (c.res == 2) x += h2f(hres16[(size_t)r * on + o]);
                x *= 0.75;
                const double got = f32out ? (double)((const float*)ho.data())[(size_t)r * on + o] : (double)h2f(((const uint16_t*)ho.data())[(size_t)r * on + o]);
                max_ref = std::max(max_ref, fabs(x));
                max_err = std::max(max_err, fabs(got - x));
                if (mirror) max_merr = std::max(max_merr, fabs((double)h2f(((const uint16_t*)hm.data())[(size_t)r * on + o]) - x));
            }
        }
        const double rel = max_err / max_ref, mrel = max_merr / max_ref, tol = f32out ? 1e-5 : 6e-4;
        const bool ok = rel <= tol && (!mirror || mrel <= 6e-4) && max_ref > 0.1;
        say("%-50s rows %4zu  max|ref| %.3f  rel-inf %.2e%s   %s\n", c.name, rows.size(), max_ref, rel,
            mirror ? (std::string("  mirror ") + std::to_string(mrel)).c_str() : "", ok ? "ok" : "FAIL");
        bad += !ok;
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(out));
        if (res) CK(hipFree(res));
        if (mirror) CK(hipFree(mirror));
        if (rowvec) CK(hipFree(rowvec));
    }
    say(bad ? "\n%d case(s) FAILED\n" : "\nall cases ok\n", bad);
    return bad ? 4 : 0;
}

// The path's heaviest GEMM / convolution shapes with their real epilogue kinds (profiles/r0N_per_kernel_*.json), random
// operands, median of 9 launches each -- the table the round-4 review's "done" criteria read (M131072 N512 K2048, M32768 N640
// K5760 taps9, M131072 N4096 K512 geglu, ...).  `CTRL_IGEMM8=0 gemm_order_bench out.txt shapes` times the round-3 kernels.
static int shapes_mode(bool small = false) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // kind: 0 plain fp16 rows, 1 geglu, 2 fp32 stream update (fp32 residual in, fp32 master + fp16 mirror out), 3 fp16 rows with an
    //       fp32 residual in (operand of the next GEMM), 4 transposed fp16 output (V^T), 5 conv 3x3 (taps 9) fp16 rows + time vector
    struct S { const char* name; int M, N, K, kind, hw, up; };
    const S ss[] = {
        {"ff1 geglu 512->4096 @128^2", 131072, 4096, 512, 1, 0, 0},
        {"attn out 320->512 stream @128^2", 131072, 512, 320, 2, 0, 0},
        {"ff2 2048->512 stream @128^2", 131072, 512, 2048, 2, 0, 0},
        {"q proj 512->320 @128^2", 131072, 320, 512, 0, 0, 0},
        {"v proj 512->320 transposed @128^2", 131072, 320, 512, 4, 16384, 0},
        {"q|k proj 512->640 @128^2", 131072, 640, 512, 0, 0, 0},
        {"conv 320->320 @128^2", 131072, 320, 2880, 5, 128, 1},
        {"conv 320->320 @64^2 up2 -> 128^2", 131072, 320, 2880, 5, 128, 2},
        {"conv 640->640 @64^2", 32768, 640, 5760, 5, 64, 1},
        {"conv 1280->1280 @32^2", 8192, 1280, 11520, 5, 32, 1},
        {"ff1 geglu 512->4096 @64^2", 32768, 4096, 512, 1, 0, 0},
        {"ff2 2048->512 stream @64^2", 32768, 512, 2048, 2, 0, 0},
        {"attn out 640->512 stream @64^2", 32768, 512, 640, 2, 0, 0},
        {"q|k 512->1280 @64^2", 32768, 1280, 512, 0, 0, 0},
        {"ControlNet ff1 geglu 640->5120 @32^2", 8192, 5120, 640, 1, 0, 0},
        {"plain 2048-deep operand out", 131072, 512, 2048, 3, 0, 0},
    };
    // `small`: the short launches of the ControlNet chain at b = 8 and of the 64^2 / 32^2 adapter levels (M = 32768 .. 512), where a
    // launch is a few waves of tiles: the tile experiments of round 5 (CTRL_IGEMM_FORCE=<tile> gemm_order_bench out.txt small)
    const S sm[] = {
        {"q proj 512->320 @128^2", 131072, 320, 512, 0, 0, 0},
        {"CN proj 320->320 @64^2", 32768, 320, 320, 0, 0, 0},
        {"CN q|k|v 320->960 @64^2", 32768, 960, 320, 0, 0, 0},
        {"CN attn out 320->320 stream @64^2", 32768, 320, 320, 2, 0, 0},
        {"CN ff2 1280->320 stream @64^2", 32768, 320, 1280, 2, 0, 0},
        {"adapter q|k|v 512->1920 @64^2", 32768, 1920, 512, 0, 0, 0},
        {"adapter attn out 640->512 @64^2", 32768, 512, 640, 3, 0, 0},
        {"CN proj 640->640 @32^2", 8192, 640, 640, 0, 0, 0},
        {"CN q|k|v 640->1920 @32^2", 8192, 1920, 640, 0, 0, 0},
        {"CN attn out 640->640 stream @32^2", 8192, 640, 640, 2, 0, 0},
        {"CN ff2 2560->640 stream @32^2", 8192, 640, 2560, 2, 0, 0},
        {"adapter q|k|v 512->3840 @32^2", 8192, 3840, 512, 0, 0, 0},
        {"CN q|k|v 1280->3840 @16^2", 2048, 3840, 1280, 0, 0, 0},
        {"CN attn out 1280->1280 stream @16^2", 2048, 1280, 1280, 2, 0, 0},
        {"CN ff2 5120->1280 stream @16^2", 2048, 1280, 5120, 2, 0, 0},
        {"CN attn out 1280->1280 stream @8^2", 512, 1280, 1280, 2, 0, 0},
        {"CN ff1 geglu 1280->10240 @8^2", 512, 10240, 1280, 1, 0, 0},
    };
    say("\npath shapes (median of 9, random fp16 operands; TFLOP/s = algorithmic 2 M N K / t)%s%s\n", getenv("CTRL_IGEMM_FORCE") ? "  CTRL_IGEMM_FORCE=" : "",
        getenv("CTRL_IGEMM_FORCE") ? getenv("CTRL_IGEMM_FORCE") : "");
    const S* list = small ? sm : ss;
    const int nlist = small ? (int)(sizeof(sm) / sizeof(sm[0])) : (int)(sizeof(ss) / sizeof(ss[0]));
    for (int li = 0; li < nlist; ++li) {
        const S& sh = list[li];
        const int on = sh.kind == 1 ? sh.N / 2 : sh.N;
        const bool conv = sh.kind == 5;
        const int cin = conv ? sh.K / 9 : sh.K;
        const int hin = conv ? sh.hw / sh.up : 1;
        const int nimg = conv ? sh.M / (sh.hw * sh.hw) : 1;
        const size_t a_elems = conv ? (size_t)nimg * hin * hin * cin : (size_t)sh.M * sh.K;
        void* A = dev_half(a_elems, 1);
        void* W = dev_half((size_t)sh.N * sh.K, 2);
        std::vector<float> hb(sh.N, 0.01f);
        float* bias = nullptr;
        CK(hipMalloc((void**)&bias, sh.N * 4));
        CK(hipMemcpy(bias, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        const bool f32out = sh.kind == 2;
        const size_t out_bytes = (size_t)sh.M * on * (f32out ? 4 : 2) + (sh.kind == 4 ? 4096 : 0);
        void *out = nullptr, *res = nullptr, *mirror = nullptr;
        float* rowvec = nullptr;
        CK(hipMalloc(&out, out_bytes));
        if (sh.kind == 2 || sh.kind == 3) {
            CK(hipMalloc(&res, (size_t)sh.M * on * 4));
            CK(hipMemset(res, 0, (size_t)sh.M * on * 4));
        }
        if (sh.kind == 2) CK(hipMalloc(&mirror, (size_t)sh.M * on * 2));
        if (conv) { CK(hipMalloc((void**)&rowvec, (size_t)nimg * sh.N * 4)); CK(hipMemset(rowvec, 0, (size_t)nimg * sh.N * 4)); }
        ctrl_igemm_desc d;
        memset(&d, 0, sizeof d);
        d.A = A; d.lda = cin; d.mode = conv ? 1 : 0; d.Cin = cin; d.taps = conv ? 9 : 1;
        d.Hin = d.Win = hin; d.Hout = d.Wout = conv ? sh.hw : 1; d.stride = 1; d.up = conv ? sh.up : 1;
        d.W = W; d.M = sh.M; d.Nout = sh.N; d.Ktot = sh.K;
        d.bias = bias; d.rows_per_img = conv ? sh.hw * sh.hw : 1; d.scale = 1.f; d.geglu = sh.kind == 1;
        if (conv) { d.rowvec = rowvec; d.rowvec_ld = sh.N; d.act = 0; }
        if (res) { d.res = res; d.ldres = on; d.res_f32 = 1; }
        if (mirror) { d.out16 = mirror; d.ld16 = on; }
        d.nseg = 1;
        d.seg[0].out = out; d.seg[0].col_begin = 0; d.seg[0].ncols = on;
        if (sh.kind == 4) { d.seg[0].fmt = 1; d.seg[0].ld = sh.hw; d.seg[0].L = sh.hw; d.seg[0].dtype = CTRL_F16; }
        else { d.seg[0].fmt = 0; d.seg[0].ld = on; d.seg[0].L = 1; d.seg[0].dtype = f32out ? CTRL_F32 : CTRL_F16; }
        std::vector<float> t;
        bool failed = false;
        for (int i = 0; i < 2 && !failed; ++i)
            if (ctrl_op_igemm(&d, st) != 0) { say("  %-40s launch failed: %s\n", sh.name, ctrl_last_error()); failed = true; }
        for (int i = 0; i < 9 && !failed; ++i) {
            CK(hipEventRecord(e0, st));
            ctrl_op_igemm(&d, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        if (!failed) {
            std::sort(t.begin(), t.end());
            say("  %-40s M%6d N%5d K%5d  %.4f ms  %5.0f TFLOP/s\n", sh.name, sh.M, sh.N, sh.K, t[4], 2.0 * sh.M * sh.N * sh.K / t[4] * 1e-9);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(out));
        if (res) CK(hipFree(res));
        if (mirror) CK(hipFree(mirror));
        if (rowvec) CK(hipFree(rowvec));
    }
    say("done\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    say("abi %d\n", ctrl_abi_version());
    if (argc > 2 && !strcmp(argv[2], "ksweep")) return ksweep();
    if (argc > 2 && !strcmp(argv[2], "stores")) return stores();
    if (argc > 2 && !strcmp(argv[2], "check")) return check();
    if (argc > 2 && !strcmp(argv[2], "shapes")) return shapes_mode();
    if (argc > 2 && !strcmp(argv[2], "small")) return shapes_mode(true);
    std::vector<Shape> shapes = {
        {"geglu 512->4096 @128^2 x8", 131072, 4096, 512, true, false, {"legacy", "auto", "m,8", "m,16", "m,4", "n,0"}},
        {"geglu 512->4096 @64^2 x8", 32768, 4096, 512, true, false, {"legacy", "auto", "m,8", "m,16", "n,0"}},
        {"geglu 1280->10240 @16^2 x8 (ControlNet)", 2048, 10240, 1280, true, false, {"legacy", "auto", "n,0", "n,2", "m,3"}},
        {"geglu 640->5120 @32^2 x8 (ControlNet)", 8192, 5120, 640, true, false, {"legacy", "auto", "n,0", "m,6", "m,3"}},
        {"ff out 2048->512 fp32 stream @128^2 x8", 131072, 512, 2048, false, true, {"legacy", "auto", "m,2", "m,1"}},
        {"q|k|v 512->960 @128^2 x8", 131072, 960, 512, false, false, {"legacy", "auto", "m,1", "m,2"}},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const int on = sh.geglu ? sh.N / 2 : sh.N;
        void* A = dev_half((size_t)sh.M * sh.K, 1);
        void* W = dev_half((size_t)sh.N * sh.K, 2);
        std::vector<float> hb(sh.N, 0.01f);
        float* bias = nullptr;
        CK(hipMalloc((void**)&bias, sh.N * 4));
        CK(hipMemcpy(bias, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        const size_t out_bytes = (size_t)sh.M * on * (sh.stream ? 4 : 2);
        void *out = nullptr, *res = nullptr, *mirror = nullptr;
        CK(hipMalloc(&out, out_bytes));
        if (sh.stream) {
            CK(hipMalloc(&res, out_bytes));
            CK(hipMemset(res, 0, out_bytes));
            CK(hipMalloc(&mirror, (size_t)sh.M * on * 2));
        }
        ctrl_igemm_desc d;
        memset(&d, 0, sizeof d);
        d.A = A; d.lda = sh.K; d.mode = 0; d.Cin = sh.K; d.taps = 1;
        d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
        d.W = W; d.M = sh.M; d.Nout = sh.N; d.Ktot = sh.K;
        d.bias = bias; d.rows_per_img = 1; d.scale = 1.f; d.geglu = sh.geglu;
        if (sh.stream) { d.res = res; d.ldres = on; d.res_f32 = 1; d.out16 = mirror; d.ld16 = on; }
        d.nseg = 1;
        d.seg[0].out = out; d.seg[0].ld = on; d.seg[0].col_begin = 0; d.seg[0].ncols = on; d.seg[0].fmt = 0;
        d.seg[0].dtype = sh.stream ? CTRL_F32 : CTRL_F16; d.seg[0].L = 1;
        const double flops = 2.0 * sh.M * sh.N * sh.K;
        say("\n%s   M%d N%d K%d\n", sh.name, sh.M, sh.N, sh.K);
        std::vector<std::vector<float>> t(sh.orders.size());
        std::vector<uint64_t> sums(sh.orders.size(), 0);
        for (int round = 0; round < 3; ++round) {               // interleaved so that clock drift hits every order alike
            for (size_t o = 0; o < sh.orders.size(); ++o) {
                if (ctrl_igemm_set_order(sh.orders[o]) != 0) { say("  order %s refused\n", sh.orders[o]); continue; }
                for (int i = 0; i < 2; ++i)
                    if (ctrl_op_igemm(&d, st) != 0) { say("  launch failed: %s\n", ctrl_last_error()); return 3; }
                for (int i = 0; i < 8; ++i) {
                    CK(hipEventRecord(e0, st));
                    ctrl_op_igemm(&d, st);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t[o].push_back(ms);
                }
                if (round == 0 && out_bytes <= (size_t)160 << 20) sums[o] = checksum(out, out_bytes);
            }
        }
        for (size_t o = 0; o < sh.orders.size(); ++o) {
            if (t[o].empty()) continue;
            std::sort(t[o].begin(), t[o].end());
            const float med = t[o][t[o].size() / 2], mn = t[o][0];
            say("  %-7s median %.4f ms (%.0f TFLOP/s)  min %.4f ms   checksum %016llx%s\n", sh.orders[o], med, flops / med * 1e-9,
                mn, (unsigned long long)sums[o], sums[o] == sums[0] ? "" : "   <-- DIFFERS FROM legacy");
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(out));
        if (res) CK(hipFree(res));
        if (mirror) CK(hipFree(mirror));
    }
    ctrl_igemm_set_order("legacy");
    say("\ndone\n");
    return 0;
}
