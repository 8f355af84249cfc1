// occupancy probe of the default head_dim-64 attention kernel (side build): what the runtime reports, and a timing test --
// the same kernel with 32 / 64 / 128 workgroups per XCD (1 / 2 / 4 per CU's worth): if two workgroups share a CU the time of 64
// per XCD stays near the time of 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "ctrl_hip.h"
extern "C" int ctrl_debug_attn_occupancy(int* out, int n);
int main(int argc, char** argv) {
    FILE* f = argc > 1 ? fopen(argv[1], "w") : nullptr;
    auto say = [&](const char* s) { fputs(s, stdout); if (f) fputs(s, f); };
    char buf[512];
    int o[16] = {0};
    const int k = ctrl_debug_attn_occupancy(o, 16);
    snprintf(buf, sizeof buf, "hipOccupancyMaxActiveBlocksPerMultiprocessor(<1,8,DEFER,3>, 512 threads, 48 KB): %d (after the dynamic-LDS attribute: %d); 2-deep ring 32 KB: %d; 4-deep 64 KB: %d\n", o[0], o[1], o[2], o[3]);
    say(buf);
    snprintf(buf, sizeof buf, "sharedMemPerBlock %d  maxSharedMemoryPerMultiProcessor %d  regsPerBlock %d  regsPerMultiprocessor %d  maxThreadsPerMultiProcessor %d  CUs %d  clock kHz %d\n", o[4], o[5], o[6], o[7], o[8], o[9], o[10]);
    say(buf);
    snprintf(buf, sizeof buf, "kernel attributes: numRegs %d  static LDS %d  maxThreadsPerBlock %d  scratch %d   (%d values)\n", o[11], o[12], o[13], o[14], k);
    say(buf);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int B = 8, H = 1, C = 64, Lk = 16384, LqMax = 32768;
    void *q, *kk, *v, *out;
    hipMalloc(&q, (size_t)B * LqMax * C * 2); hipMalloc(&kk, (size_t)B * Lk * C * 2); hipMalloc(&v, (size_t)B * Lk * C * 2); hipMalloc(&out, (size_t)B * LqMax * C * 2);
    hipMemset(q, 0, (size_t)B * LqMax * C * 2); hipMemset(kk, 0, (size_t)B * Lk * C * 2); hipMemset(v, 0, (size_t)B * Lk * C * 2);
    ctrl_attn_set_variant(2);
    // data dependence: pseudo-random fp16 (|x| < 1.5; K additionally scaled by log2(e)/8 like the product's K) vs zeros
    auto fill = [&](void* dst, size_t n, float scale, unsigned seed) {
        std::vector<unsigned short> h(n);
        unsigned s = seed;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            const float x = ((float)((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 3.0f * scale;
            _Float16 hx = (_Float16)x;
            memcpy(&h[i], &hx, 2);
        }
        hipMemcpy(dst, h.data(), n * 2, hipMemcpyHostToDevice);
    };
    const char* names[4] = {"Q = K = V = 0", "Q, K random, V = 0", "Q = K = 0, V random", "Q, K, V random"};
    if (argc > 4 && !strcmp(argv[2], "loop")) {      // occ_probe out.txt loop <cfg 0|3> <launches>: a long run to read the clocks beside
        const int c = atoi(argv[3]), n = atoi(argv[4]);
        if (c & 1) { fill(q, (size_t)B * LqMax * C, 1.f, 1); fill(kk, (size_t)B * Lk * C, 0.18f, 2); }
        if (c & 2) fill(v, (size_t)B * Lk * C, 1.f, 3);
        ctrl_attn_desc d; memset(&d, 0, sizeof d);
        d.Q = q; d.ldq = C; d.K = kk; d.ldk = C; d.Vt = v; d.Lkpad = Lk; d.kvB = B; d.O = out; d.ldo = C;
        d.B = B; d.heads = H; d.D = 64; d.Lq = 32768; d.Lk = Lk; d.scale = 0.125f; d.k_prescaled = 1;
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) ctrl_op_flash_attn(&d, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        snprintf(buf, sizeof buf, "loop %-18s %d launches %.1f ms  %.1f TFLOP/s\n", names[c & 3], n, ms, 4.0 * B * H * 32768.0 * Lk * 64 * n / ms * 1e-9);
        say(buf);
        if (f) fclose(f);
        return 0;
    }
    for (int cfg = 0; cfg < 5; ++cfg) {
        const int c = cfg % 4;
        if (c & 1) { fill(q, (size_t)B * LqMax * C, 1.f, 1); fill(kk, (size_t)B * Lk * C, 0.18f, 2); }
        else { hipMemset(q, 0, (size_t)B * LqMax * C * 2); hipMemset(kk, 0, (size_t)B * Lk * C * 2); }
        if (c & 2) fill(v, (size_t)B * Lk * C, 1.f, 3); else hipMemset(v, 0, (size_t)B * Lk * C * 2);
        ctrl_attn_desc d; memset(&d, 0, sizeof d);
        d.Q = q; d.ldq = C; d.K = kk; d.ldk = C; d.Vt = v; d.Lkpad = Lk; d.kvB = B; d.O = out; d.ldo = C;
        d.B = B; d.heads = H; d.D = 64; d.Lq = 32768; d.Lk = Lk; d.scale = 0.125f; d.k_prescaled = 1;
        for (int i = 0; i < 2; ++i) ctrl_op_flash_attn(&d, st);
        hipEventRecord(e0, st);
        for (int i = 0; i < 12; ++i) ctrl_op_flash_attn(&d, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 12;
        snprintf(buf, sizeof buf, "B8 h1 Lq32768 Lk16384  %-22s %.4f ms  %.1f TFLOP/s\n", names[c], ms, 4.0 * B * H * 32768.0 * Lk * 64 / ms * 1e-9);
        say(buf);
    }
    hipMemset(q, 0, (size_t)B * LqMax * C * 2); hipMemset(kk, 0, (size_t)B * Lk * C * 2); hipMemset(v, 0, (size_t)B * Lk * C * 2);
    for (int Lq : {8192, 16384}) {
        ctrl_attn_desc d; memset(&d, 0, sizeof d);
        d.Q = q; d.ldq = C; d.K = kk; d.ldk = C; d.Vt = v; d.Lkpad = Lk; d.kvB = B; d.O = out; d.ldo = C;
        d.B = B; d.heads = H; d.D = 64; d.Lq = Lq; d.Lk = Lk; d.scale = 0.125f; d.k_prescaled = 1;
        for (int i = 0; i < 2; ++i) if (ctrl_op_flash_attn(&d, st)) { say(ctrl_last_error()); return 2; }
        hipEventRecord(e0, st);
        for (int i = 0; i < 5; ++i) ctrl_op_flash_attn(&d, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        snprintf(buf, sizeof buf, "B8 h1 Lq%-6d Lk16384: %3d workgroups per XCD (32 CUs)  %.4f ms  %.1f TFLOP/s\n", Lq, Lq / 256, ms, 4.0 * B * H * (double)Lq * Lk * 64 / ms * 1e-9);
        say(buf);
    }
    if (f) fclose(f);
    return 0;
}
