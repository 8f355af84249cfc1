// Lab harness for the round-4 implicit-GEMM main loop (torch-free, self-contained): out[M][N] fp16 = A[M][K] . W[N][K]^T + bias.
//
//   gemm8_kernel<EPI, PERSIST, OPT>: 256x256x64 tile, 8 waves (2 M x 4 N, 128x64 per wave), 2 LDS buffers x 4 half-tiles of 16 KiB,
//   8 phases per two k-tiles (cdna_hip_programming.md "The 256^2 8-phase template"): per phase {fragment reads of one sub-block,
//   one half-tile of LDS-DMA, barrier, 16 MFMAs of one C quadrant, barrier}, the two wave groups (wm = 0 / 1) one barrier apart.
//   EPI 0: rows staged per 16-row slab through LDS (the product's scheme); EPI 1: weight rows permuted while staging so a lane
//   owns 8 consecutive output columns of a row per fragment pair -> 16-byte stores straight from registers, no LDS.
//   PERSIST 1: one workgroup per CU walks its tiles, the LDS ring runs on across tiles (needs EPI 1: the epilogue must not
//   touch LDS while the next tile's k-tiles stream in).
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/experiments/gemm8_lab.hip -o tools/bin/gemm8_lab
//   run:    tools/bin/gemm8_lab [out.txt]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ctrl_hip.h"          // the product library: the baseline timed beside the lab kernels

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

namespace g8 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT = 128 * BK * 2;          // bytes of one half-tile (128 rows x 128 B)
constexpr int BUF = 4 * HT;               // one k-tile: HA0, HA1, HB0, HB1
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };

// wait until at most `later` half-tiles (2 loads each) + `extra` other vector-memory operations (the previous tile's stores,
// persistent form) are in flight.  The counter is in order: everything older than those has then landed.
__device__ __forceinline__ void vmcnt_pairs(int later, int extra = 0) {
    if (extra == 16) {
        if (later >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (later == 3) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        return;
    }
    if (later >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace g8

// OPT bits: 1 = skew the workgroups' start (persistent form) so the CUs' store phases do not coincide; 2 = buffer_load ... lds
// (32-bit offsets off a descriptor, k offset in an SGPR) instead of global_load_lds; 4 = re-read b0 in phase 3 instead of keeping it
template <int EPI, int PERSIST, int OPT>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const half_t* __restrict__ A, long lda, const half_t* __restrict__ W, long ldw,
                                                       const float* __restrict__ bias, half_t* __restrict__ out, long ldo, int M, int N, int K,
                                                       int ntm, int ntn) {
    using namespace g8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / BK;
    const int ntiles = ntm * ntn;

    // ---- staging: LDS row j of a half-tile; pass i of this wave writes rows (i*8 + wave)*8 + lane/8 ----
    const int lrow = lane >> 3, lpos = lane & 7;
    long a_row[2][2], b_row[2][2];        // [sub-block][pass]: row index inside the 256-row tile
    int s_c8[2];                          // source chunk (in halfs) after the swizzle, per pass (same for A and B)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (i * 8 + wave) * 8 + lrow;                  // 0..127
        s_c8[i] = (lpos ^ ((j >> 1) & 7)) * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            a_row[s][i] = (j >> 6) * 128 + s * 64 + (j & 63);     // rows of wave row wm = j>>6, sub-block s
            const int wn_j = j >> 5, f = (j >> 4) & 1, g = (j >> 2) & 3, r = j & 3;
            if (EPI == 1) b_row[s][i] = wn_j * 64 + s * 32 + g * 8 + f * 4 + r;      // permuted: lane group g owns 8 consecutive columns
            else b_row[s][i] = wn_j * 64 + s * 32 + (j & 31);
        }
    }

    // XCD-aware: consecutive workgroup ids land on different XCDs; every XCD gets a contiguous range of the M-major tile list
    auto tile_origin = [&](int t, int& tm0, int& tn0) __attribute__((always_inline)) {
        const int xcd = t & 7, k = t >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        const int tm = lin / ntn, tn = lin - tm * ntn;
        tm0 = tm * BM;
        tn0 = tn * BN;
    };
    int tile = blockIdx.x;
    int m0, n0;
    tile_origin(tile, m0, n0);

    // source pointers of the tile being STAGED (the stream of k-tiles may run ahead into the next tile, persistent form)
    const half_t* a_src[2][2];
    const half_t* b_src[2][2];
    unsigned a_off[2][2], b_off[2][2];                     // OPT & 2: byte offsets off the descriptors
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)M * lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((size_t)N * ldw * 2), 0x00020000);
    auto set_src = [&](int sm0, int sn0) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (OPT & 2) {
                    a_off[s][i] = (unsigned)(((sm0 + (int)a_row[s][i]) * (int)lda + s_c8[i]) * 2);
                    b_off[s][i] = (unsigned)(((sn0 + (int)b_row[s][i]) * (int)ldw + s_c8[i]) * 2);
                } else {
                    a_src[s][i] = A + (sm0 + a_row[s][i]) * lda + s_c8[i];
                    b_src[s][i] = W + (sn0 + b_row[s][i]) * ldw + s_c8[i];
                }
            }
    };
    set_src(m0, n0);

    // stage half-tile `which` of k-tile kt (of the staged tile) into buffer `buf`
    auto stage = [&](int which, int kt, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * BUF + which * HT + wave * 1024;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (OPT & 2) {
                const unsigned off = which == HA0 ? a_off[0][i] : which == HA1 ? a_off[1][i] : which == HB0 ? b_off[0][i] : b_off[1][i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds((which == HA0 || which == HA1) ? a_rs : w_rs, (lptr_t)(dst + i * 8192), 16, off, k0 * 2, 0, 0);
            } else {
                const half_t* src = (which == HA0 ? a_src[0][i] : which == HA1 ? a_src[1][i] : which == HB0 ? b_src[0][i] : b_src[1][i]) + k0;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + i * 8192), 16, 0, 0);
            }
        }
    };

    // ---- fragment reads: one per-lane base per operand and k-step, everything else is an immediate; the buffer toggles by XOR ----
    const int frow = lane & 15, fch = lane >> 4;
    const int sw = (frow >> 1) & 7;
    int ra0 = (wm * 64 + frow) * 128 + ((fch ^ sw) << 4);           // + mi*2048
    int rb0 = (wn * 32 + frow) * 128 + ((fch ^ sw) << 4);           // + f*2048
    int ra1 = ra0 ^ 64, rb1 = rb0 ^ 64;                             // second k-step: logical chunk + 4

    f4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};

    h8 af[2][4], b0f[2][2], b1f[2][2];       // [kk][frag]
    auto read_a = [&](int s) __attribute__((always_inline)) {
        const char* base = smem + (s ? HA1 : HA0) * HT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[kk][mi] = *(const h8*)(base + (kk ? ra1 : ra0) + mi * 2048);
    };
    auto read_b = [&](int s, h8 (&bf)[2][2]) __attribute__((always_inline)) {
        const char* base = smem + (s ? HB1 : HB0) * HT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < 2; ++f) bf[kk][f] = *(const h8*)(base + (kk ? rb1 : rb0) + f * 2048);
    };
    auto mma = [&](int as, int bs, h8 (&bf)[2][2]) __attribute__((always_inline)) {          // quadrant (as, bs): rows as*64.., columns bs*32..
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    acc[as * 4 + mi][bs * 2 + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][f], af[kk][mi], acc[as * 4 + mi][bs * 2 + f], 0, 0, 0);
    };

#define PHASE_SYNC_PRE()                                   \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_setprio(1);
#define PHASE_SYNC_POST()                                  \
    __builtin_amdgcn_s_setprio(0);                         \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

    // Half-tiles are issued in the order HA0, HB0, HB1, HA1 of k-tile 0, HA0, HB0 of k-tile 1, then one per phase:
    //   phase 0 of k-tile e: HB1[e+1]   phase 1: HA1[e+1]   phase 2: HA0[e+2]   phase 3: HB0[e+2]
    // and are read one phase after the wait that covers them (the wave groups run one barrier apart):
    //   HB1[e] in phase 1 (wait in phase 0), HA1[e] in phase 2 (wait in phase 1), HA0 / HB0[e+1] in phase 0 of e+1 (wait in phase 3).
    // The counter is in order, so "at most n younger operations in flight" = the wanted half-tile has landed:
    //   phase 0 / 1: 4 younger half-tiles (8 loads) when k-tile e+1 exists, else HA1[e] (2) / nothing (0)
    //   phase 3:     4 younger when k-tile e+2 exists, else HB1, HA1 of e+1 (4)
    //   + 16 in the first k-tile after an epilogue (persistent form): that epilogue's stores sit in the window.
    // A region is re-staged two phases after its last read at the earliest (the other group's reads of phase p are only
    // known complete after its second barrier of phase p).
    //
    // One loop iteration = one k-tile of the stream; n1 / n2 / x16 are wave-uniform (scalar branches around loads and waits).
    auto ktile = [&](const bool n1, const bool n2, const bool x16, auto&& stage1, auto&& stage2) __attribute__((always_inline)) {
        // phase 0: a0, b0 | stage HB1[e+1] | wait HB1[e]
        read_b(0, b0f);
        __builtin_amdgcn_sched_barrier(0);
        read_a(0);
        if (n1) stage1(HB1);
        if (x16) VMCNT(24); else if (n1) VMCNT(8); else VMCNT(2);
        PHASE_SYNC_PRE();
        mma(0, 0, b0f);
        PHASE_SYNC_POST();
        // phase 1: b1 | stage HA1[e+1] | wait HA1[e]
        read_b(1, b1f);
        if (n1) stage1(HA1);
        if (x16) VMCNT(24); else if (n1) VMCNT(8); else VMCNT(0);
        PHASE_SYNC_PRE();
        mma(0, 1, b1f);
        PHASE_SYNC_POST();
        // phase 2: a1 | stage HA0[e+2]
        read_a(1);
        if (n2) stage2(HA0);
        PHASE_SYNC_PRE();
        mma(1, 1, b1f);
        PHASE_SYNC_POST();
        // phase 3: (b0 still in registers) | stage HB0[e+2] | wait HA0[e+1], HB0[e+1]
        if constexpr (OPT & 4) read_b(0, b0f);
        if (n2) stage2(HB0);
        if (x16) { if (n2) VMCNT(24); else VMCNT(20); } else if (n2) VMCNT(8); else if (n1) VMCNT(4);
        PHASE_SYNC_PRE();
        mma(1, 0, b0f);
        PHASE_SYNC_POST();
        // the next k-tile lives in the other buffer
        ra0 ^= BUF; ra1 ^= BUF; rb0 ^= BUF; rb1 ^= BUF;
    };

    // ---- epilogues ----
    auto epilogue = [&]() __attribute__((always_inline)) {
        if constexpr (EPI == 1) {
            // lane (g = lane>>4, tok = lane&15): row mi*16 + tok, columns s*32 + g*8 + [0, 8) of the wave's 64 for sub-block s
            const int g = lane >> 4, tok = lane & 15;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int col = n0 + wn * 64 + s * 32 + g * 8;
                f4 bv0 = f4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
                if (bias) { bv0 = *(const f4*)(bias + col); bv1 = *(const f4*)(bias + col + 4); }
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int row = m0 + wm * 128 + mi * 16 + tok;
                    const f4 x0 = acc[mi][2 * s] + bv0, x1 = acc[mi][2 * s + 1] + bv1;
                    h8 pk = {(half_t)x0[0], (half_t)x0[1], (half_t)x0[2], (half_t)x0[3], (half_t)x1[0], (half_t)x1[1], (half_t)x1[2], (half_t)x1[3]};
                    *(h8*)(out + (size_t)row * ldo + col) = pk;          // (the lab's M is a multiple of 256; exactly 16 stores per lane)
                }
            }
        } else {
            // per 16-row slab: fragments -> wave-private LDS area (fp32 [16][68]) -> 16-byte pieces of full 128-byte row segments
            __syncthreads();
            constexpr int SLD = 68;
            float* stg = (float*)smem + wave * (16 * SLD);
            const int erow = lane & 15, ecol = (lane >> 4) * 4;
            f4 bias_v[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bias_v[ni] = bias ? *(const f4*)(bias + n0 + wn * 64 + ni * 16 + ecol) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) *(f4*)(stg + erow * SLD + ni * 16 + ecol) = acc[mi][ni] + bias_v[ni];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                    const int row = m0 + wm * 128 + mi * 16 + r;
                    const f4 v0 = *(const f4*)(stg + r * SLD + c8 * 8), v1 = *(const f4*)(stg + r * SLD + c8 * 8 + 4);
                    h8 pk = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3], (half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                    if (row < M) *(h8*)(out + (size_t)row * ldo + n0 + wn * 64 + c8 * 8) = pk;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    };

    if constexpr ((OPT & 1) && PERSIST) {
        // start the workgroups 1/16 of a tile period apart (16 phases, two CUs of every XCD per phase): the tiles of one launch
        // take the same time, so without it every CU reaches its store burst at the same moment and HBM sees write bursts
        // separated by idle gaps instead of a steady stream
        const int ph = (blockIdx.x >> 3) & 15;
        const int units = ph * (nk * 56 + 96) / 16;            // s_sleep 1 = 64 cycles; a tile ~ nk * 3600 + 6000 cycles
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // ---- prologue of the stream: HA0, HB0, HB1, HA1 of k-tile 0, HA0, HB0 of k-tile 1 ----
    const int stride = gridDim.x;
    const int my_tiles = PERSIST ? (ntiles - tile + stride - 1) / stride : 1;
    const int total = my_tiles * nk;                       // k-tiles of this workgroup's stream (persistent form: nk >= 2)
    stage(HA0, 0, 0); stage(HB0, 0, 0); stage(HB1, 0, 0); stage(HA1, 0, 0);
    if (nk > 1) { stage(HA0, 1, 1); stage(HB0, 1, 1); VMCNT(8); } else VMCNT(4);      // (persistent form: nk >= 2)
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // the second wave group runs one barrier behind

    if constexpr (!PERSIST) {
        for (int kt = 0; kt < nk; ++kt) {
            const int nb = (kt + 1) & 1;
            ktile(kt + 1 < nk, kt + 2 < nk, false, [&](int which) __attribute__((always_inline)) { stage(which, kt + 1, nb); },
                  [&](int which) __attribute__((always_inline)) { stage(which, kt + 2, nb ^ 1); });
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        epilogue();
    } else {
        // Stream element e = it * nk + kin.  The four half-tiles of an element are issued back to back (HA0, HB0 in phases 2 / 3 of
        // element e - 2, HB1, HA1 in phases 0 / 1 of element e - 1), so ONE set of source pointers serves: it moves to the next tile
        // right before HA0 of that tile's k-tile 0 is staged.
        int e = 0;
        int skt = nk > 2 ? 2 : 0;                            // k-tile index, within its own tile, of stream element e + 2
        int staged_tile = tile;
        for (int it = 0; it < my_tiles; ++it) {
            for (int kin = 0; kin < nk; ++kin, ++e) {
                const bool n1 = e + 1 < total, n2 = e + 2 < total;
                const int nb = (e + 1) & 1;
                const int k1 = (skt == 0) ? nk - 1 : skt - 1;      // k-tile index of element e + 1 within ITS tile
                const bool sw_tile = n2 && skt == 0;
                ktile(n1, n2, it > 0 && kin == 0,
                      [&](int which) __attribute__((always_inline)) { stage(which, k1, nb); },
                      [&](int which) __attribute__((always_inline)) {
                          if (which == HA0 && sw_tile) { int tm0, tn0; staged_tile += stride; tile_origin(staged_tile, tm0, tn0); set_src(tm0, tn0); }
                          stage(which, skt, nb ^ 1);
                      });
                skt = skt + 1 == nk ? 0 : skt + 1;
            }
            epilogue();
            tile += stride;
            if (it + 1 < my_tiles) {
                tile_origin(tile, m0, n0);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    }
}

// ---------------- probe: what an out-of-range buffer_load ... lds leaves in LDS ----------------
__global__ void oob_probe_kernel(const half_t* A, int bytes, unsigned* dump) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* w = (unsigned*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) w[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, bytes, 0x00020000);
    // lanes 0-31 in range, lanes 32-63 out of range (offset 0x80000000 >= num_records)
    const unsigned off = threadIdx.x < 32 ? threadIdx.x * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dump[i] = w[i];
}

// ---------------- reference (naive, fp32 accumulate) ----------------
__global__ void ref_kernel(const half_t* A, long lda, const half_t* W, long ldw, const float* bias, half_t* out, long ldo, int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * lda + k] * (float)W[(size_t)n * ldw + k];
    out[(size_t)m * ldo + n] = (half_t)(s + (bias ? bias[n] : 0.f));
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

static void fill_half(std::vector<uint16_t>& v, uint64_t seed) {
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

template <int EPI, int PERSIST, int OPT>
static void launch(const half_t* A, long lda, const half_t* W, long ldw, const float* bias, half_t* out, long ldo, int M, int N, int K, hipStream_t st) {
    static bool attr = false;
    const int smem = EPI == 0 ? (2 * g8::BUF > 8 * 16 * 68 * 4 ? 2 * g8::BUF : 8 * 16 * 68 * 4) : 2 * g8::BUF;
    if (!attr) { CK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI, PERSIST, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); attr = true; }
    const int ntm = M / 256, ntn = N / 256;
    const int grid = PERSIST ? std::min(256, ntm * ntn) : ntm * ntn;
    hipLaunchKernelGGL((gemm8_kernel<EPI, PERSIST, OPT>), dim3(grid), dim3(512), smem, st, A, lda, W, ldw, bias, out, ldo, M, N, K, ntm, ntn);
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int MMAX = 131072, NMAX = 4096, KMAX = 4096;
    half_t* A = (half_t*)dev_half((size_t)MMAX * 2048, 1);
    half_t* W = (half_t*)dev_half((size_t)NMAX * KMAX, 2);
    std::vector<float> hb(NMAX);
    for (int i = 0; i < NMAX; ++i) hb[i] = 0.01f * (float)((i * 37) % 101 - 50);
    float* bias = nullptr;
    CK(hipMalloc(&bias, NMAX * 4));
    CK(hipMemcpy(bias, hb.data(), NMAX * 4, hipMemcpyHostToDevice));
    half_t *out = nullptr, *ref = nullptr;
    CK(hipMalloc(&out, (size_t)MMAX * 2048 * 2));
    CK(hipMalloc(&ref, (size_t)2048 * 1024 * 2));

    typedef void (*launch_fn)(const half_t*, long, const half_t*, long, const float*, half_t*, long, int, int, int, hipStream_t);
    struct Var { const char* name; launch_fn fn; };
    const Var vars[] = {{"8ph lds-epi            ", launch<0, 0, 0>}, {"8ph lds-epi  buf      ", launch<0, 0, 2>}, {"8ph lds-epi  reread-b0", launch<0, 0, 4>},
                        {"8ph direct             ", launch<1, 0, 0>}, {"8ph persist            ", launch<1, 1, 0>},
                        {"8ph persist skew       ", launch<1, 1, 1>}, {"8ph persist buf        ", launch<1, 1, 2>}, {"8ph persist skew buf   ", launch<1, 1, 3>}};
    auto is_persist = [&](launch_fn f) { return f == launch<1, 1, 0> || f == launch<1, 1, 1> || f == launch<1, 1, 2> || f == launch<1, 1, 3>; };

    {
        unsigned* dump = nullptr;
        CK(hipMalloc(&dump, 1024));
        hipLaunchKernelGGL(oob_probe_kernel, dim3(1), dim3(64), 4096, st, A, 4096, dump);
        unsigned h[256];
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h, dump, 1024, hipMemcpyDeviceToHost));
        int zeros = 0, untouched = 0, other = 0;
        for (int i = 128; i < 256; ++i) { if (h[i] == 0) ++zeros; else if (h[i] == 0xdeadbeefu) ++untouched; else ++other; }
        say("out-of-range buffer_load lds lanes: %d dwords zero, %d untouched, %d other (of 128); in-range dword 0 = %08x\n", zeros, untouched, other, h[0]);
    }

    // ---- correctness: every variant against the naive kernel, K odd / even multiples of 64, several tiles per workgroup ----
    say("correctness (max |diff| vs naive fp32-accumulate kernel, fp16 outputs)\n");
    {
        struct C { int M, N, K; };
        for (C c : {C{256, 256, 64}, C{512, 512, 128}, C{512, 256, 320}, C{1024, 768, 192}, C{2048, 1024, 512}}) {
            hipLaunchKernelGGL(ref_kernel, dim3(c.N / 64, c.M / 4), dim3(256), 0, st, A, (long)c.K, W, (long)c.K, bias, ref, (long)c.N, c.M, c.N, c.K);
            std::vector<uint16_t> hr((size_t)c.M * c.N), ho((size_t)c.M * c.N);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hr.data(), ref, hr.size() * 2, hipMemcpyDeviceToHost));
            for (const Var& v : vars) {
                if (is_persist(v.fn) && c.K < 128) continue;       // the persistent form needs two k-tiles per tile
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(out, 0xff, (size_t)c.M * c.N * 2, st));
                    v.fn(A, c.K, W, c.K, bias, out, c.N, c.M, c.N, c.K, st);
                    CK(hipStreamSynchronize(st));
                    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
                    double worst = 0;
                    size_t bad = 0;
                    for (size_t i = 0; i < hr.size(); ++i) {
                        const float a = (float)*(const half_t*)&hr[i], b = (float)*(const half_t*)&ho[i];
                        const double d = std::fabs((double)a - (double)b);
                        if (!(d <= 0.02 * (1.0 + std::fabs(a)))) ++bad;
                        if (d > worst || d != d) worst = d;
                    }
                    say("  M%5d N%5d K%4d  %s run %d  worst %.4g  bad %zu\n", c.M, c.N, c.K, v.name, rep, worst, bad);
                }
            }
        }
    }

    // ---- timing ----
    struct S { int M, N, K; };
    say("\ntiming (median of 9, random fp16 operands)\n");
    auto product = [&](const S& s, const char* force) {
        if (force) setenv("CTRL_IGEMM_FORCE", force, 1); else unsetenv("CTRL_IGEMM_FORCE");
        ctrl_igemm_desc d;
        memset(&d, 0, sizeof d);
        d.A = A; d.lda = s.K; d.mode = 0; d.Cin = s.K; d.taps = 1;
        d.Hin = d.Win = d.Hout = d.Wout = d.stride = d.up = 1;
        d.W = W; d.M = s.M; d.Nout = s.N; d.Ktot = s.K; d.rows_per_img = 1; d.scale = 1.f; d.bias = bias;
        d.nseg = 1;
        d.seg[0].out = out; d.seg[0].ld = s.N; d.seg[0].ncols = s.N; d.seg[0].dtype = CTRL_F16; d.seg[0].L = 1;
        std::vector<float> t;
        for (int i = 0; i < 2; ++i)
            if (ctrl_op_igemm(&d, st) != 0) { say("  product launch failed: %s\n", ctrl_last_error()); return; }
        for (int i = 0; i < 9; ++i) {
            CK(hipEventRecord(e0, st));
            ctrl_op_igemm(&d, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        say("  M%6d N%5d K%4d  product %-12s  %.4f ms  %.0f TFLOP/s\n", s.M, s.N, s.K, force ? force : "default", t[4], 2.0 * s.M * s.N * s.K / t[4] * 1e-9);
        unsetenv("CTRL_IGEMM_FORCE");
    };
    for (S s : {S{4096, 4096, 4096}, S{131072, 2048, 256}, S{131072, 2048, 512}, S{131072, 2048, 1024}, S{131072, 2048, 2048}, S{131072, 512, 2048},
                S{131072, 512, 320}, S{131072, 1024, 512}, S{32768, 2048, 512}, S{32768, 512, 2048}, S{8192, 1280, 4096}}) {
        product(s, nullptr);
        product(s, "256x128");
        for (const Var& v : vars) {
            if (is_persist(v.fn) && s.K < 128) continue;
            std::vector<float> t;
            for (int i = 0; i < 2; ++i) v.fn(A, s.K, W, s.K, bias, out, s.N, s.M, s.N, s.K, st);
            for (int i = 0; i < 9; ++i) {
                CK(hipEventRecord(e0, st));
                v.fn(A, s.K, W, s.K, bias, out, s.N, s.M, s.N, s.K, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            say("  M%6d N%5d K%4d  %s  %.4f ms  %.0f TFLOP/s\n", s.M, s.N, s.K, v.name, t[4], 2.0 * s.M * s.N * s.K / t[4] * 1e-9);
        }
    }
    say("done\n");
    return 0;
}
