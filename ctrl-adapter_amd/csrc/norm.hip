// GroupNorm / LayerNorm for channels-last fp16 activations, fp32 statistics (gfx950).
//
// Reference ops: torch.nn.GroupNorm(32, C) in ResnetBlock2D (model/resnet_block_2d.py:116,128),
// TemporalResnetBlock, Transformer2DModel.norm and AdapterSpatioTemporal.norm
// (model/adapter_spatial_temporal.py:61); torch.nn.LayerNorm in (Temporal)BasicTransformerBlock.
//
// These kernels are HBM-bound: each reads its input once with 16-byte vector loads and reduces with
// wavefront shuffles / LDS.  GroupNorm is split into a statistics pass (sum, sum of squares per (image, group),
// reduced in a fixed order: no floating-point atomics anywhere, so results are bit-reproducible) and an apply pass
// (normalise + affine + optional SiLU) because one group of an SDXL-sized map (10 ch x 128x128) does not fit one
// workgroup.
#include "ops.h"
#include <cstdlib>

namespace {

// 8 consecutive channels as fp32, from the fp16 activations or the fp32 residual stream
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float (&o)[8]) {
    const h8 v = *(const h8*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
    const f4 a = *(const f4*)p, b = *(const f4*)(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}

// x [imgs][rows][C]; grid (chunks, imgs); each thread owns an 8-channel chunk and strides over rows, four to eight
// 16-byte loads in flight per thread (these kernels are pure HBM streaming: memory-level parallelism is what matters).
// The reduction is ORDER-FIXED so a forward is bit-reproducible run to run: per-thread partials go through LDS slots
// (no LDS atomics), each (image, chunk) workgroup stores its 2*G partial sums, and the last workgroup of an image to
// arrive (ticket counter) adds the chunks in index order.  stats layout: [imgs][G][2] results | [imgs] tickets (zero on
// entry, reset to zero on exit) | [imgs][chunks][G][2] partials.
// Grouped launches (ops.h: OpCollector): up to four problems of one shape share a launch, problem = blockIdx.z; their pointers
// travel as plain kernel arguments and are selected value by value (scalar selects).
template <typename P> struct Ptr4 { P p[4]; };
template <typename P> __device__ __forceinline__ P pick4(const Ptr4<P>& q, int z) {
    P a = q.p[0], b = q.p[1], c = q.p[2], d = q.p[3];
    return z == 0 ? a : (z == 1 ? b : (z == 2 ? c : d));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(Ptr4<const T*> xs, Ptr4<float*> statss,
                                                       int rows, int C, int G, int rows_per_block) {
    extern __shared__ float sh[];   // [rpi][2*C]: per (row lane, channel) sum | sumsq
    __shared__ int is_last;
    const T* __restrict__ x = pick4(xs, blockIdx.z);
    float* __restrict__ stats = pick4(statss, blockIdx.z);
    const int tid = threadIdx.x;
    const int lpr = C >> 3;                 // lanes per row
    const int rpi = 256 / lpr;              // rows per iteration
    const int tr = tid / lpr, tc = tid - tr * lpr;
    const int img = blockIdx.y, imgs = gridDim.y, chunks = gridDim.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
    if (tr < rpi) {
        const T* xp = x + ((size_t)img * rows) * C + tc * 8;
        constexpr int U = sizeof(T) == 2 ? 8 : 4;       // loads in flight per thread
        int r = r0 + tr;
        for (; r + (U - 1) * rpi < r1; r += U * rpi) {
            float v[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) load8<T>(xp + (size_t)(r + u * rpi) * C, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = v[u][j]; s[j] += f; ss[j] += f * f; }
        }
        for (; r < r1; r += rpi) {
            float v[8];
            load8<T>(xp + (size_t)r * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = v[j]; s[j] += f; ss[j] += f * f; }
        }
        float* row = sh + (size_t)tr * 2 * C;
        *(f4*)(row + tc * 8) = f4{s[0], s[1], s[2], s[3]};
        *(f4*)(row + tc * 8 + 4) = f4{s[4], s[5], s[6], s[7]};
        *(f4*)(row + C + tc * 8) = f4{ss[0], ss[1], ss[2], ss[3]};
        *(f4*)(row + C + tc * 8 + 4) = f4{ss[4], ss[5], ss[6], ss[7]};
    }
    __syncthreads();
    // group g = tid / 8 is summed by 8 lanes (fixed element -> lane assignment), then a fixed xor tree
    const int cg = C / G;
    const int g = tid >> 3, part = tid & 7;
    float a = 0.f, b = 0.f;
    if (g < G) {
        const int n = cg * rpi;
        for (int e = part; e < n; e += 8) {
            const int t = e / cg, c = g * cg + (e - t * cg);
            a += sh[(size_t)t * 2 * C + c];
            b += sh[(size_t)t * 2 * C + C + c];
        }
    }
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    float* res = stats + (size_t)img * G * 2;
    unsigned* ticket = (unsigned*)(stats + (size_t)imgs * G * 2) + img;
    float* part_all = stats + (size_t)imgs * G * 2 + imgs + (size_t)img * chunks * G * 2;
    if (chunks == 1) {
        if (g < G && part == 0) { res[g * 2] = a; res[g * 2 + 1] = b; }
        return;
    }
    // Publication without a cache-wide release fence (a `__threadfence()` per workgroup costs an L2 write-back each):
    // the partials are agent-scope atomic stores (written through to the coherence point), the wave waits for their
    // acknowledgement, and only then is the ticket taken; the last workgroup reads them back with agent-scope loads.
    if (g < G && part == 0) {
        float* pp = part_all + (size_t)blockIdx.x * G * 2;
        __hip_atomic_store(pp + g * 2, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + g * 2 + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): stores acknowledged
    __syncthreads();
    if (tid == 0)
        is_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(chunks - 1));
    __syncthreads();
    if (!is_last) return;
    // The last workgroup adds the chunks of its image.  Round 5: by all four waves -- wave q owns the q-th quarter of the chunk list
    // (index order inside, 8 loads in flight), the four quarter sums are added in quarter order: still one fixed order whoever
    // arrives last.  (One wave walking 96 chunks of a 128^2 map was 12 dependent round trips to the coherence point, ~20 us of the
    // 76 us launch: the statistics pass ran at 2.2 TB/s of its read while the apply pass streams at 4.2.)
    {
        const int v = tid & 63, q = tid >> 6;
        const int cq = (chunks + 3) >> 2;
        const int k0 = q * cq, k1 = min(chunks, k0 + cq);
        float acc = 0.f;
        if (v < 2 * G) {
            const float* pv = part_all + v;
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                float u8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    u8[u] = __hip_atomic_load(pv + (size_t)(k + u) * G * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += u8[u];
            }
            for (; k < k1; ++k) acc += __hip_atomic_load(pv + (size_t)k * G * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sh[q * 64 + v] = acc;            // (the per-row partial area of `sh` is dead: every thread passed the barriers above)
        __syncthreads();
        if (tid < 2 * G) res[tid] = ((sh[tid] + sh[64 + tid]) + sh[128 + tid]) + sh[192 + tid];
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this buffer
}

// same decomposition as the statistics pass: a thread folds mean / rstd / gamma / beta of its 8 channels into one
// (scale, shift) pair each, once, then streams its rows: y = x*scale + shift (optional SiLU)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(Ptr4<const T*> xs, Ptr4<const float*> statss,
                                                       Ptr4<const float*> gammas, Ptr4<const float*> betas,
                                                       Ptr4<half_t*> ys, int rows, int C, int G, float eps,
                                                       int silu, int rows_per_block, long ldy, int lo_off,
                                                       float stat_rows, long y_img_rows, long y_row0) {
    const T* __restrict__ x = pick4(xs, blockIdx.z);
    const float* __restrict__ stats = pick4(statss, blockIdx.z);
    const float* __restrict__ gamma = pick4(gammas, blockIdx.z);
    const float* __restrict__ beta = pick4(betas, blockIdx.z);
    half_t* __restrict__ y = pick4(ys, blockIdx.z);
    const int tid = threadIdx.x;
    const int lpr = C >> 3;
    const int rpi = 256 / lpr;
    const int tr = tid / lpr, tc = tid - tr * lpr;
    if (tr >= rpi) return;
    const int img = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const int cg = C / G;
    const float inv_cnt = 1.0f / (stat_rows * (float)cg);     // rows the statistics span (>= rows when frames are sharded)
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = tc * 8 + j;
        const int g = c / cg;
        const float s = stats[((size_t)img * G + g) * 2], ss = stats[((size_t)img * G + g) * 2 + 1];
        const float mean = s * inv_cnt;
        const float var = fmaxf(ss * inv_cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        sc[j] = rstd * gamma[c];
        sf[j] = beta[c] - mean * sc[j];
    }
    const T* xp = x + ((size_t)img * rows) * C + tc * 8;
    half_t* yp = y + ((size_t)img * y_img_rows + y_row0) * ldy + tc * 8;
    // split operand (lo_off > 0): the row also carries lo = fp16(f - hi), so that hi + lo == f to ~2^-22
    auto emit = [&](const float (&v)[8], half_t* dst) {
        h8 o, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = v[j] * sc[j] + sf[j];
            if (silu) f = silu_f(f);
            o[j] = (half_t)f;
            l[j] = (half_t)(f - (float)o[j]);
        }
        *(h8*)dst = o;
        if (lo_off) *(h8*)(dst + lo_off) = l;
    };
    int r = r0 + tr;
    for (; r + 3 * rpi < r1; r += 4 * rpi) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<T>(xp + (size_t)(r + u * rpi) * C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) emit(v[u], yp + (size_t)(r + u * rpi) * ldy);
    }
    for (; r < r1; r += rpi) {
        float v[8];
        load8<T>(xp + (size_t)r * C, v);
        emit(v, yp + (size_t)r * ldy);
    }
}

// GroupNorm of a SMALL map in one launch (round 5): a workgroup owns 80 channels (8 / 4 / 2 whole groups at C = 320 / 640 / 1280) of one
// image -- statistics over its rows (per-thread partials through LDS slots, a fixed order: bit-reproducible), then the apply pass over the
// same rows, which still sit in L2.  The two-kernel form above costs the low-resolution levels of the ControlNet chain two launches of
// ~10 us for a few hundred KB; it remains the form for every map whose 80-channel slice exceeds kGnFusedBytes.
constexpr int kGnFusedCh = 80;
constexpr int kGnFusedRL = 102;          // row lanes: 1020 of the 1024 threads, 10 lanes of 8 channels per row
// (1024 threads: a workgroup streams its slice alone, so its memory-level parallelism is what a pass costs -- 102 rows x 4 loads in
// flight per iteration; the first 256-thread form spent ~20 us of dependent round trips on a 32^2 map)
template <typename T>
__global__ __launch_bounds__(1024) void gn_fused_kernel(Ptr4<const T*> xs, Ptr4<const float*> gammas, Ptr4<const float*> betas, Ptr4<half_t*> ys,
                                                        int rows, int C, int cg, float eps, int silu, long ldy, int lo_off) {
    __shared__ float sh[kGnFusedRL][2 * kGnFusedCh];      // per (row lane, channel): sum | sum of squares
    __shared__ float gst[8][2];                           // per group of the block: mean, rstd
    const T* __restrict__ x = pick4(xs, blockIdx.z);
    const float* __restrict__ gamma = pick4(gammas, blockIdx.z);
    const float* __restrict__ beta = pick4(betas, blockIdx.z);
    half_t* __restrict__ y = pick4(ys, blockIdx.z);
    const int tid = threadIdx.x;
    const int tr = tid / 10, tc = tid - tr * 10;
    constexpr int RL = kGnFusedRL;
    const int img = blockIdx.y, c0 = blockIdx.x * kGnFusedCh;
    const T* xp = x + ((size_t)img * rows) * C + c0 + tc * 8;
    if (tr < RL) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
        int r = tr;
        for (; r + 3 * RL < rows; r += 4 * RL) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8<T>(xp + (size_t)(r + u * RL) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = v[u][j]; s[j] += f; ss[j] += f * f; }
        }
        for (; r < rows; r += RL) {
            float v[8];
            load8<T>(xp + (size_t)r * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = v[j]; s[j] += f; ss[j] += f * f; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { sh[tr][tc * 8 + j] = s[j]; sh[tr][kGnFusedCh + tc * 8 + j] = ss[j]; }
    }
    __syncthreads();
    {
        // group g of the block = 16 lanes, fixed element -> lane assignment, fixed xor tree
        const int g = tid >> 4, part = tid & 15, ngrp = kGnFusedCh / cg;
        float a = 0.f, b = 0.f;
        if (g < ngrp) {
            const int n = cg * RL;
            for (int e = part; e < n; e += 16) {
                const int t = e / cg, c = g * cg + (e - t * cg);
                a += sh[t][c];
                b += sh[t][kGnFusedCh + c];
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (g < ngrp && part == 0) {
            const float inv_cnt = 1.0f / ((float)rows * (float)cg);
            const float mean = a * inv_cnt;
            const float var = fmaxf(b * inv_cnt - mean * mean, 0.f);
            gst[g][0] = mean;
            gst[g][1] = rsqrtf(var + eps);
        }
    }
    __syncthreads();
    if (tr >= RL) return;
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = tc * 8 + j, g = cl / cg;
        sc[j] = gst[g][1] * gamma[c0 + cl];
        sf[j] = beta[c0 + cl] - gst[g][0] * sc[j];
    }
    half_t* yp = y + ((size_t)img * rows) * ldy + c0 + tc * 8;
    auto emit = [&](const float (&v)[8], half_t* dst) {
        h8 o, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = v[j] * sc[j] + sf[j];
            if (silu) f = silu_f(f);
            o[j] = (half_t)f;
            l[j] = (half_t)(f - (float)o[j]);
        }
        *(h8*)dst = o;
        if (lo_off) *(h8*)(dst + lo_off) = l;
    };
    int r = tr;
    for (; r + 3 * RL < rows; r += 4 * RL) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<T>(xp + (size_t)(r + u * RL) * C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) emit(v[u], yp + (size_t)(r + u * RL) * ldy);
    }
    for (; r < rows; r += RL) {
        float v[8];
        load8<T>(xp + (size_t)r * C, v);
        emit(v, yp + (size_t)r * ldy);
    }
}

// one wavefront per row; C <= 64*8*NCH
// addv (optional): a per-image fp32 vector added to the row BEFORE the norm, addv[((row / rows_per_img) % vmod) * ldv + c] (the
// frame-index embedding in front of the temporal transformer, model/adapter_spatial_temporal.py:279); the sum is also
// written out (xsum, in x's dtype) because it is the block's residual stream -- one pass instead of add_rowvec + layernorm
template <int NCH, typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(Ptr4<const T*> xs, long ldx,
                                                        Ptr4<const float*> gammas, Ptr4<const float*> betas,
                                                        Ptr4<half_t*> ys, long ldy, int M, int C, float eps,
                                                        Ptr4<const float*> addvs, long ldv, int rows_per_img, int vmod,
                                                        Ptr4<T*> xsums) {
    const T* __restrict__ x = pick4(xs, blockIdx.z);
    const float* __restrict__ gamma = pick4(gammas, blockIdx.z);
    const float* __restrict__ beta = pick4(betas, blockIdx.z);
    half_t* __restrict__ y = pick4(ys, blockIdx.z);
    const float* __restrict__ addv = pick4(addvs, blockIdx.z);
    T* __restrict__ xsum = pick4(xsums, blockIdx.z);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const T* xp = x + (size_t)row * ldx;
    const float* av = addv ? addv + (size_t)((row / rows_per_img) % vmod) * ldv : nullptr;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = (lane + i * 64) * 8;
        if (c0 < C) {
            load8<T>(xp + c0, v[i]);
            if (av) {
                const f4 a0 = *(const f4*)(av + c0), a1 = *(const f4*)(av + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][j] += a0[j]; v[i][4 + j] += a1[j]; }
                T* sp = xsum + (size_t)row * ldx + c0;
                if constexpr (sizeof(T) == 4) {
                    *(f4*)sp = f4{v[i][0], v[i][1], v[i][2], v[i][3]};
                    *(f4*)(sp + 4) = f4{v[i][4], v[i][5], v[i][6], v[i][7]};
                } else {
                    h8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o[j] = (half_t)v[i][j]; v[i][j] = (float)o[j]; }     // the norm sees what is stored
                    *(h8*)sp = o;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = (lane + i * 64) * 8;
        if (c0 < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    half_t* yp = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = (lane + i * 64) * 8;
        if (c0 < C) {
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[i][j] - mean) * rstd * gamma[c0 + j] + beta[c0 + j]);
            *(h8*)(yp + c0) = o;
        }
    }
}

}  // namespace

static int gn_rows_per_block(int imgs, int rows_per_img, int C, int target_blocks) {
    // aim for ~target_blocks workgroups in total with at least a few iterations of the 4x-unrolled loop each
    const int rpi = 256 / (C / 8);
    int chunks = (target_blocks + imgs - 1) / imgs;
    int rpb = (rows_per_img + chunks - 1) / chunks;
    if (rpb < 8 * rpi) rpb = 8 * rpi;
    rpb = ((rpb + rpi - 1) / rpi) * rpi;
    return rpb;
}

size_t op_gn_stats_floats(int imgs, int rows_per_img, int C, int G) {
    if (imgs < 1 || rows_per_img < 1 || C < 8 || G < 1) return 0;
    const int rows_per_block = gn_rows_per_block(imgs, rows_per_img, C, 768);
    const int chunks = (rows_per_img + rows_per_block - 1) / rows_per_block;
    return (size_t)imgs * G * 2 + imgs + (chunks > 1 ? (size_t)imgs * chunks * G * 2 : 0);
}

template <typename P, typename A, typename F> static Ptr4<P> ptrs_of(const A* a, int n, F f) {
    Ptr4<P> q;
    for (int i = 0; i < 4; ++i) q.p[i] = (P)f(a[i < n ? i : 0]);
    return q;
}
static int pk16(const void* p) { return p ? 1 + (int)((uintptr_t)p & 15) : 0; }

int op_gn_stats(const void* x, int x_dtype, float* stats, int imgs, int rows_per_img, int C, int G, hipStream_t s) {
    const GnStatsArgs a = {x, x_dtype, stats, imgs, rows_per_img, C, G};
    if (t_collect) {
        int rc = 0;
        const int i = t_collect->slot(OpCollector::GN_STATS, s, &rc);
        if (i < 0) return rc;
        t_collect->gs[i] = a;
        return 0;
    }
    return op_gn_stats_group(&a, 1, s);
}

int op_gn_stats_group(const GnStatsArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxGroup, "gn_stats_group: 1..4 problems");
    bool same = group_launches_enabled();
    for (int i = 1; i < n; ++i)
        same = same && a[i].x_dtype == a[0].x_dtype && a[i].imgs == a[0].imgs && a[i].rows_per_img == a[0].rows_per_img && a[i].C == a[0].C && a[i].G == a[0].G;
    if (n > 1 && !same) {
        for (int i = 0; i < n; ++i) TRY(op_gn_stats_group(a + i, 1, s));
        return 0;
    }
    const int x_dtype = a[0].x_dtype, imgs = a[0].imgs, rows_per_img = a[0].rows_per_img, C = a[0].C, G = a[0].G;
    CTRL_CHECK(x_dtype == DT_F16 || x_dtype == DT_F32, "gn_stats: input must be fp16 or fp32");
    CTRL_CHECK(C % 8 == 0 && C % G == 0 && C / 8 <= 256, "gn_stats: C must be a multiple of 8 and of G, <= 2048");
    CTRL_CHECK(G <= 32, "gn_stats: at most 32 groups");
    // fewer, fatter workgroups than the apply pass: the last workgroup of an image adds all its chunks serially
    const int rows_per_block = gn_rows_per_block(imgs, rows_per_img, C, 768);
    const int chunks = (rows_per_img + rows_per_block - 1) / rows_per_block;
    PROF_WORK(0, n * (x_dtype == DT_F32 ? 4.0 : 2.0) * imgs * rows_per_img * C);
    if (n > 1) prof_detail("x%d", n);
    const size_t lds = (size_t)(256 / (C / 8)) * 2 * C * sizeof(float);
    const auto st = ptrs_of<float*>(a, n, [](const GnStatsArgs& q) { return q.stats; });
    if (x_dtype == DT_F32)
        LAUNCH("gn_stats", gn_stats_kernel<float>, dim3(chunks, imgs, n), dim3(256), lds, s,
               ptrs_of<const float*>(a, n, [](const GnStatsArgs& q) { return q.x; }), st, rows_per_img, C, G, rows_per_block);
    else
        LAUNCH("gn_stats", gn_stats_kernel<half_t>, dim3(chunks, imgs, n), dim3(256), lds, s,
               ptrs_of<const half_t*>(a, n, [](const GnStatsArgs& q) { return q.x; }), st, rows_per_img, C, G, rows_per_block);
    return 0;
}

int op_gn_apply(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, half_t* y,
                int imgs, int rows_per_img, int C, int G, float eps, int silu, hipStream_t s, long ldy, int lo_off,
                long stat_rows, long y_img_rows, long y_row0) {
    const GnApplyArgs a = {x, x_dtype, stats, gamma, beta, y, imgs, rows_per_img, C, G, eps, silu, ldy, lo_off, stat_rows, y_img_rows, y_row0};
    if (t_collect) {
        int rc = 0;
        const int i = t_collect->slot(OpCollector::GN_APPLY, s, &rc);
        if (i < 0) return rc;
        t_collect->ga[i] = a;
        return 0;
    }
    return op_gn_apply_group(&a, 1, s);
}

int op_gn_apply_group(const GnApplyArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxGroup, "gn_apply_group: 1..4 problems");
    bool same = group_launches_enabled();
    for (int i = 1; i < n; ++i)
        same = same && a[i].x_dtype == a[0].x_dtype && a[i].imgs == a[0].imgs && a[i].rows_per_img == a[0].rows_per_img && a[i].C == a[0].C &&
               a[i].G == a[0].G && a[i].eps == a[0].eps && a[i].silu == a[0].silu && a[i].ldy == a[0].ldy && a[i].lo_off == a[0].lo_off &&
               a[i].stat_rows == a[0].stat_rows && a[i].y_img_rows == a[0].y_img_rows && a[i].y_row0 == a[0].y_row0;
    if (n > 1 && !same) {
        for (int i = 0; i < n; ++i) TRY(op_gn_apply_group(a + i, 1, s));
        return 0;
    }
    const int x_dtype = a[0].x_dtype, imgs = a[0].imgs, rows_per_img = a[0].rows_per_img, C = a[0].C, G = a[0].G, lo_off = a[0].lo_off;
    long ldy = a[0].ldy, stat_rows = a[0].stat_rows, y_img_rows = a[0].y_img_rows;
    const long y_row0 = a[0].y_row0;
    CTRL_CHECK(C % 8 == 0 && C % G == 0 && C / 8 <= 256, "gn_apply: C must be a multiple of 8 and of G, <= 2048");
    if (ldy == 0) ldy = C;
    if (stat_rows <= 0) stat_rows = rows_per_img;
    if (y_img_rows <= 0) y_img_rows = rows_per_img;
    CTRL_CHECK(y_row0 >= 0 && y_row0 + rows_per_img <= y_img_rows, "gn_apply: bad output image layout");
    CTRL_CHECK(ldy % 8 == 0 && lo_off % 8 == 0 && (lo_off == 0 || (lo_off >= C && lo_off + C <= ldy)), "gn_apply: bad split layout");
    const int rows_per_block = gn_rows_per_block(imgs, rows_per_img, C, 2048);
    const int chunks = (rows_per_img + rows_per_block - 1) / rows_per_block;
    CTRL_CHECK(x_dtype == DT_F16 || x_dtype == DT_F32, "gn_apply: input must be fp16 or fp32");
    PROF_WORK(0, n * ((x_dtype == DT_F32 ? 6.0 : 4.0) + (lo_off ? 2.0 : 0.0)) * imgs * rows_per_img * C);
    if (n > 1) prof_detail("x%d", n);
    const auto st = ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.stats; });
    const auto ga = ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.gamma; });
    const auto be = ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.beta; });
    const auto ys = ptrs_of<half_t*>(a, n, [](const GnApplyArgs& q) { return q.y; });
    if (x_dtype == DT_F32)
        LAUNCH("gn_apply", gn_apply_kernel<float>, dim3(chunks, imgs, n), dim3(256), 0, s,
               ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.x; }), st, ga, be, ys, rows_per_img, C, G, a[0].eps, a[0].silu,
               rows_per_block, ldy, lo_off, (float)stat_rows, y_img_rows, y_row0);
    else
        LAUNCH("gn_apply", gn_apply_kernel<half_t>, dim3(chunks, imgs, n), dim3(256), 0, s,
               ptrs_of<const half_t*>(a, n, [](const GnApplyArgs& q) { return q.x; }), st, ga, be, ys, rows_per_img, C, G, a[0].eps, a[0].silu,
               rows_per_block, ldy, lo_off, (float)stat_rows, y_img_rows, y_row0);
    return 0;
}

int op_layernorm(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, half_t* y, long ldy,
                 int M, int C, float eps, hipStream_t s) {
    return op_layernorm_add(x, x_dtype, ldx, nullptr, 0, 1, 1, nullptr, gamma, beta, y, ldy, M, C, eps, s);
}

int op_layernorm_add(const void* x, int x_dtype, long ldx, const float* addv, long ldv, int rows_per_img, int vmod, void* xsum,
                     const float* gamma, const float* beta, half_t* y, long ldy, int M, int C, float eps, hipStream_t s) {
    const LnArgs a = {x, x_dtype, ldx, addv, ldv, rows_per_img, vmod, xsum, gamma, beta, y, ldy, M, C, eps};
    if (t_collect) {
        int rc = 0;
        const int i = t_collect->slot(OpCollector::LAYERNORM, s, &rc);
        if (i < 0) return rc;
        t_collect->ln[i] = a;
        return 0;
    }
    return op_layernorm_group(&a, 1, s);
}

int op_layernorm_group(const LnArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxGroup, "layernorm_group: 1..4 problems");
    bool same = group_launches_enabled();
    for (int i = 1; i < n; ++i)
        same = same && a[i].x_dtype == a[0].x_dtype && a[i].ldx == a[0].ldx && a[i].ldv == a[0].ldv && a[i].rows_per_img == a[0].rows_per_img &&
               a[i].vmod == a[0].vmod && a[i].ldy == a[0].ldy && a[i].M == a[0].M && a[i].C == a[0].C && a[i].eps == a[0].eps &&
               pk16(a[i].addv) == pk16(a[0].addv) && pk16(a[i].xsum) == pk16(a[0].xsum);
    if (n > 1 && !same) {
        for (int i = 0; i < n; ++i) TRY(op_layernorm_group(a + i, 1, s));
        return 0;
    }
    const void* addv = a[0].addv; const void* xsum = a[0].xsum;
    const int x_dtype = a[0].x_dtype, M = a[0].M, C = a[0].C;
    const long ldx = a[0].ldx, ldy = a[0].ldy, ldv = a[0].ldv;
    int rows_per_img = a[0].rows_per_img, vmod = a[0].vmod;
    const float eps = a[0].eps;
    for (int i = 0; i < n; ++i)
        CTRL_CHECK(!a[i].addv || (a[i].xsum && rows_per_img > 0 && vmod > 0 && ldv % 4 == 0 && (((uintptr_t)a[i].addv | (uintptr_t)a[i].xsum) & 15) == 0),
                   "layernorm: the added vector needs an output for the sum, 16-byte aligned");
    (void)addv; (void)xsum;
    CTRL_CHECK(C % 8 == 0 && C <= 2048, "layernorm: C must be a multiple of 8 and <= 2048");
    CTRL_CHECK(ldx % 8 == 0 && ldy % 8 == 0, "layernorm: leading dims must be multiples of 8");
    CTRL_CHECK(x_dtype == DT_F16 || x_dtype == DT_F32, "layernorm: input must be fp16 or fp32");
    const dim3 grid((M + 3) / 4, 1, n), block(256);
    PROF_WORK(0, n * ((x_dtype == DT_F32 ? 6.0 : 4.0) + (a[0].addv ? (x_dtype == DT_F32 ? 4.0 : 2.0) : 0.0)) * M * C);
    if (n > 1) prof_detail("x%d", n);
    if (!rows_per_img) rows_per_img = 1;
    if (!vmod) vmod = 1;
    const auto ga = ptrs_of<const float*>(a, n, [](const LnArgs& q) { return q.gamma; });
    const auto be = ptrs_of<const float*>(a, n, [](const LnArgs& q) { return q.beta; });
    const auto ys = ptrs_of<half_t*>(a, n, [](const LnArgs& q) { return q.y; });
    const auto av = ptrs_of<const float*>(a, n, [](const LnArgs& q) { return q.addv; });
#define LN_LAUNCH(NCH)                                                                                              \
    do {                                                                                                            \
        if (x_dtype == DT_F32)                                                                                      \
            LAUNCH("layernorm", (layernorm_kernel<NCH, float>), grid, block, 0, s,                                   \
                   ptrs_of<const float*>(a, n, [](const LnArgs& q) { return q.x; }), ldx, ga, be, ys, ldy, M, C, eps, \
                   av, ldv, rows_per_img, vmod, ptrs_of<float*>(a, n, [](const LnArgs& q) { return q.xsum; }));     \
        else                                                                                                        \
            LAUNCH("layernorm", (layernorm_kernel<NCH, half_t>), grid, block, 0, s,                                  \
                   ptrs_of<const half_t*>(a, n, [](const LnArgs& q) { return q.x; }), ldx, ga, be, ys, ldy, M, C, eps, \
                   av, ldv, rows_per_img, vmod, ptrs_of<half_t*>(a, n, [](const LnArgs& q) { return q.xsum; }));    \
    } while (0)
    if (C <= 512) LN_LAUNCH(1);
    else if (C <= 1024) LN_LAUNCH(2);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    return 0;
}


// ---- GroupNorm of a small map in one launch (gn_fused_kernel) ----
constexpr size_t kGnFusedBytes = 512 * 1024;     // largest 80-channel slice of one image a workgroup takes (read twice: HBM, then L2)
// the shapes the fused kernel takes: GroupNorm(32) whose 80-channel blocks hold whole groups, slice of one image <= kGnFusedBytes
bool op_gn_fused_fits(int x_dtype, int rows_per_img, int C, int G) {
    if (G != 32 || C % kGnFusedCh != 0 || (C / G) <= 0 || kGnFusedCh % (C / G) != 0 || kGnFusedCh / (C / G) > 8) return false;
    return (size_t)rows_per_img * kGnFusedCh * (x_dtype == DT_F32 ? 4 : 2) <= kGnFusedBytes;
}
// whether the PLANS use it (run_groupnorm).  OFF by default: measured in the step (one gpurun call, profiles/r05_ab_same_box.txt) the
// fused form LOSES 0.5-0.6 ms per SDXL step -- 32 .. 128 workgroups stream a 10-20 MB map at a fraction of the chip's bandwidth (22-32 us
// per launch against 10 + 10 us for the two launches it replaces, which run on 768-2048 workgroups); it would pay only on the 8^2 level
// (~5 us per norm).  CTRL_GN_FUSED=1 switches it on; the op stays covered by tests/test_gpu_ops.py::test_groupnorm_fused_small_maps.
bool op_gn_fused_applies(int x_dtype, int rows_per_img, int C, int G) {
    return policy_is1(P_GN_FUSED) && op_gn_fused_fits(x_dtype, rows_per_img, C, G);
}

int op_gn_fused(const void* x, int x_dtype, const float* gamma, const float* beta, half_t* y, int imgs, int rows_per_img, int C, int G,
                float eps, int silu, hipStream_t s, long ldy, int lo_off) {
    const GnApplyArgs a = {x, x_dtype, nullptr, gamma, beta, y, imgs, rows_per_img, C, G, eps, silu, ldy, lo_off, 0, 0, 0};
    if (t_collect) {
        int rc = 0;
        const int i = t_collect->slot(OpCollector::GN_FUSED, s, &rc);
        if (i < 0) return rc;
        t_collect->ga[i] = a;
        return 0;
    }
    return op_gn_fused_group(&a, 1, s);
}

int op_gn_fused_group(const GnApplyArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxGroup, "gn_fused_group: 1..4 problems");
    bool same = group_launches_enabled();
    for (int i = 1; i < n; ++i)
        same = same && a[i].x_dtype == a[0].x_dtype && a[i].imgs == a[0].imgs && a[i].rows_per_img == a[0].rows_per_img && a[i].C == a[0].C &&
               a[i].G == a[0].G && a[i].eps == a[0].eps && a[i].silu == a[0].silu && a[i].ldy == a[0].ldy && a[i].lo_off == a[0].lo_off;
    if (n > 1 && !same) {
        for (int i = 0; i < n; ++i) TRY(op_gn_fused_group(a + i, 1, s));
        return 0;
    }
    const int x_dtype = a[0].x_dtype, imgs = a[0].imgs, rows = a[0].rows_per_img, C = a[0].C, G = a[0].G, lo_off = a[0].lo_off;
    const long ldy = a[0].ldy ? a[0].ldy : C;
    CTRL_CHECK(op_gn_fused_fits(x_dtype, rows, C, G), "gn_fused: not a small GroupNorm(32) map");
    CTRL_CHECK(x_dtype == DT_F16 || x_dtype == DT_F32, "gn_fused: input must be fp16 or fp32");
    CTRL_CHECK(ldy % 8 == 0 && lo_off % 8 == 0 && (lo_off == 0 || (lo_off >= C && lo_off + C <= ldy)), "gn_fused: bad split layout");
    PROF_WORK(0, n * ((x_dtype == DT_F32 ? 6.0 : 4.0) + (lo_off ? 2.0 : 0.0)) * imgs * rows * C);     // (the second read comes from L2)
    if (n > 1) prof_detail("x%d", n);
    const dim3 grid(C / kGnFusedCh, imgs, n);
    const auto ga = ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.gamma; });
    const auto be = ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.beta; });
    const auto ys = ptrs_of<half_t*>(a, n, [](const GnApplyArgs& q) { return q.y; });
    if (x_dtype == DT_F32)
        LAUNCH("gn_fused", gn_fused_kernel<float>, grid, dim3(1024), 0, s, ptrs_of<const float*>(a, n, [](const GnApplyArgs& q) { return q.x; }),
               ga, be, ys, rows, C, C / G, a[0].eps, a[0].silu, ldy, lo_off);
    else
        LAUNCH("gn_fused", gn_fused_kernel<half_t>, grid, dim3(1024), 0, s, ptrs_of<const half_t*>(a, n, [](const GnApplyArgs& q) { return q.x; }),
               ga, be, ys, rows, C, C / G, a[0].eps, a[0].silu, ldy, lo_off);
    return 0;
}
