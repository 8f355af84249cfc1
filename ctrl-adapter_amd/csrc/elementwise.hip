// Layout conversion, pooling, embeddings, small-M linears, blends, router softmax / merge and the
// load-time weight packers (gfx950).  All HBM- or latency-bound helper kernels of the hot path.
#include "ops.h"
#include <algorithm>

namespace {

// ---- [N][C][HW] any dtype -> [N][HW][C] fp16 through a 32(c) x 64(hw) LDS tile ----
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const void* __restrict__ x, int dt, half_t* __restrict__ y,
                                                           int C, int HW) {
    __shared__ float tile[32][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    {
        const int pl = t & 63;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cl = (t >> 6) + 4 * i;
            const int c = c0 + cl, p = p0 + pl;
            tile[cl][pl] = (c < C && p < HW) ? load_as_f32(x, ((size_t)n * C + c) * HW + p, dt) : 0.f;
        }
    }
    __syncthreads();
    {
        const int cl = t & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pl = (t >> 5) + 8 * i;
            const int c = c0 + cl, p = p0 + pl;
            if (c < C && p < HW) y[((size_t)n * HW + p) * C + c] = (half_t)tile[cl][pl];
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const half_t* __restrict__ x, void* __restrict__ y, int dt,
                                                           int C, int HW, float scale, const int* __restrict__ img_map) {
    __shared__ float tile[32][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    {
        const int cl = t & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pl = (t >> 5) + 8 * i;
            const int c = c0 + cl, p = p0 + pl;
            tile[cl][pl] = (c < C && p < HW) ? (float)x[((size_t)n * HW + p) * C + c] : 0.f;
        }
    }
    __syncthreads();
    {
        const int pl = t & 63;
        const int nd = img_map ? img_map[n] : n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cl = (t >> 6) + 4 * i;
            const int c = c0 + cl, p = p0 + pl;
            if (c < C && p < HW) store_from_f32(y, ((size_t)nd * C + c) * HW + p, dt, tile[cl][pl] * scale);
        }
    }
}

__global__ __launch_bounds__(256) void avgpool_kernel(const void* __restrict__ x, void* __restrict__ y, int dt,
                                                      int Hin, int Win, int Hout, int Wout, size_t total) {
    const int ky = Hin / Hout, kx = Win / Wout;
    const float inv = 1.0f / (float)(ky * kx);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % Wout);
        const size_t r = i / Wout;
        const int oy = (int)(r % Hout);
        const size_t nc = r / Hout;
        float s = 0.f;
        for (int a = 0; a < ky; ++a)
            for (int b = 0; b < kx; ++b) s += load_as_f32(x, (nc * Hin + oy * ky + a) * Win + ox * kx + b, dt);
        store_from_f32(y, i, dt, s * inv);
    }
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin],
// freq_k = exp(-ln(10000) * k / half)
__global__ void sincos_kernel(const float* __restrict__ t, int t_count, int Fmod, float* __restrict__ out, int N, int dim) {
    const int half_dim = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half_dim) return;
    const int n = i / half_dim, k = i - n * half_dim;
    const float tv = t ? t[t_count == 1 ? 0 : n] : (float)(n % Fmod);
    const float freq = expf(-9.210340371976184f * (float)k / (float)half_dim);
    const float arg = tv * freq;
    out[(size_t)n * dim + k] = cosf(arg);
    out[(size_t)n * dim + half_dim + k] = sinf(arg);
}

template <int MT>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, long ldx, const half_t* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ out, long ldo,
                                                           int M, int N, int K, int in_silu, int out_silu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int m0 = blockIdx.y * MT;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    const half_t* wp = w + (size_t)n * K;
    for (int k = lane * 8; k < K; k += 512) {
        const h8 wv = *(const h8*)(wp + k);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (m0 + i < M) {
                const float* xp = x + (size_t)(m0 + i) * ldx + k;
                const f4 a0 = *(const f4*)xp, a1 = *(const f4*)(xp + 4);
                float xv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xx = in_silu ? silu_f(xv[j]) : xv[j];
                    acc[i] += xx * (float)wv[j];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float r = wave_sum(acc[i]);
        if (lane == 0 && m0 + i < M) {
            float v = r + (b ? b[n] : 0.f);
            if (out_silu) v = silu_f(v);
            out[(size_t)(m0 + i) * ldo + n] = v;
        }
    }
}

// Grouped form: up to kSmallGroup independent small-M linears in ONE launch (block -> problem by a scan of the block
// prefix).  The adapter's time / frame-index embedding MLPs, per-layer time projections and single-key cross-attention
// vectors depend on (timestep, encoder states) only: ~10 such linears per adapter block, ~130 per video forward, are
// gathered into one launch per dependency level at the start of the forward (plan_adapter.cpp:precompute_small).
__global__ __launch_bounds__(256) void linear_small_group_kernel(SmallLinGroup g) {
    int p = 0;
#pragma unroll 1
    while (p + 1 < g.count && (int)blockIdx.x >= g.blk_begin[p + 1]) ++p;
    const SmallLin q = g.p[p];
    const int blk = blockIdx.x - g.blk_begin[p];
    const int nbn = (q.N + 3) >> 2;
    const int bm = blk / nbn, bn = blk - bm * nbn;
    constexpr int MT = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = bn * 4 + wave;
    if (n >= q.N) return;
    const int m0 = bm * MT;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    const half_t* wp = q.w + (size_t)n * q.K;
    for (int k = lane * 8; k < q.K; k += 512) {
        const h8 wv = *(const h8*)(wp + k);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (m0 + i < q.M) {
                const float* xp = q.x + (size_t)(m0 + i) * q.ldx + k;
                const f4 a0 = *(const f4*)xp, a1 = *(const f4*)(xp + 4);
                float xv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xx = q.in_silu ? silu_f(xv[j]) : xv[j];
                    acc[i] += xx * (float)wv[j];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float r = wave_sum(acc[i]);
        if (lane == 0 && m0 + i < q.M) {
            float v = r + (q.b ? q.b[n] : 0.f);
            if (q.out_silu) v = silu_f(v);
            q.out[(size_t)(m0 + i) * q.ldo + n] = v;
        }
    }
}

// 8-element chunk i of an fp16 or fp32 tensor (the fp32 residual stream) as floats, and back
__device__ __forceinline__ void ld8(const void* p, size_t i, int dt, float (&o)[8]) {
    if (dt == DT_F32) {
        const f4 a = ((const f4*)p)[2 * i], b = ((const f4*)p)[2 * i + 1];
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    } else {
        const h8 v = ((const h8*)p)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
    }
}
__device__ __forceinline__ void st8(void* p, size_t i, int dt, const float (&v)[8]) {
    if (dt == DT_F32) {
        ((f4*)p)[2 * i] = f4{v[0], v[1], v[2], v[3]};
        ((f4*)p)[2 * i + 1] = f4{v[4], v[5], v[6], v[7]};
    } else {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
        ((h8*)p)[i] = o;
    }
}

__global__ __launch_bounds__(256) void blend_kernel(const void* __restrict__ xs, int xs_dt, const void* __restrict__ xt, int xt_dt,
                                                    const float* __restrict__ mix, void* __restrict__ y, int y_dt, size_t nchunks) {
    const float alpha = 1.0f / (1.0f + __expf(-mix[0]));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
        float a[8], b[8], o[8];
        ld8(xs, i, xs_dt, a);
        ld8(xt, i, xt_dt, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = alpha * a[j] + (1.0f - alpha) * b[j];
        st8(y, i, y_dt, o);
    }
}

__global__ __launch_bounds__(256) void add_rowvec_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ v, long ldv,
                                                         void* __restrict__ y, int y_dt, size_t nchunks, int C, int rows_per_img, int vmod) {
    const int lpr = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
        const size_t row = i / lpr;
        const int c0 = (int)(i - row * lpr) * 8;
        const size_t img = (row / rows_per_img) % vmod;
        float a[8];
        ld8(x, i, x_dt, a);
        const float* vp = v + img * ldv + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += vp[j];
        st8(y, i, y_dt, a);
    }
}

// y = x + v[idx(row)], rows ordered (clip, frame, pixel) [(b f p)], idx = (b*L + p) % B: the reference hands the temporal
// transformer a time context ordered (pixel, clip) -- broadcast_to(h*w, batch, 1, C).reshape(h*w*batch, 1, C),
// model/adapter_spatial_temporal.py:246-249 -- while diffusers' TemporalBasicTransformerBlock orders its rows (clip, pixel):
// row i = b*L + p of the block attends context row i, which is clip i % B.  With one key per row the cross-attention term is
// that clip's to_out(to_v(context)) vector.
__global__ __launch_bounds__(256) void add_rowvec_clip_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ v, long ldv,
                                                              void* __restrict__ y, int y_dt, size_t nchunks, int C, int F, int L, int B) {
    const int lpr = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
        const size_t row = i / lpr;
        const int c0 = (int)(i - row * lpr) * 8;
        const size_t b = row / ((size_t)F * L), p = row % L;
        const size_t idx = (b * L + p) % B;
        float a[8];
        ld8(x, i, x_dt, a);
        const float* vp = v + idx * ldv + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += vp[j];
        st8(y, i, y_dt, a);
    }
}

struct MergeArgs {
    const void* x[8];
    int widx[8];
    int K;
};
__global__ __launch_bounds__(256) void merge_kernel(MergeArgs m, const float* __restrict__ w, void* __restrict__ out,
                                                    int dt, size_t n) {
    float wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wk[k] = (k < m.K) ? w[m.widx[k]] : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < m.K) s += wk[k] * load_as_f32(m.x[k], i, dt);
        store_from_f32(out, i, dt, s);
    }
}

struct RouterMask { int m[16]; };
// reference: model/ctrl_router.py:85-112 (logits = wg.weight[:,0] or zeros; -1e6 on masked experts; softmax)
__global__ void router_softmax_kernel(const float* __restrict__ wg, RouterMask mask, float* __restrict__ out,
                                      int R, int E, int equal_weights) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float lg[16];
    float mx = -3.0e38f;
    for (int e = 0; e < E; ++e) {
        float v = equal_weights ? 0.f : wg[(size_t)r * E + e];
        if (mask.m[e] == 0) v -= 1e6f;
        lg[e] = v;
        mx = fmaxf(mx, v);
    }
    float s = 0.f;
    for (int e = 0; e < E; ++e) { lg[e] = expf(lg[e] - mx); s += lg[e]; }
    for (int e = 0; e < E; ++e) out[(size_t)r * E + e] = lg[e] / s;
}

// nearest x2 up-sampling of a channels-last map (F.interpolate(scale_factor=2, mode="nearest"), the adapter's transformer-
// only SDXL blocks, model/adapter_spatial_temporal.py:235-237): y[n][2h+dy][2w+dx][c] = x[n][h][w][c], 16-byte vectors
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int H, int W, int C8, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        size_t r = i / C8;
        const int xo = (int)(r % (2 * W)); r /= (2 * W);
        const int yo = (int)(r % (2 * H));
        const size_t n = r / (2 * H);
        ((h8*)y)[i] = ((const h8*)x)[((n * H + (yo >> 1)) * W + (xo >> 1)) * C8 + c];
    }
}

// ---------------- weight packers (run once at plan build) ----------------
// fp16 range guard of the GEMM-weight packers: bf16 / fp32 checkpoints may hold values fp16 cannot (|w| > 65504)
__device__ __forceinline__ half_t to_h_checked(float v, int* ovf) {
    if (ovf && !(fabsf(v) <= 65504.f)) *ovf = 1;      // also catches inf / nan; benign race (all writers store 1)
    return (half_t)v;
}
__global__ void pack_conv_w_kernel(const void* __restrict__ w, int dt, half_t* __restrict__ out, int Cout, int Cin, int taps, int* ovf) {
    const size_t total = (size_t)Cout * taps * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        const size_t r = i / Cin;
        const int tp = (int)(r % taps);
        const size_t co = r / taps;
        out[i] = to_h_checked(load_as_f32(w, (co * Cin + ci) * taps + tp, dt), ovf);
    }
}
// split-operand convolution: [Cout][taps][2*Cin], out[co][tp][h*Cin + ci] = w[co][ci][tp] for h = 0, 1
__global__ void pack_conv_w_dup_kernel(const void* __restrict__ w, int dt, half_t* __restrict__ out, int Cout, int Cin, int taps, int* ovf) {
    const size_t total = (size_t)Cout * taps * 2 * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c2 = (int)(i % (2 * Cin));
        const int ci = c2 >= Cin ? c2 - Cin : c2;
        const size_t r = i / (2 * Cin);
        const int tp = (int)(r % taps);
        const size_t co = r / taps;
        out[i] = to_h_checked(load_as_f32(w, (co * Cin + ci) * taps + tp, dt), ovf);
    }
}
__global__ void pack_conv_w_direct_kernel(const void* __restrict__ w, int dt, float* __restrict__ out, int Cout, int Cin) {
    const size_t total = (size_t)9 * Cin * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const size_t r = i / Cout;
        const int ci = (int)(r % Cin);
        const int tp = (int)(r / Cin);
        out[i] = load_as_f32(w, ((size_t)co * Cin + ci) * 9 + tp, dt);
    }
}
// GEGLU interleave: packed row p -> source row:  blk = p/32, within = p%32;
//   within < 16 : hidden row  blk*16 + within ;  else : gate row  N/2 + blk*16 + within-16
__device__ __forceinline__ int geglu_src_row(int p, int N) {
    const int blk = p >> 5, wi = p & 31;
    return wi < 16 ? blk * 16 + wi : N / 2 + blk * 16 + (wi - 16);
}
__global__ void pack_linear_w_kernel(const void* __restrict__ w, int dt, half_t* __restrict__ out, int N, int K, int geglu, int* ovf) {
    const size_t total = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % K);
        const int p = (int)(i / K);
        const int srow = geglu ? geglu_src_row(p, N) : p;
        out[i] = to_h_checked(load_as_f32(w, (size_t)srow * K + k, dt), ovf);
    }
}
__global__ void pack_vec_kernel(const void* __restrict__ v, int dt, float* __restrict__ out, int N, int geglu) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    out[i] = load_as_f32(v, geglu ? geglu_src_row(i, N) : i, dt);
}

unsigned grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

int op_nchw_to_nhwc(const void* x, int dtype, half_t* y, int N, int C, int HW, hipStream_t s) {
    dim3 grid((HW + 63) / 64, (C + 31) / 32, N);
    PROF_WORK(0, (double)N * C * HW * ((dtype == DT_F32 ? 4.0 : 2.0) + 2.0));          // one read + one fp16 write
    LAUNCH("nchw_to_nhwc", nchw_to_nhwc_kernel, grid, dim3(256), 0, s, x, dtype, y, C, HW);
    return 0;
}
int op_nhwc_to_nchw(const half_t* x, void* y, int dtype, int N, int C, int HW, float scale, hipStream_t s, const int* img_map) {
    dim3 grid((HW + 63) / 64, (C + 31) / 32, N);
    PROF_WORK(0, (double)N * C * HW * (2.0 + (dtype == DT_F32 ? 4.0 : 2.0)));
    LAUNCH("nhwc_to_nchw", nhwc_to_nchw_kernel, grid, dim3(256), 0, s, x, y, dtype, C, HW, scale, img_map);
    return 0;
}
int op_avgpool_nchw(const void* x, void* y, int dtype, int NC, int Hin, int Win, int Hout, int Wout, hipStream_t s) {
    CTRL_CHECK(Hin % Hout == 0 && Win % Wout == 0, "avgpool: only integer pooling ratios are supported");
    const size_t total = (size_t)NC * Hout * Wout;
    PROF_WORK(0, (dtype == DT_F32 ? 4.0 : 2.0) * ((double)NC * Hin * Win + (double)total));
    LAUNCH("avgpool", avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, y, dtype, Hin, Win, Hout, Wout, total);
    return 0;
}
int op_timestep_sincos(const float* t, int t_count, float* out, int N, int dim, hipStream_t s) {
    CTRL_CHECK(dim % 2 == 0, "timestep embedding: dim must be even");
    CTRL_CHECK(t_count == 1 || t_count == N, "timestep embedding: need 1 or N timesteps");
    const int total = N * (dim / 2);
    PROF_WORK(0, 4.0 * N * dim);
    LAUNCH("sincos", sincos_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, t_count, 1, out, N, dim);
    return 0;
}
int op_frameidx_sincos(float* out, int N, int F, int dim, hipStream_t s) {
    const int total = N * (dim / 2);
    PROF_WORK(0, 4.0 * N * dim);
    LAUNCH("sincos", sincos_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const float*)nullptr, 0, F, out, N, dim);
    return 0;
}
int op_linear_small(const float* x, long ldx, const half_t* w, const float* b, float* out, long ldo,
                    int M, int N, int K, int in_silu, int out_silu, hipStream_t s) {
    CTRL_CHECK(K % 8 == 0 && ldx % 4 == 0, "linear_small: K must be a multiple of 8 and ldx of 4");
    CTRL_CHECK(M >= 1 && M <= 4096, "linear_small: M out of range");
    dim3 grid((N + 3) / 4, (M + 7) / 8);
    PROF_WORK(0, 2.0 * N * K + 4.0 * M * ((double)K + N));        // the weights once (fp16) + activations in / out (fp32): weight-streaming bound
    LAUNCH("linear_small", linear_small_kernel<8>, grid, dim3(256), 0, s, x, ldx, w, b, out, ldo, M, N, K, in_silu, out_silu);
    return 0;
}
int op_linear_small_group(const SmallLin* probs, int count, hipStream_t s) {
    for (int base = 0; base < count; base += kSmallGroup) {
        SmallLinGroup g;
        g.count = std::min(kSmallGroup, count - base);
        int blocks = 0;
        double bytes = 0;
        for (int i = 0; i < g.count; ++i) {
            const SmallLin& q = probs[base + i];
            bytes += 2.0 * q.N * q.K + 4.0 * q.M * ((double)q.K + q.N);
            CTRL_CHECK(q.K % 8 == 0 && q.ldx % 4 == 0 && q.M >= 1 && q.N >= 1, "linear_small_group: K must be a multiple of 8");
            g.p[i] = q;
            g.blk_begin[i] = blocks;
            blocks += ((q.N + 3) / 4) * ((q.M + 7) / 8);
        }
        PROF_WORK(0, bytes);
        prof_detail("%d problems", g.count);
        LAUNCH("linear_small", linear_small_group_kernel, dim3(blocks), dim3(256), 0, s, g);
    }
    return 0;
}

int op_blend(const void* xs, int xs_dt, const void* xt, int xt_dt, const float* mix, void* y, int y_dt, size_t n, hipStream_t s) {
    CTRL_CHECK(n % 8 == 0, "blend: element count must be a multiple of 8");
    CTRL_CHECK(xs_dt != DT_BF16 && xt_dt != DT_BF16 && y_dt != DT_BF16, "blend: fp16 / fp32 only");
    PROF_WORK(0, (double)n * ((xs_dt == DT_F32 ? 4.0 : 2.0) + (xt_dt == DT_F32 ? 4.0 : 2.0) + (y_dt == DT_F32 ? 4.0 : 2.0)));
    LAUNCH("blend", blend_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, xs, xs_dt, xt, xt_dt, mix, y, y_dt, n / 8);
    return 0;
}
int op_add_rowvec(const void* x, int x_dt, const float* v, long ldv, void* y, int y_dt, size_t M, int C, int rows_per_img, int vmod, hipStream_t s) {
    CTRL_CHECK(C % 8 == 0, "add_rowvec: C must be a multiple of 8");
    CTRL_CHECK(x_dt != DT_BF16 && y_dt != DT_BF16, "add_rowvec: fp16 / fp32 only");
    const size_t nch = M * (size_t)(C / 8);
    PROF_WORK(0, (double)M * C * ((x_dt == DT_F32 ? 4.0 : 2.0) + (y_dt == DT_F32 ? 4.0 : 2.0)));
    LAUNCH("add_rowvec", add_rowvec_kernel, dim3(grid_for(nch)), dim3(256), 0, s, x, x_dt, v, ldv, y, y_dt, nch, C, rows_per_img, vmod);
    return 0;
}
int op_add_rowvec_clip(const void* x, int x_dt, const float* v, long ldv, void* y, int y_dt, int B, int F, int L, int C, hipStream_t s) {
    CTRL_CHECK(C % 8 == 0, "add_rowvec_clip: C must be a multiple of 8");
    CTRL_CHECK(x_dt != DT_BF16 && y_dt != DT_BF16, "add_rowvec_clip: fp16 / fp32 only");
    const size_t nch = (size_t)B * F * L * (size_t)(C / 8);
    PROF_WORK(0, (double)B * F * L * C * ((x_dt == DT_F32 ? 4.0 : 2.0) + (y_dt == DT_F32 ? 4.0 : 2.0)));
    LAUNCH("add_rowvec_clip", add_rowvec_clip_kernel, dim3(grid_for(nch)), dim3(256), 0, s, x, x_dt, v, ldv, y, y_dt, nch, C, F, L, B);
    return 0;
}
int op_upsample2x_nhwc(const half_t* x, half_t* y, int N, int H, int W, int C, hipStream_t s) {
    CTRL_CHECK(C % 8 == 0, "upsample2x: C must be a multiple of 8");
    const size_t total = (size_t)N * 2 * H * 2 * W * (C / 8);
    PROF_WORK(0, 2.0 * N * H * W * C * 5);
    LAUNCH("upsample2x", upsample2x_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, y, H, W, C / 8, total);
    return 0;
}
// Zero fill as a KERNEL, not hipMemsetAsync (round 5): captured into a hipGraph on ONE stream (CTRL_ADAPTER_LANES=1), the forwards'
// memset nodes -- GroupNorm ticket / statistics pool, split-K ticket words, zero slots -- were not ordered against the kernels around
// them from the second replay on (every output of replay 1.. differed, the eager forward and the 4-lane graph did not: tools/diag/
// graph_1lane.py, ROCm 7.2); a kernel node always is.  16-byte stores, byte-wise head / tail.
namespace {
struct FillArgs { unsigned char* p[kMaxGroup]; size_t head[kMaxGroup], n16[kMaxGroup], tail[kMaxGroup]; };
__global__ __launch_bounds__(256) void fill_zero_kernel(FillArgs a) {
    const int z = blockIdx.y;
    unsigned char* p = z == 0 ? a.p[0] : (z == 1 ? a.p[1] : (z == 2 ? a.p[2] : a.p[3]));
    const size_t head = z == 0 ? a.head[0] : (z == 1 ? a.head[1] : (z == 2 ? a.head[2] : a.head[3]));
    const size_t n16 = z == 0 ? a.n16[0] : (z == 1 ? a.n16[1] : (z == 2 ? a.n16[2] : a.n16[3]));
    const size_t tail = z == 0 ? a.tail[0] : (z == 1 ? a.tail[1] : (z == 2 ? a.tail[2] : a.tail[3]));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4* body = (u4*)(p + head);
    for (size_t k = i; k < n16; k += stride) body[k] = u4{0u, 0u, 0u, 0u};
    if (i < head) p[i] = 0;
    if (i < tail) p[head + n16 * 16 + i] = 0;
}
}  // namespace
int op_fill_zero(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (t_collect) {                                 // lock-step replay of sibling blocks: deposited, launched by the collector's flush()
        int rc = 0;
        const int i = t_collect->slot(OpCollector::FILL, s, &rc);
        if (i < 0) return rc;
        t_collect->fz[i].p = p; t_collect->fz[i].bytes = bytes;
        return 0;
    }
    return op_fill_zero_group(&p, &bytes, 1, s);
}
int op_fill_zero_group(void* const* ps, const size_t* bytes, int n, hipStream_t s) {
    CTRL_CHECK(ps && bytes && n >= 1 && n <= kMaxGroup, "fill_zero_group: 1..4 fills");
    static_assert(kMaxGroup == 4, "fill_zero_kernel selects among four problems");
    FillArgs a = {};
    size_t most = 0;
    double total = 0;
    for (int i = 0; i < n; ++i) {
        a.p[i] = (unsigned char*)ps[i];
        size_t head = (16 - ((uintptr_t)ps[i] & 15)) & 15;
        if (head > bytes[i]) head = bytes[i];
        a.head[i] = head;
        a.n16[i] = (bytes[i] - head) / 16;
        a.tail[i] = bytes[i] - head - a.n16[i] * 16;
        if (a.n16[i] > most) most = a.n16[i];
        total += (double)bytes[i];
    }
    size_t blocks = (most + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    PROF_WORK(0, total);
    if (n > 1) prof_detail("x%d", n);
    LAUNCH("fill_zero", fill_zero_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, a);
    return 0;
}
int op_weighted_merge(const void* const* xs, const float* w, const int* widx, int K, void* out, int dtype, size_t n, hipStream_t s) {
    CTRL_CHECK(K >= 1 && K <= 8, "weighted_merge: 1..8 experts supported");
    MergeArgs m = {};
    for (int k = 0; k < K; ++k) { m.x[k] = xs[k]; m.widx[k] = widx[k]; }
    m.K = K;
    LAUNCH("merge", merge_kernel, dim3(grid_for(n)), dim3(256), 0, s, m, w, out, dtype, n);
    return 0;
}
int op_router_softmax(const float* wg, const int* mask, float* out, int R, int E, int equal_weights, hipStream_t s) {
    CTRL_CHECK(E >= 1 && E <= 16, "router: 1..16 experts supported");
    RouterMask rm;
    for (int e = 0; e < 16; ++e) rm.m[e] = (e < E && mask) ? mask[e] : 1;
    LAUNCH("router", router_softmax_kernel, dim3((R + 63) / 64), dim3(64), 0, s, wg, rm, out, R, E, equal_weights);
    return 0;
}
int op_pack_conv_w(const void* w, int dtype, half_t* out, int Cout, int Cin, int taps, hipStream_t s, int* ovf) {
    LAUNCH("pack", pack_conv_w_kernel, dim3(grid_for((size_t)Cout * Cin * taps)), dim3(256), 0, s, w, dtype, out, Cout, Cin, taps, ovf);
    return 0;
}
int op_pack_conv_w_dup(const void* w, int dtype, half_t* out, int Cout, int Cin, int taps, hipStream_t s, int* ovf) {
    LAUNCH("pack", pack_conv_w_dup_kernel, dim3(grid_for((size_t)Cout * Cin * taps * 2)), dim3(256), 0, s, w, dtype, out, Cout, Cin, taps, ovf);
    return 0;
}
int op_pack_conv_w_direct(const void* w, int dtype, float* out, int Cout, int Cin, hipStream_t s) {
    LAUNCH("pack", pack_conv_w_direct_kernel, dim3(grid_for((size_t)Cout * Cin * 9)), dim3(256), 0, s, w, dtype, out, Cout, Cin);
    return 0;
}
int op_pack_linear_w(const void* w, int dtype, half_t* out, int N, int K, int geglu, hipStream_t s, int* ovf) {
    CTRL_CHECK(!geglu || N % 32 == 0, "pack_linear_w: GEGLU needs N % 32 == 0");
    LAUNCH("pack", pack_linear_w_kernel, dim3(grid_for((size_t)N * K)), dim3(256), 0, s, w, dtype, out, N, K, geglu, ovf);
    return 0;
}
int op_pack_vec(const void* v, int dtype, float* out, int N, int geglu, hipStream_t s) {
    LAUNCH("pack", pack_vec_kernel, dim3((N + 255) / 256), dim3(256), 0, s, v, dtype, out, N, geglu);
    return 0;
}
