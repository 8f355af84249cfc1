// Direct 3x3 convolution (pad 1, stride 1|2) for the tiny-channel head of the hot path (gfx950):
//   ControlNetModel.conv_in            4 -> 320 @64x64        (controlnet/controlnet.py:307-309, :802)
//   ControlNetConditioningEmbedding    3->16, 16->16 @512^2, 16->32 s2, 32->32 @256^2
//                                      (controlnet/controlnet.py:80-104)
// K = 9*Cin is 27..288 here, far below an MFMA tile, and the 512x512 maps make these layers
// HBM/VALU-bound: one thread per output pixel, COT output channels in registers, weights fetched
// through the scalar cache (wave-uniform addresses -> s_load), fp32 math, fused bias (+SiLU), output
// channels-last fp16 written with 16-byte stores.  The wider embedder layers (Cin >= 32 with Cout >= 96)
// go through the MFMA implicit GEMM instead.
#include "ops.h"

namespace {

template <int CIN, int COT, bool NCHW_IN>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const void* __restrict__ in, int in_dt,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             half_t* __restrict__ out, int N, int Cout, int Hin, int Win,
                                                             int Hout, int Wout, int stride, int silu) {
    const size_t npix = (size_t)N * Hout * Wout;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const int co0 = blockIdx.y * COT;
    const int ox = (int)(pix % Wout);
    const size_t r = pix / Wout;
    const int oy = (int)(r % Hout);
    const int n = (int)(r / Hout);

    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = bias ? bias[co0 + c] : 0.f;

    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
        const bool ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
        float xin[CIN];
        if (NCHW_IN) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
                xin[ci] = ok ? load_as_f32(in, (((size_t)n * CIN + ci) * Hin + iy) * Win + ix, in_dt) : 0.f;
        } else {
            const half_t* ip = (const half_t*)in + (((size_t)n * Hin + (ok ? iy : 0)) * Win + (ok ? ix : 0)) * CIN;
#pragma unroll
            for (int c8 = 0; c8 < CIN / 8; ++c8) {
                const h8 v = *(const h8*)(ip + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[c8 * 8 + j] = ok ? (float)v[j] : 0.f;
            }
        }
        const float* wt = w + (size_t)tap * CIN * Cout + co0;   // wave-uniform
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[c] += xin[ci] * wt[(size_t)ci * Cout + c];
    }
    half_t* op = out + pix * Cout + co0;
#pragma unroll
    for (int c8 = 0; c8 < COT / 8; ++c8) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[c8 * 8 + j];
            if (silu) v = silu_f(v);
            o[j] = (half_t)v;
        }
        *(h8*)(op + c8 * 8) = o;
    }
}

// ---- the same convolution on the matrix cores for the channels-last layers (Cin = 16 | 32, Cout = 16 | 32; round 5) ----
// The direct kernel above runs these at ~70 TFLOP/s of fp32 VALU work (16 -> 16 on a 512^2 map: 9.7 GFLOP, 138 us at b = 8, against
// 134 MB of HBM traffic = 27 us): with N = 32 frames they are 2 ms of every video step.  Here K = (tap, cin) is walked in 32-deep
// steps of v_mfma_f32_16x16x32_f16 (Cin = 16: two taps per step, the tenth half-step is zero; Cin = 32: one tap per step); an M
// fragment is 16 consecutive output pixels of a row, and a lane fetches ITS 16-byte A chunk -- 8 channels of one shifted input
// pixel -- straight from global memory (the nine taps of neighbouring pixels overlap: the reuse sits in L1 / L2; no LDS).  The
// weights ([Cout][tap][Cin] fp16, the implicit GEMM's pack) stay in registers for the whole strip a wave walks down the image.
// Operands swapped (D = W.A^T) so a lane ends up with 4 consecutive output channels of one pixel: bias, SiLU, one 8-byte store.
typedef const h8* h8p;
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_mfma_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                 const float* __restrict__ bias, half_t* __restrict__ out,
                                                                 int N, int Hin, int Win, int Hout, int Wout, int stride, int silu, int rows_per_unit) {
    constexpr int KS = (9 * CIN + 31) / 32;          // k-steps: 5 (Cin 16) | 9 (Cin 32)
    constexpr int NF = COUT / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 15, q = lane >> 4;
    // work unit = (image, 16-pixel column strip, block of rows_per_unit rows); one per wave
    const int fxn = (Wout + 15) >> 4, ryn = (Hout + rows_per_unit - 1) / rows_per_unit;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= (long)N * fxn * ryn) return;
    const int n = (int)(unit / ((long)fxn * ryn));
    const int rem = (int)(unit - (long)n * fxn * ryn);
    const int ry = rem / fxn, fx = rem - ry * fxn;      // strips of one row block are neighbours: they share their halo columns in L1 / L2
    const int ox = fx * 16 + p;
    // per k-step: tap and channel offset of this lane's 8-value chunk
    int dky[KS], dkx[KS], cio[KS];
    bool kok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kc = s * 32 + q * 8;
        const int tap = kc / CIN;
        kok[s] = tap < 9;
        dky[s] = tap / 3 - 1;
        dkx[s] = tap - 3 * (tap / 3) - 1;
        cio[s] = kc - tap * CIN;
    }
    // weights: B fragment f of step s = W[f*16 + p][s*32 + q*8 .. +8]
    h8 wf[KS][NF];
    const h8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int f = 0; f < NF; ++f) wf[s][f] = kok[s] ? *(h8p)(w + (size_t)(f * 16 + p) * (9 * CIN) + s * 32 + q * 8) : hz;
    f4 bv[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bv[f] = bias ? *(const f4*)(bias + f * 16 + q * 4) : f4{0.f, 0.f, 0.f, 0.f};
    const half_t* xin = x + (size_t)n * Hin * Win * CIN;
    const int oy0 = ry * rows_per_unit, oy1 = min(Hout, oy0 + rows_per_unit);
    for (int oy = oy0; oy < oy1; ++oy) {
        h8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int iy = oy * stride + dky[s], ix = ox * stride + dkx[s];
            const bool ok = kok[s] && ox < Wout && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            af[s] = ok ? *(h8p)(xin + ((size_t)iy * Win + ix) * CIN + cio[s]) : hz;
        }
        f4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = bv[f];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][f], af[s], acc[f], 0, 0, 0);
        if (ox < Wout) {
            half_t* op = out + (((size_t)n * Hout + oy) * Wout + ox) * COUT + q * 4;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                f4 v = acc[f];
                if (silu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
                }
                const h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *(h4*)(op + f * 16) = o;
            }
        }
    }
}

template <int CIN, int COUT>
int launch_small_mfma(const half_t* x, const half_t* w, const float* bias, half_t* out, int N, int Hin, int Win, int stride, int silu, hipStream_t s) {
    const int Hout = (Hin + 2 - 3) / stride + 1, Wout = (Win + 2 - 3) / stride + 1;
    const int rpu = 8;
    const long units = (long)N * ((Wout + 15) / 16) * ((Hout + rpu - 1) / rpu);
    PROF_WORK(2.0 * N * Hout * Wout * COUT * 9.0 * CIN, 2.0 * ((double)N * Hin * Win * CIN + (double)N * Hout * Wout * COUT));
    prof_detail("N%d %d->%d %dx%d s%d", N, CIN, COUT, Hin, Win, stride);
    LAUNCH("conv3x3_small_mfma", (conv3x3_small_mfma_kernel<CIN, COUT>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s,
           x, w, bias, out, N, Hin, Win, Hout, Wout, stride, silu, rpu);
    return 0;
}

template <int CIN, int COT, bool NCHW_IN>
int launch_direct(const void* in, int in_dt, const float* w, const float* bias, half_t* out, int N, int Cout,
                  int Hin, int Win, int stride, int silu, hipStream_t s) {
    const int Hout = (Hin + 2 - 3) / stride + 1, Wout = (Win + 2 - 3) / stride + 1;
    const size_t npix = (size_t)N * Hout * Wout;
    dim3 grid((unsigned)((npix + 255) / 256), Cout / COT);
    // roofline = HBM, its maps read / written once (the weights are a few KB; 2 B per input value: the boundary tensors of the
    // benchmark are fp16).  (Its fp32 VALU work -- 2 * 9 * Cin * Cout flops per pixel -- is what actually bounds the 16 / 32-channel
    // layers, which is why they moved to the matrix cores: conv3x3_small_mfma_kernel.)
    PROF_WORK(0, 2.0 * ((double)N * Hin * Win * CIN + (double)npix * Cout));
    prof_detail("N%d %d->%d %dx%d s%d%s", N, CIN, Cout, Hin, Win, stride, NCHW_IN ? " nchw" : "");
    LAUNCH("conv3x3_direct", (conv3x3_direct_kernel<CIN, COT, NCHW_IN>), grid, dim3(256), 0, s,
           in, in_dt, w, bias, out, N, Cout, Hin, Win, Hout, Wout, stride, silu);
    return 0;
}

}  // namespace

int op_conv3x3_direct(const void* in, int in_dtype, int in_nchw, const float* w, const float* bias, half_t* out,
                      int N, int Cin, int Cout, int Hin, int Win, int stride, int silu, hipStream_t s) {
    CTRL_CHECK(stride == 1 || stride == 2, "conv3x3_direct: stride must be 1 or 2");
    CTRL_CHECK(Cout % 16 == 0, "conv3x3_direct: Cout must be a multiple of 16");
    const bool c32 = (Cout % 32) == 0;
#define DIRECT_CASE(CI, NCHW)                                                                                   \
    if (Cin == CI && (in_nchw != 0) == NCHW) {                                                                  \
        return c32 ? launch_direct<CI, 32, NCHW>(in, in_dtype, w, bias, out, N, Cout, Hin, Win, stride, silu, s) \
                   : launch_direct<CI, 16, NCHW>(in, in_dtype, w, bias, out, N, Cout, Hin, Win, stride, silu, s); \
    }
    DIRECT_CASE(3, true)
    DIRECT_CASE(4, true)
    DIRECT_CASE(16, false)
    DIRECT_CASE(32, false)
#undef DIRECT_CASE
    CTRL_FAIL("conv3x3_direct: unsupported (Cin=" + std::to_string(Cin) + ", nchw=" + std::to_string(in_nchw) +
              "); supported: NCHW Cin 3|4, NHWC Cin 16|32");
}


bool op_conv3x3_small_mfma_applies(int Cin, int Cout) { return (Cin == 16 || Cin == 32) && (Cout == 16 || Cout == 32); }

// x NHWC fp16 [N][Hin][Win][Cin]; w fp16 [Cout][9][Cin] (op_pack_conv_w); out NHWC fp16 (+ bias, optional SiLU)
int op_conv3x3_small_mfma(const half_t* x, const half_t* w, const float* bias, half_t* out, int N, int Cin, int Cout, int Hin, int Win,
                          int stride, int silu, hipStream_t s) {
    CTRL_CHECK(stride == 1 || stride == 2, "conv3x3_small_mfma: stride must be 1 or 2");
    CTRL_CHECK(op_conv3x3_small_mfma_applies(Cin, Cout), "conv3x3_small_mfma: Cin and Cout must be 16 or 32");
    CTRL_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "conv3x3_small_mfma: pointers must be 16-byte aligned");
    if (Cin == 16 && Cout == 16) return launch_small_mfma<16, 16>(x, w, bias, out, N, Hin, Win, stride, silu, s);
    if (Cin == 16 && Cout == 32) return launch_small_mfma<16, 32>(x, w, bias, out, N, Hin, Win, stride, silu, s);
    if (Cin == 32 && Cout == 16) return launch_small_mfma<32, 16>(x, w, bias, out, N, Hin, Win, stride, silu, s);
    return launch_small_mfma<32, 32>(x, w, bias, out, N, Hin, Win, stride, silu, s);
}
