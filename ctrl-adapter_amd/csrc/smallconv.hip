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

template <int CIN, int COT, bool NCHW_IN>
int launch_direct(const void* in, int in_dt, const float* w, const float* bias, half_t* out, int N, int Cout,
                  int Hin, int Win, int stride, int silu, hipStream_t s) {
    const int Hout = (Hin + 2 - 3) / stride + 1, Wout = (Win + 2 - 3) / stride + 1;
    const size_t npix = (size_t)N * Hout * Wout;
    dim3 grid((unsigned)((npix + 255) / 256), Cout / COT);
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
