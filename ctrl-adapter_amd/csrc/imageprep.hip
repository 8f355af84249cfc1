// Conditioning-image preparation on the GPU: model/ctrl_helper.py:268-296 `prepare_images` =
// per frame PIL convert("RGB") -> PIL resize(LANCZOS) -> uint8 / 255 -> NCHW, batch repeat, CFG duplication.
// The arithmetic is Pillow's 8-bit separable resampling (Resample.c): 22-bit fixed-point weights, a horizontal pass and a
// vertical pass with rounding + clamping to uint8 after EACH pass.  Integer work -> bit-exact with the reference; the
// weight tables come from the host (ctrl-adapter_amd/image_prep.py restates precompute_coeffs in double precision).
// HBM-bound byte work: one thread per output pixel, the three channels of a pixel together; the second pass converts and
// writes every replica (batch repeat x CFG) of the pixel so the result is produced in one sweep, in the caller's dtype.
#include "ops.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;                                   // arithmetic shift (Resample.c clip8 lookup index)
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// src [F][H][Win][3] -> dst [F][H][Wout][3]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                         long rows, int Win, int Wout) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Wout) return;
    const long r = i / Wout;
    const int xo = (int)(i - r * Wout);
    const int xmin = bounds[2 * xo], xmax = bounds[2 * xo + 1];
    const int* k = kk + (size_t)xo * ksize;
    const unsigned char* p = src + ((size_t)r * Win + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int w = k[x];
        s0 += (int)p[3 * x] * w; s1 += (int)p[3 * x + 1] * w; s2 += (int)p[3 * x + 2] * w;
    }
    unsigned char* q = dst + (size_t)i * 3;
    q[0] = (unsigned char)clip8(s0); q[1] = (unsigned char)clip8(s1); q[2] = (unsigned char)clip8(s2);
}

// src [F][Hin][W][3] uint8 -> out [cfg][rep*F][3][Hout][W] (value / 255 in fp32, then the output dtype); bounds == null:
// no vertical resampling (Hin == Hout)
__global__ __launch_bounds__(256) void resample_v_store_kernel(const unsigned char* __restrict__ src, void* __restrict__ out, int dt,
                                                               const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                               int F, int Hin, int Hout, int W, int rep, int cfg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)Hout * W;
    if (i >= (long)F * per) return;
    const int f = (int)(i / per);
    const long rem = i - (long)f * per;
    const int yo = (int)(rem / W), x = (int)(rem - (long)yo * W);
    int v[3];
    if (bounds) {
        const int ymin = bounds[2 * yo], ymax = bounds[2 * yo + 1];
        const int* k = kk + (size_t)yo * ksize;
        const unsigned char* p = src + (((size_t)f * Hin + ymin) * W + x) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < ymax; ++y) {
            const int w = k[y];
            const unsigned char* q = p + (size_t)y * W * 3;
            s0 += (int)q[0] * w; s1 += (int)q[1] * w; s2 += (int)q[2] * w;
        }
        v[0] = clip8(s0); v[1] = clip8(s1); v[2] = clip8(s2);
    } else {
        const unsigned char* q = src + (((size_t)f * Hin + yo) * W + x) * 3;
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
    }
    const long frames = (long)rep * F;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float val = (float)v[c] / 255.0f;              // np.float32(u8) / 255.0: one correctly rounded fp32 division
        for (int g = 0; g < cfg; ++g)
            for (int r = 0; r < rep; ++r)                    // torch .repeat(rep, 1, 1, 1): frame index r*F + f
                store_from_f32(out, (((size_t)g * frames + (size_t)r * F + f) * 3 + c) * per + rem, dt, val);
    }
}

}  // namespace

int op_prepare_images(const unsigned char* src, int F, int Hin, int Win, const int* hbounds, const int* hk, int hks,
                      const int* vbounds, const int* vk, int vks, unsigned char* tmp, void* out, int out_dtype,
                      int W, int H, int rep, int cfg, hipStream_t s) {
    CTRL_CHECK(src && out && F >= 1 && Hin >= 1 && Win >= 1 && W >= 1 && H >= 1 && rep >= 1 && (cfg == 1 || cfg == 2), "prepare_images: bad argument");
    CTRL_CHECK((hbounds != nullptr) == (Win != W) && (vbounds != nullptr) == (Hin != H), "prepare_images: weight tables must be given exactly for the resampled axes");
    CTRL_CHECK(!hbounds || (tmp && hk && hks > 0), "prepare_images: horizontal pass needs a scratch image and weights");
    CTRL_CHECK(!vbounds || (vk && vks > 0), "prepare_images: vertical pass needs weights");
    const unsigned char* mid = src;
    if (hbounds) {
        const long n = (long)F * Hin * W;
        PROF_WORK(0, 3.0 * F * Hin * ((double)Win + W));
        LAUNCH("prepare_images", resample_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, tmp, hbounds, hk, hks, (long)F * Hin, Win, W);
        mid = tmp;
    }
    const long n = (long)F * H * W;
    PROF_WORK(0, 3.0 * F * (double)Hin * W + 3.0 * rep * cfg * F * (double)H * W * (out_dtype == DT_F32 ? 4 : 2));
    LAUNCH("prepare_images", resample_v_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mid, out, out_dtype, vbounds, vk, vks,
           F, Hin, H, W, rep, cfg);
    return 0;
}
