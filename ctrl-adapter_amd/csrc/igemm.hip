// Implicit-GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x32_f16), gfx950 only.
//
// One kernel family covers every dense contraction of the hot path:
//   * Linear / 1x1 conv            (IG_ROWS)      -- reference: every nn.Linear / Conv2d(k=1) under
//                                                     controlnet/controlnet.py:366-424 and
//                                                     model/adapter_spatial_temporal.py:56-69
//   * 3x3 conv, stride 1|2, pad 1, optional nearest x2 up-sampling folded into the gather
//                                   (IG_CONV2D)    -- model/resnet_block_2d.py:164-221 (conv1/conv2,
//                                                     Upsample2D), diffusers Downsample2D
//   * Conv3d (3,1,1) over frames    (IG_TEMPORAL)  -- TemporalResnetBlock, adapter_spatial_temporal.py:96-104
//
// Layout: activations are channels-last fp16 ([pixels][C]); the K axis of the GEMM is (tap, cin) with
// cin contiguous, so every A-tile row is one 16-byte-vectorisable run of channels of one (shifted)
// input pixel; zero padding is a predicated load.  Weights are pre-packed [Cout][tap][Cin].
//
// Tiling: BM x BN x BK per 256-thread workgroup (4 wavefronts in a 2x2 grid), A/B tiles staged
// global -> VGPR -> LDS (double-buffered, next tile's global loads in flight during the MFMAs),
// LDS rows padded to an odd number of 16-byte slots so the ds_read_b128 fragment reads of 16 distinct
// rows are bank-conflict free (MI355X_MICROARCH.md, LDS section).  fp32 accumulation; the epilogue
// fuses bias, per-image channel vector (time embedding), GEGLU, residual add, scale and the output
// layout (row-major slice, or transposed [img][C][tokens] for NCHW results / the V^T attention operand).
#include "ops.h"

namespace {

template <int BM, int BN, int BK, int MODE>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmArgs a, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LDS_LD = BK + 8;           // halfs; (BK*2+16) bytes = odd number of 16-B slots
    constexpr int WM = BM / 2, WN = BN / 2;  // per-wave tile
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int CPR = BK / 8;              // 16-B chunks per tile row
    constexpr int RPP = 256 / CPR;           // rows staged per pass
    constexpr int APASS = BM / RPP, BPASS = BN / RPP;
    static_assert(APASS >= 1 && BPASS >= 1, "tile too small for 256 threads");

    half_t* As = (half_t*)smem_raw;                  // [2][BM][LDS_LD]
    half_t* Bs = As + 2 * BM * LDS_LD;               // [2][BN][LDS_LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: each XCD (blockIdx % 8) walks a contiguous range of tiles so the
    // A rows / weight panels it re-reads stay in that XCD's private L2.
    const int nblk = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging coordinates ----
    const int srow = tid / CPR;          // row within a pass
    const int sc8 = (tid % CPR) * 8;     // channel offset of this thread's 16-B chunk
    size_t a_off[APASS];                 // MODE ROWS/TEMPORAL: element offset of row start; CONV2D: image base pixel
    int a_y[APASS], a_x[APASS];          // CONV2D: oy*stride-pad, ox*stride-pad; TEMPORAL: a_y = frame index
    bool a_ok[APASS];
    const int pad = (a.taps == 9) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + srow + i * RPP;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        if (MODE == IG_ROWS) {
            a_off[i] = (size_t)mm * a.lda;
            a_y[i] = a_x[i] = 0;
        } else if (MODE == IG_CONV2D) {
            const int hw = a.Hout * a.Wout;
            const int n = mm / hw, rem = mm - n * hw;
            const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
            a_off[i] = (size_t)n * a.Hin * a.Win;
            a_y[i] = oy * a.stride - pad;
            a_x[i] = ox * a.stride - pad;
        } else {
            a_off[i] = (size_t)mm * a.lda;
            a_y[i] = (mm / a.HW) % a.F;
            a_x[i] = 0;
        }
    }
    const int VH = a.Hin * a.up, VW = a.Win * a.up;   // virtual (up-sampled) input grid
    const int ushift = (a.up == 2) ? 1 : 0;

    size_t b_off[BPASS];
    bool b_ok[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + srow + i * RPP;
        b_ok[i] = n < a.Nout;
        b_off[i] = (size_t)(b_ok[i] ? n : 0) * a.Ktot;
    }

    const half_t* Aptr = (const half_t*)a.A;
    const half_t* Wptr = (const half_t*)a.W;
    const half_t* Rptr = (const half_t*)a.res;
    h8 areg[APASS], breg[BPASS];
    const h8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        int tap = 0, c0 = k0;
        if (MODE != IG_ROWS) { tap = k0 / a.Cin; c0 = k0 - tap * a.Cin; }
        int ky = 0, kx = 0;
        if (MODE == IG_CONV2D && a.taps == 9) { ky = tap / 3; kx = tap - 3 * ky; }
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            bool ok = a_ok[i];
            size_t off;
            if (MODE == IG_ROWS) {
                off = a_off[i] + k0 + sc8;
            } else if (MODE == IG_CONV2D) {
                const int vy = a_y[i] + ky, vx = a_x[i] + kx;
                ok = ok && vy >= 0 && vy < VH && vx >= 0 && vx < VW;
                const int sy = vy >> ushift, sx = vx >> ushift;
                off = (a_off[i] + (size_t)(ok ? sy : 0) * a.Win + (ok ? sx : 0)) * a.lda + c0 + sc8;
            } else {
                const int f = a_y[i] + tap - 1;
                ok = ok && f >= 0 && f < a.F;
                off = a_off[i] + (ok ? (long)(tap - 1) * a.HW * a.lda : 0) + c0 + sc8;
            }
            areg[i] = ok ? *(const h8*)(Aptr + off) : hzero;
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            breg[i] = b_ok[i] ? *(const h8*)(Wptr + b_off[i] + k0 + sc8) : hzero;
    };
    auto lds_store = [&](int buf) {
        half_t* Ab = As + buf * BM * LDS_LD;
        half_t* Bb = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < APASS; ++i) *(h8*)(Ab + (srow + i * RPP) * LDS_LD + sc8) = areg[i];
#pragma unroll
        for (int i = 0; i < BPASS; ++i) *(h8*)(Bb + (srow + i * RPP) * LDS_LD + sc8) = breg[i];
    };

    f4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.Ktot / BK;
    gload(0);
    lds_store(0);
    __syncthreads();

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);            // global loads stay in flight under the MFMAs
        const half_t* Ab = As + cur * BM * LDS_LD + (wm * WM + frow) * LDS_LD + fk;
        const half_t* Bb = Bs + cur * BN * LDS_LD + (wn * WN + frow) * LDS_LD + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            h8 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *(const h8*)(Ab + mi * 16 * LDS_LD + kk * 32);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const h8*)(Bb + ni * 16 * LDS_LD + kk * 32);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) lds_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---------------- epilogue ----------------
    // C/D fragment map of v_mfma_f32_16x16x32: row = (lane>>4)*4 + i, col = lane&15
    const int erow = (lane >> 4) * 4, ecol = lane & 15;
    constexpr int NSTEP = 1;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        if (a.geglu && (ni & 1)) continue;           // odd fragments are the gates of the even ones
        const int pcb = n0 + wn * WN + ni * 16;      // packed column block start
        if (pcb >= a.Nout) continue;
        const int pcol = pcb + ecol;                 // packed column (bias index)
        const int ocol = a.geglu ? ((pcb >> 5) << 4) + ecol : pcol;   // output column
        const int ocb = ocol - ecol;
        // segment lookup (segment boundaries are multiples of 16 -> uniform per fragment)
        int si = 0;
#pragma unroll
        for (int k = 1; k < 3; ++k)
            if (k < a.nseg && ocb >= a.seg[k].col_begin) si = k;
        const IGemmSeg sg = a.seg[si];
        const int scol = ocol - sg.col_begin;
        const float bh = a.bias ? a.bias[pcol] : 0.f;
        const float bg = (a.geglu && a.bias) ? a.bias[pcol + 16] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rbase = m0 + wm * WM + mi * 16 + erow;
            if (rbase >= a.M) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rbase + i;
                float x = acc[mi][ni][i] + bh;
                if (a.rowvec && row < a.M) x += a.rowvec[(size_t)(row / a.rows_per_img) * a.rowvec_ld + pcol];
                if (a.geglu) {
                    const float g = acc[mi][ni + (NI > 1 ? 1 : 0)][i] + bg;
                    x = x * gelu_erf_f(g);
                }
                if (a.act == 1) x = silu_f(x);
                if (Rptr && row < a.M) x += (float)Rptr[(size_t)row * a.ldres + ocol];
                v[i] = x * a.scale;
            }
            if (sg.fmt == SEG_ROW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = rbase + i;
                    if (row < a.M) store_from_f32(sg.out, (size_t)row * sg.ld + scol, sg.dtype, v[i]);
                }
            } else {
                const int img = rbase / sg.L, tok = rbase - img * sg.L;
                const size_t base = ((size_t)img * sg.ncols + scol) * sg.ld + tok;
                const bool vec = ((sg.L & 3) == 0) && ((sg.ld & 3) == 0) && (rbase + 3 < a.M);
                if (vec) {
                    if (sg.dtype == DT_F16) {
                        h4 p = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *(h4*)((half_t*)sg.out + base) = p;
                    } else if (sg.dtype == DT_F32) {
                        *(f4*)((float*)sg.out + base) = f4{v[0], v[1], v[2], v[3]};
                    } else {
                        typedef u16 us4 __attribute__((ext_vector_type(4)));
                        us4 p = {f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
                        *(us4*)((u16*)sg.out + base) = p;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = rbase + i;
                        if (row >= a.M) break;
                        const int im = row / sg.L, tk = row - im * sg.L;
                        store_from_f32(sg.out, ((size_t)im * sg.ncols + scol) * sg.ld + tk, sg.dtype, v[i]);
                    }
                }
            }
        }
    }
    (void)NSTEP;
}

template <int BM, int BN, int BK, int MODE>
int launch_cfg(const IGemmArgs& a, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * (BM + BN) * (BK + 8) * sizeof(half_t);
    static bool attr_done = false;
    if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, BK, MODE>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Nout + BN - 1) / BN;
    PROF_WORK(2.0 * a.M * a.Nout * a.Ktot, 2.0 * ((double)a.M * a.Cin + (double)a.Nout * a.Ktot + (double)a.M * a.Nout));
    const char* tag = MODE == IG_ROWS ? "igemm_rows" : (MODE == IG_CONV2D ? "igemm_conv" : "igemm_temporal");
    LAUNCH(tag, (igemm_kernel<BM, BN, BK, MODE>), dim3(ntm * ntn), dim3(256), smem, s, a, ntm, ntn);
    return 0;
}

template <int MODE>
int dispatch(const IGemmArgs& a, hipStream_t s) {
    const bool bk64 = (a.Cin % 64) == 0;
    // Prefer 128x128 tiles; fall back to 64x64 when the big tiling would leave most of the 256 CUs idle
    // or the N extent is not a multiple of 128 but is of 64 (e.g. 320, 960).
    const long big = (long)((a.M + 127) / 128) * ((a.Nout + 127) / 128);
    const bool n_waste = (a.Nout % 128) != 0 && (a.Nout % 128) <= 64 && a.Nout < 1024;
    const bool small = big < 192 || n_waste;
    if (bk64) {
        if (!small) return launch_cfg<128, 128, 64, MODE>(a, s);
        return launch_cfg<64, 64, 64, MODE>(a, s);
    } else {
        if (!small) return launch_cfg<128, 128, 32, MODE>(a, s);
        return launch_cfg<64, 64, 32, MODE>(a, s);
    }
}

}  // namespace

int op_igemm(const IGemmArgs& a, hipStream_t s) {
    CTRL_CHECK(a.M > 0 && a.Nout > 0 && a.Ktot > 0, "igemm: empty problem");
    CTRL_CHECK(a.Cin % 32 == 0, "igemm: Cin must be a multiple of 32 (got " + std::to_string(a.Cin) + ")");
    CTRL_CHECK(a.Ktot == a.taps * a.Cin, "igemm: Ktot != taps*Cin");
    CTRL_CHECK(a.lda % 8 == 0, "igemm: lda must be a multiple of 8 (16-byte vector loads)");
    CTRL_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "igemm: A/W must be 16-byte aligned");
    CTRL_CHECK(a.nseg >= 1 && a.nseg <= 3, "igemm: nseg must be 1..3");
    CTRL_CHECK(!a.geglu || (a.Nout % 32) == 0, "igemm: GEGLU needs Nout % 32 == 0");
    for (int i = 0; i < a.nseg; ++i) {
        CTRL_CHECK(a.seg[i].col_begin % 16 == 0, "igemm: segment boundary must be a multiple of 16");
        CTRL_CHECK(a.seg[i].out != nullptr, "igemm: null segment output");
    }
    if (a.mode == IG_CONV2D) {
        CTRL_CHECK(a.taps == 9 || a.taps == 1, "igemm conv2d: taps must be 1 or 9");
        CTRL_CHECK((a.up == 1 || a.up == 2) && (a.stride == 1 || a.stride == 2), "igemm conv2d: up/stride must be 1|2");
        return dispatch<IG_CONV2D>(a, s);
    } else if (a.mode == IG_TEMPORAL) {
        CTRL_CHECK(a.taps == 3 && a.F > 0 && a.HW > 0, "igemm temporal: taps must be 3");
        return dispatch<IG_TEMPORAL>(a, s);
    }
    CTRL_CHECK(a.taps == 1, "igemm rows: taps must be 1");
    return dispatch<IG_ROWS>(a, s);
}

int op_linear(const half_t* A, long lda, const half_t* W, const float* bias, half_t* out, long ldo,
              int M, int N, int K, const half_t* res, long ldres, hipStream_t s) {
    IGemmArgs g = {};
    g.A = A; g.lda = lda; g.mode = IG_ROWS; g.Cin = K; g.taps = 1;
    g.W = W; g.M = M; g.Nout = N; g.Ktot = K;
    g.bias = bias; g.res = res; g.ldres = ldres; g.scale = 1.f;
    g.nseg = 1;
    g.seg[0] = IGemmSeg{out, ldo, 0, N, SEG_ROW, DT_F16, 1, 0};
    return op_igemm(g, s);
}
