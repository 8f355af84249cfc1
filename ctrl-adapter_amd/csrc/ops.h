// Internal op layer of libctrlhip: every function enqueues hand-written gfx950 kernels on `stream`
// and returns 0 / non-zero (message via ctrl_last_error()).  Activations are fp16, channels-last
// ("NHWC": a feature map [n][y][x][c] is a row-major token matrix [n*H*W][C]); accumulation,
// normalisation statistics, softmax and all epilogue math are fp32.
#pragma once
#include "common.h"
#include "policy.h"
#include "../../include/ctrl_hip.h"

// ------------------------------------------------------------------------------------------
// Implicit GEMM (MFMA f16 -> f32):  out[m][n] = epilogue( sum_k Agather[m][k] * W[n][k] )
// ------------------------------------------------------------------------------------------
enum { IG_ROWS = 0, IG_CONV2D = 1, IG_TEMPORAL = 2 };
enum { SEG_ROW = 0, SEG_TRANSPOSED = 1 };

typedef ctrl_igemm_seg IGemmSeg;     // see include/ctrl_hip.h for field docs
typedef ctrl_igemm_desc IGemmArgs;
int op_igemm(const IGemmArgs& a, hipStream_t s);
// K-split factor op_igemm would use given scratch (1 = no split); scratch needed = factor * M * Nout * sizeof(float)
int igemm_splitk_factor(const IGemmArgs& a);
// scratch bytes for that factor (>= factor * M * Nout * sizeof(float): the in-launch reduction keeps tile-shaped slabs)
size_t igemm_splitk_ws_bytes(const IGemmArgs& a, int sk);
// ticket words (ctrl_igemm_desc::splitk_tickets) the in-launch reduction of `sk` splits uses; 0 = that form does not apply
size_t igemm_splitk_ticket_words(const IGemmArgs& a, int sk);
// tile walk order of the implicit GEMM (tile_order.h): "legacy" | "auto" | "m,G" | "n,G"; 0 = accepted
int igemm_set_order(const char* spec);
// which problems take the 8-phase wide-tile kernel (igemm8_kernel): 0 none, 1 where the grid fills the chip (default), 2 every
// eligible problem (the parity tests run their small shapes through it this way), -1 back to CTRL_IGEMM8 / the default; 0 = accepted
int igemm_set_wide(int mode);
void igemm_tile_of(int bid, int ntm, int ntn, int mode, int group, int* tile_m, int* tile_n);
// Grouped launch (round 5): n <= kMaxIGemmGroup problems of the same shape and epilogue form (sibling adapter blocks of one
// pyramid level) in ONE launch -- every problem computed exactly as it would be alone; problems that do not agree are launched one
// by one.  See igemm.hip: IGemmGroup.
constexpr int kMaxIGemmGroup = 4;
constexpr int kMaxGroup = kMaxIGemmGroup;
int op_igemm_group(const IGemmArgs* a, int n, hipStream_t s);
// Fused GEGLU feed-forward, dim 512 / hidden 2048 (ffn.hip; include/ctrl_hip.h: ctrl_ffn_desc): one launch instead of the GEGLU GEMM +
// the K = 2048 GEMM; op_ffn_fused deposits into an installed collector like op_igemm does
typedef ctrl_ffn_desc FfnArgs;
bool op_ffn_fused_shape_ok(int dim, int inner);
int op_ffn_fused(const FfnArgs& a, hipStream_t s);
int op_ffn_fused_group(const FfnArgs* a, int n, hipStream_t s);
int op_ffn_pack_w2(const half_t* w2, half_t* out, int N, int K, hipStream_t s);
// convenience: plain linear out[M][N] (fp16 row-major) = A[M][K] * W[N][K]^T + bias
int op_linear(const half_t* A, long lda, const half_t* W, const float* bias, half_t* out, long ldo,
              int M, int N, int K, const half_t* res, long ldres, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Flash attention, one (query tile, head, batch) per workgroup, online softmax in fp32.
//   Q [B*Lq][ldq] (head h at column h*D), K [B*Lk][ldk], Vt [B][heads*D][Lkpad] (V transposed),
//   O [B*Lq][ldo].
// ------------------------------------------------------------------------------------------
typedef ctrl_attn_desc AttnArgs;
int op_flash_attn(const AttnArgs& a, hipStream_t s);
int op_flash_attn_group(const AttnArgs* a, int n, hipStream_t s);      // same-shape problems in one launch (see op_igemm_group)
// Grouped launch of the attention kernels (attention.hip, attention_d64.hip): up to kMaxGroup problems of one shape -- the attentions
// of sibling adapter blocks -- share a launch; the descriptors travel as an array at offset 0 of the kernel-argument segment,
// workgroup b works on problem b / per (per = the workgroups of one problem, a multiple of 8: the XCD of a workgroup stays its local
// index & 7).  A plain launch is a group of one.
struct AttnGroup { AttnArgs a[kMaxGroup]; };
#define ATTN_GROUP_ARGS(gp) (((const AttnArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gp])
// XCD-aware work map of the attention kernels (1-D grid of 8 * ceil(pairs / 8) * qtiles workgroups per problem; workgroup id -> XCD
// id & 7, the observed dispatch order): XCD x owns the whole (batch, head) pairs x, x + 8, ... and walks their query tiles back to back, so the
// workgroups resident on one XCD stream the SAME K / V tiles through that XCD's L2.  Round 6: when pairs is not a multiple of 8 the LAST
// round used to leave 8 - pairs % 8 XCDs without work (B = 2, 5 heads: 10 pairs in 16 slots -- the config-1 step ran its self-attention at 0.29
// of the peak instead of 0.39); the query tiles of those last pairs are now dealt round-robin over all eight XCDs.  false = no work.
#ifdef __HIPCC__
// max of two scores as IEEE-754-2019 maximum (v_maximum3_f32 on gfx950): fmaxf() is maxnum, for which hipcc first QUIETS every operand it has
// not seen produced -- a canonicalising v_max_f32 x, x, x per MFMA output, 5 extra VALU per 64-key tile beside the MFMAs of the attention
// loops (round 6, profiles/r06_attention_pmc.txt).  Same value for every non-NaN input.
#ifdef CTRL_ATTN_FMAXF          // (build.py CTRL_BUILD_FMAXF=1: the rounds 2-5 form, for A/B runs)
__device__ __forceinline__ float vmaxf(float a, float b) { return fmaxf(a, b); }
#else
__device__ __forceinline__ float vmaxf(float a, float b) { return __builtin_elementwise_maximum(a, b); }
#endif
__host__ __device__ __forceinline__ bool attn_work_map(int gbid, int qtiles, int pairs, int* pair, int* qt) {
    const int xcd = gbid & 7, j = gbid >> 3;
    const int full = pairs & ~7;
    const int jr = j - (full >> 3) * qtiles;          // row inside the remainder round (negative before it)
    if (jr < 0) {
        const int r = j / qtiles;
        *pair = r * 8 + xcd;
        *qt = j - r * qtiles;
        return true;
    }
    const int t = jr * 8 + xcd;
    if (t >= (pairs - full) * qtiles) return false;
    const int r = t / qtiles;
    *pair = full + r;
    *qt = t - r * qtiles;
    return true;
}
#endif
// descriptors of the grouped launch being dispatched (op_flash_attn_group); [0] is the problem the dispatcher sees
extern thread_local const AttnArgs* t_attn_grp;
extern thread_local int t_attn_grp_n;
inline int attn_grp_count() { return t_attn_grp ? t_attn_grp_n : 1; }
inline AttnGroup attn_grp_make(const AttnArgs& a) {
    AttnGroup g;
    for (int i = 0; i < attn_grp_count(); ++i) g.a[i] = t_attn_grp ? t_attn_grp[i] : a;
    return g;
}
// instruction-selection variant of the head_dim-64 long-sequence kernel (attention_d64.hip); performance only
int attn_set_variant(int v);

// temporal attention over the frame axis: tokens X [(b*F+f)*HW + p][...]; seq = F (<= 32)
typedef ctrl_tattn_desc TAttnArgs;
int op_temporal_attn(const TAttnArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Normalisation
// ------------------------------------------------------------------------------------------
// GroupNorm statistics over [img][rows_per_img][C]: stats[img][G][2] = (sum, sumsq), reduced in a fixed order.
// `stats` holds op_gn_stats_floats(...) floats (results first, then tickets and per-workgroup partials) and must have
// been zeroed once (the kernel leaves its tickets zero again).
size_t op_gn_stats_floats(int imgs, int rows_per_img, int C, int G);
int op_gn_stats(const void* x, int x_dtype, float* stats, int imgs, int rows_per_img, int C, int G, hipStream_t s);
// y = (x-mean)*rstd*gamma+beta (optionally SiLU); x,y [imgs*rows][C]
// ldy / lo_off: row stride of y (0 = C) and, when > 0, the column offset of the LOW half of a split operand
// (y[r][c] = hi = fp16(v), y[r][lo_off + c] = fp16(v - hi)): consumed by a convolution packed with dup weights
int op_gn_apply(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, half_t* y,
                int imgs, int rows_per_img, int C, int G, float eps, int silu, hipStream_t s, long ldy = 0, int lo_off = 0,
                long stat_rows = 0, long y_img_rows = 0, long y_row0 = 0);
//   stat_rows: rows per image the statistics in `stats` span (0 = rows_per_img; larger when a clip's frames are sharded
//   over ranks and the sums were all-reduced); y_img_rows / y_row0: image i of y starts at row i*y_img_rows + y_row0
//   (0 = dense): the padded operand layout of the frame-sharded temporal convolution
// LayerNorm over the last dim of x [M][C] -> y fp16
int op_layernorm(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, half_t* y, long ldy,
                 int M, int C, float eps, hipStream_t s);
// xsum[m][c] = x[m][c] + addv[((m / rows_per_img) % vmod) * ldv + c] (in x's dtype, row stride ldx) and y = LayerNorm(xsum):
// the frame-index embedding add in front of the temporal transformer and its first LayerNorm in one pass
int op_layernorm_add(const void* x, int x_dtype, long ldx, const float* addv, long ldv, int rows_per_img, int vmod, void* xsum,
                     const float* gamma, const float* beta, half_t* y, long ldy, int M, int C, float eps, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Element-wise / layout / small kernels
// ------------------------------------------------------------------------------------------
// [N][C][HW] (any dtype) -> [N][HW][C] fp16
int op_nchw_to_nhwc(const void* x, int dtype, half_t* y, int N, int C, int HW, hipStream_t s);
// [N][HW][C] fp16 -> [N][C][HW] (any dtype), y = x*scale
int op_nhwc_to_nchw(const half_t* x, void* y, int dtype, int N, int C, int HW, float scale, hipStream_t s,
                    const int* img_map = nullptr);   // img_map (device): image n is written at image img_map[n]
// exact kxk mean pooling of an NCHW tensor (adaptive_avg_pool2d with integer ratio), dtype preserved
int op_avgpool_nchw(const void* x, void* y, int dtype, int NC, int Hin, int Win, int Hout, int Wout, hipStream_t s);
// sinusoidal timestep embedding, flip_sin_to_cos=True, shift 0: out[n][dim] fp32 = [cos | sin]
int op_timestep_sincos(const float* t, int t_count, float* out, int N, int dim, hipStream_t s);
// frame-index variant: t[n] = n % F
int op_frameidx_sincos(float* out, int N, int F, int dim, hipStream_t s);
// small-M linear (M <= 64): out[m][n] = act(sum_k in_act(x[m][k]) * w[n][k] + b[n]); x,out fp32; w fp16
//   in_silu: apply SiLU to x on load; out_silu: apply SiLU to the result
int op_linear_small(const float* x, long ldx, const half_t* w, const float* b, float* out, long ldo,
                    int M, int N, int K, int in_silu, int out_silu, hipStream_t s);
// grouped form: `count` independent small-M linears, kSmallGroup per launch
struct SmallLin { const float* x; long ldx; const half_t* w; const float* b; float* out; long ldo; int M, N, K, in_silu, out_silu; };
constexpr int kSmallGroup = 40;
struct SmallLinGroup { SmallLin p[kSmallGroup]; int blk_begin[kSmallGroup]; int count; };
int op_linear_small_group(const SmallLin* probs, int count, hipStream_t s);
// y[m][c] = a*x1[m][c] + b*x2[m][c]   (AlphaBlender; a=alpha, b=1-alpha read from device: alpha=sigmoid(*mix))
int op_blend(const void* x_spatial, int xs_dt, const void* x_temporal, int xt_dt, const float* mix_factor, void* y, int y_dt, size_t n, hipStream_t s);
// y[m][c] = x[m][c] + v[(m / rows_per_img) % vmod][c]   (fp32 per-image vector broadcast add)
int op_add_rowvec(const void* x, int x_dt, const float* v, long ldv, void* y, int y_dt, size_t M, int C, int rows_per_img, int vmod, hipStream_t s);
// rows (clip, frame, pixel): y = x + v[(b*L + p) % B]  (per-clip time context of the temporal transformer, see the kernel)
int op_add_rowvec_clip(const void* x, int x_dt, const float* v, long ldv, void* y, int y_dt, int B, int F, int L, int C, hipStream_t s);
// nearest x2 up-sampling of an NHWC fp16 map
int op_upsample2x_nhwc(const half_t* x, half_t* y, int N, int H, int W, int C, hipStream_t s);
// fill fp16/any
int op_fill_zero(void* p, size_t bytes, hipStream_t s);
int op_fill_zero_group(void* const* p, const size_t* bytes, int n, hipStream_t s);      // up to kMaxGroup fills in one launch
// K-way weighted merge of NCHW tensors: out = sum_e w[widx[e]] * x_e   (router merge; weights on device)
int op_weighted_merge(const void* const* xs_dev, const float* w, const int* widx_dev, int K, void* out, int dtype,
                      size_t n, hipStream_t s);
// router: weights[r][e] = softmax_e( wg[r][e] (or 0) - 1e6*(1-mask[e]) ),  r in [0,R)
int op_router_softmax(const float* wg, const int* mask, float* out, int R, int E, int equal_weights, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Direct 3x3 convolution for tiny channel counts (ControlNet stem / conditioning embedder head)
//   in : NCHW any dtype (in_nchw=1) or NHWC fp16;  w fp32 [9][Cin][Cout]; out NHWC fp16 (+bias, optional SiLU)
// ------------------------------------------------------------------------------------------
int op_conv3x3_direct(const void* in, int in_dtype, int in_nchw, const float* w, const float* bias, half_t* out,
                      int N, int Cin, int Cout, int Hin, int Win, int stride, int silu, hipStream_t s);

// The same convolution on the matrix cores for the channels-last layers with Cin, Cout in {16, 32} (the conditioning embedder's
// 16 -> 16, 16 -> 32 s2, 32 -> 32): x NHWC fp16, w fp16 [Cout][9][Cin] (op_pack_conv_w), out NHWC fp16 (+ bias, optional SiLU)
bool op_conv3x3_small_mfma_applies(int Cin, int Cout);
int op_conv3x3_small_mfma(const half_t* x, const half_t* w, const float* bias, half_t* out, int N, int Cin, int Cout, int Hin, int Win,
                          int stride, int silu, hipStream_t s);

// weight packing (load time)
// conv weight [Cout][Cin][kh][kw] (any dtype) -> fp16 [Cout][kh*kw][Cin]   (taps-major K)
int op_pack_conv_w(const void* w, int dtype, half_t* out, int Cout, int Cin, int taps, hipStream_t s, int* ovf = nullptr);
//   ovf (optional, device): set to 1 when a value does not fit fp16 (|w| > 65504, inf, nan)
// split-operand form: fp16 [Cout][taps][2*Cin], the Cin weights of every tap twice
int op_pack_conv_w_dup(const void* w, int dtype, half_t* out, int Cout, int Cin, int taps, hipStream_t s, int* ovf = nullptr);
// conv weight [Cout][Cin][3][3] -> fp32 [9][Cin][Cout] for the direct kernel
int op_pack_conv_w_direct(const void* w, int dtype, float* out, int Cout, int Cin, hipStream_t s);
// linear weight [N][K] -> fp16 [N][K]; optional GEGLU interleave of rows (N = 2*inner)
int op_pack_linear_w(const void* w, int dtype, half_t* out, int N, int K, int geglu, hipStream_t s, int* ovf = nullptr);
// vector -> fp32 (optional GEGLU interleave)
int op_pack_vec(const void* v, int dtype, float* out, int N, int geglu, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Conditioning-image preparation (model/ctrl_helper.py:268-296): Pillow-exact 8-bit Lanczos resize + /255 + NCHW + repeats
//   src uint8 [F][Hin][Win][3]; h/v bounds int32 [out][2] + weights int32 [out][ks] (null = axis not resampled);
//   tmp uint8 [F][Hin][W][3] (horizontal pass result); out [cfg][rep*F][3][H][W] in out_dtype
// ------------------------------------------------------------------------------------------
int op_prepare_images(const unsigned char* src, int F, int Hin, int Win, const int* hbounds, const int* hk, int hks,
                      const int* vbounds, const int* vk, int vks, unsigned char* tmp, void* out, int out_dtype,
                      int W, int H, int rep, int cfg, hipStream_t s);


// ------------------------------------------------------------------------------------------
// Grouped launches over sibling problems (round 5).  The adapter blocks of one pyramid level are independent and have identical
// shapes (model/ctrl_adapter.py:181-191 simply runs them one after the other): the plan records the op sequence of every sibling and
// replays the sequences in lock-step; while a collector is installed, the ops below deposit their arguments instead of launching,
// and flush() launches what was deposited for one step -- as ONE grouped launch when the problems agree in shape and form,
// one by one otherwise.  Every problem is computed exactly as it would be alone: results are bit-identical to the ungrouped forward.
// ------------------------------------------------------------------------------------------
struct GnStatsArgs { const void* x; int x_dtype; float* stats; int imgs, rows_per_img, C, G; };
struct GnApplyArgs { const void* x; int x_dtype; const float* stats; const float* gamma; const float* beta; half_t* y;
                     int imgs, rows_per_img, C, G; float eps; int silu; long ldy; int lo_off; long stat_rows, y_img_rows, y_row0; };
struct LnArgs { const void* x; int x_dtype; long ldx; const float* addv; long ldv; int rows_per_img, vmod; void* xsum;
                const float* gamma; const float* beta; half_t* y; long ldy; int M, C; float eps; };
int op_gn_stats_group(const GnStatsArgs* a, int n, hipStream_t s);
int op_gn_apply_group(const GnApplyArgs* a, int n, hipStream_t s);
int op_layernorm_group(const LnArgs* a, int n, hipStream_t s);
// GroupNorm(32) of a SMALL map (the 80-channel slice of one image <= 512 KB) in ONE launch: statistics + apply by the same workgroup
// (norm.hip: gn_fused_kernel); op_gn_fused_applies says whether the plans use it for a problem (only with CTRL_GN_FUSED=1: measured
// slower than the two-kernel form in the step) -- op_gn_fused itself takes every small map
bool op_gn_fused_fits(int x_dtype, int rows_per_img, int C, int G);
bool op_gn_fused_applies(int x_dtype, int rows_per_img, int C, int G);
int op_gn_fused(const void* x, int x_dtype, const float* gamma, const float* beta, half_t* y, int imgs, int rows_per_img, int C, int G,
                float eps, int silu, hipStream_t s, long ldy = 0, int lo_off = 0);
int op_gn_fused_group(const GnApplyArgs* a, int n, hipStream_t s);

struct OpCollector {
    enum { NONE = 0, IGEMM, GN_STATS, GN_APPLY, LAYERNORM, ATTN, GN_FUSED, FILL, FFN };
    int type = NONE, n = 0;
    hipStream_t s = nullptr;
    IGemmArgs ig[kMaxGroup];
    GnStatsArgs gs[kMaxGroup];
    GnApplyArgs ga[kMaxGroup];
    LnArgs ln[kMaxGroup];
    AttnArgs at[kMaxGroup];
    FfnArgs ff[kMaxGroup];
    struct Fill { void* p; size_t bytes; } fz[kMaxGroup];
    // make room for one more problem of `type` on stream `s` (flushes what is pending when it cannot join); returns its index or -1
    int slot(int type_, hipStream_t s_, int* rc);
    int flush();
};
extern thread_local OpCollector* t_collect;
// CTRL_GROUP=0 turns grouped launches off (every problem in its own launch, as in rounds 1-4); 2 keeps them but lets the GEMM
// dispatcher choose the tile as if every problem ran alone (then results are bit-identical to the ungrouped forward: the tests)
bool group_launches_enabled();
bool group_tiles_as_alone();
