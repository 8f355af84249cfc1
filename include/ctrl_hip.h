/* libctrlhip -- C ABI of the MI355X-native (gfx950) Ctrl-Adapter denoising hot path.
 *
 * Drop-in boundary for the per-timestep ControlNet forward + Ctrl-Adapter forward (+ router / merge) of
 * HL-hanlin/Ctrl-Adapter.  Every entry point takes plain device pointers and sizes (no torch types),
 * enqueues hand-written HIP kernels on the caller's stream (`stream` is a hipStream_t passed as void*;
 * NULL = default stream), never synchronises, and returns 0 on success / non-zero on error with the
 * message available from ctrl_last_error().  No C++ exception crosses this boundary.
 *
 * Plan-level entry points (what a reference-side binding calls):
 *   ctrl_controlnet_forward   replaces ControlNetModel.forward            controlnet/controlnet.py:662-881
 *   ctrl_adapter_forward      replaces ControlNetAdapter.forward          model/ctrl_adapter.py:171-224
 *                             (and AdapterSpatioTemporal.forward          model/adapter_spatial_temporal.py:175-292,
 *                              ResnetBlock2D.forward                      model/resnet_block_2d.py:164-221)
 *   ctrl_router_weights       replaces ControlNetRouter.forward           model/ctrl_router.py:85-112
 *   ctrl_router_merge         replaces the caller-side expert merge       i2vgen_xl/pipelines/
 *                             i2vgen_xl_controlnet_adapter_pipeline.py:1000-1022 (and train.py:1262-1276)
 *   ctrl_avgpool_nchw         replaces F.adaptive_avg_pool2d(latents,(64,64))  sdxl/pipelines/
 *                             sdxl_controlnet_adapter_pipeline.py:1306-1309
 * Op-level entry points (ctrl_op_*) expose the individual kernels for parity tests and micro-benchmarks.
 */
#ifndef CTRL_HIP_H
#define CTRL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTRL_ABI_VERSION 7

/* element types of boundary tensors */
enum { CTRL_F32 = 0, CTRL_F16 = 1, CTRL_BF16 = 2 };

/* ---------------------------------------------------------------- runtime */
int ctrl_abi_version(void);
const char* ctrl_last_error(void);           /* thread-local, valid until the next failing call */
/* per-kernel-class HIP-event profiler: begin -> run forwards -> end(sync) -> query */
int ctrl_prof_begin(void);
int ctrl_prof_end(void);
int ctrl_prof_count(void);
/* flops / bytes: ALGORITHMIC work summed over the class's launches (2*M*N*K, 4*B*h*Lq*Lk*D, one read + one write, ...) */
int ctrl_prof_get(int i, char* name, int name_len, double* total_ms, int* launches, double* flops, double* bytes);
/* per-launch records of the last finished profile, in launch order: kernel class tag, kernel symbol with template
   arguments (as rocprofv3 prints it), shape note, HIP-event time on the launch stream, algorithmic work, grid size in
   work-items (= rocprofv3's Grid_Size, which tells the shapes of one symbol apart) */
int ctrl_prof_launch_count(void);
int ctrl_prof_launch_get(int i, char* tag, int tag_len, char* symbol, int symbol_len, char* detail, int detail_len,
                         double* ms, double* flops, double* bytes, int64_t* grid_threads);

/* ---------------------------------------------------------------- op level */
typedef struct ctrl_igemm_seg {
    void* out;          /* destination */
    int64_t ld;         /* ROW: row stride (elements); TRANSPOSED: padded token stride */
    int32_t col_begin;  /* first output column of the segment (multiple of 16) */
    int32_t ncols;      /* segment width */
    int32_t fmt;        /* 0 ROW: out[m*ld + c];  1 TRANSPOSED: out[((m/L)*ncols + c)*ld + m%L] */
    int32_t dtype;      /* CTRL_F32 / CTRL_F16 / CTRL_BF16 */
    int32_t L;          /* tokens per image (TRANSPOSED) */
    int32_t pad_;
    const int32_t* img_map;   /* TRANSPOSED, optional (device): image m/L is written at image img_map[m/L] (frame scatter) */
} ctrl_igemm_seg;

typedef struct ctrl_igemm_desc {
    const void* A;      /* fp16 activations, channels-last */
    int64_t lda;        /* elements between consecutive pixels/rows */
    int32_t mode;       /* 0 rows (linear / 1x1), 1 conv2d, 2 temporal (3 frame taps) */
    int32_t Cin;        /* K per tap */
    int32_t taps;       /* 1 | 9 | 3 */
    int32_t Hin, Win, Hout, Wout, stride, up;   /* conv2d geometry; up = nearest up-sampling folded into the gather */
    int32_t F, HW;      /* temporal: frames per clip, rows per frame */
    int32_t t_pad;      /* temporal, frame-sharded clip: A = padded [clip][F+2][HW][lda] (slots 0 / F+1 = halo frames), M rows = the F local frames */
    const void* W;      /* fp16 [Nout][taps*Cin] */
    int32_t M, Nout, Ktot;
    int32_t pad1_;
    const float* bias;  /* [Nout] or NULL */
    const float* rowvec; int32_t rowvec_ld; int32_t rows_per_img;   /* + rowvec[(m/rows_per_img)*ld + n] */
    const void* res; int64_t ldres;                                 /* + res[m*ldres + n] (fp16, or fp32 if res_f32) */
    float scale;
    int32_t geglu;      /* weights packed in interleaved (hidden,gate) 16-column blocks; out width Nout/2 */
    int32_t nseg;
    int32_t act;        /* 0 none, 1 SiLU applied after bias/rowvec (before residual) */
    int32_t res_f32;    /* residual is fp32 (the fp32 residual stream) */
    int32_t a_split;    /* split operand: A rows hold [hi | lo] fp16 halves of an fp32 operand along Cin (lo = fp16(x - hi)), so the
                           product is exact in A to ~2^-22 at twice the MFMA work (the ControlNet's convolutions).
                           1: W = every tap's Cin/2 weights packed twice ([Cout][taps][Cin]), K walked as one long axis;
                           2: W = the plain pack [Cout][taps][Cin/2]; k-tiles alternate hi / lo chunk of the same channels and
                              each weight tile is staged once per pair (Cin/2 % 64 == 0) */
    void* splitk_ws; int64_t splitk_ws_bytes;   /* optional fp32 scratch: enables split-K for small-M / long-K problems */
    int32_t* splitk_tickets;   /* optional (device): ZEROED ticket words for the in-launch reduction -- one per 256-row output tile for 2..4 splits, five per tile
                                  for the two-level reduction of 5..16 splits (ctrl-adapter_amd/csrc/igemm.hip: igemm_splitk_ticket_words) (the
                                  plans hand out slices of a pool they zero once per forward); NULL = the library zeroes ticket words
                                  at the end of splitk_ws with a fill launch of its own */
    void* out16; int64_t ld16;   /* optional fp16 row-major mirror of the (single, row-major) output: GEMM-operand copy of an fp32 stream */
    /* optional AlphaBlender fold (diffusers AlphaBlender, model/adapter_spatial_temporal.py:229,282), row-major outputs
     * only: out = (1-a) * y + a * blend_x[m*ld_blend + n], a = sigmoid(*blend_mix), y = the epilogue result above */
    const float* blend_mix; const void* blend_x; int64_t ld_blend; int32_t blend_f32;
    int32_t out16_lo_off;   /* > 0: the fp16 mirror is a split operand, out16[m*ld16 + n] = hi, out16[m*ld16 + out16_lo_off + n] = lo */
    float scale2; int32_t scale2_from;   /* scale2_from > 0: output columns scale2_from <= n < scale2_to are multiplied by scale2 instead
                                            of scale: the K third of a Q|K|V projection leaves pre-multiplied by
                                            softmax_scale * log2(e) for ctrl_attn_desc::k_prescaled */
    int32_t res_up;     /* 2: conv2d only -- `res` holds [img][Hout/2][Wout/2][ldres] and is read through a nearest x2 up-sampling
                           (res[img][oy/2][ox/2]); 0 | 1: res[m*ldres + n] */
    int32_t scale2_to;  /* end of the scale2 column range (0 = Nout) */
    /* Output segments (1..3, ascending col_begin).  Mixed layouts -- row-major segments followed by ONE transposed last segment,
       the Q | K | V^T outputs of a self-attention projection in one launch -- run on the vector epilogue when the transposed
       segment starts at a multiple of 64 columns (the dispatcher then only picks tiles whose width divides that boundary, so
       every tile lies in one segment); anything else takes the scalar epilogue. */
    ctrl_igemm_seg seg[3];
    int32_t* nonfinite;  /* optional (device): set to 1 when an fp16 value this launch writes is inf / nan, i.e. an activation left the
                            fp16 range (|x| > 65504).  NULL = no check; op_igemm fills it in itself while the range check is on
                            (ctrl_range_check / CTRL_CHECK_FINITE=1) */
} ctrl_igemm_desc;
int ctrl_op_igemm(const ctrl_igemm_desc* d, void* stream);
/* Fused GEGLU feed-forward of a transformer block with dim 512 / hidden 2048 (csrc/ffn.hip; diffusers FeedForward(GEGLU) under
   BasicTransformerBlock / TemporalBasicTransformerBlock, model/adapter_spatial_temporal.py:108-130):
       out = epilogue( ((X W1h^T + b1h) * gelu_erf(X W1g^T + b1g)) W2^T )
   in ONE launch -- the [M][2048] hidden activation never reaches HBM.  X fp16 [M][ldx] (the LayerNorm output), W1 / b1 the GEGLU-
   interleaved pack of ff.net.0.proj (ctrl_op_pack_linear_w / ctrl_op_pack_vec with geglu = 1: [4096][512], [4096]), W2p the pack of
   ff.net.2 re-ordered along K by ctrl_op_ffn_pack_w2, `out` the descriptor of the OUTPUT GEMM ([M][512]: bias, res / ldres / res_f32,
   seg[0] row-major, out16 ...; its A / W / K fields are ignored).  ctrl_op_ffn_pack_w2: fp16 [512][2048] linear pack -> fp16 [512][2048]. */
typedef struct ctrl_ffn_desc {
    const void* X; int64_t ldx;
    const void* W1; const float* b1;
    const void* W2p;
    ctrl_igemm_desc out;
} ctrl_ffn_desc;
int ctrl_op_ffn(const ctrl_ffn_desc* d, void* stream);
int ctrl_op_ffn_pack_w2(const void* w2_packed, void* out, int N, int K, void* stream);
/* Range safety of the fp16 activations (debug aid for real checkpoints: the synthetic N(0, 0.02^2) weights never leave the fp16 range).
   ctrl_range_check(1) -- or CTRL_CHECK_FINITE=1 in the environment -- makes every GEMM / convolution epilogue OR "this fp16 value is
   inf / nan" into a per-device flag word; ctrl_range_check(0) turns it off, ctrl_range_check(-1) only queries.  ctrl_range_status(reset)
   synchronises the current device and returns 1 when the flag was raised since the last reset (2 = the check is off).  The Python
   mirrors call it after every forward while the check is on and raise RuntimeError. */
int ctrl_range_check(int on);
int ctrl_range_status(int reset);
/* Walk order of the GEMM's output tiles over the 8 XCDs (csrc/tile_order.h; performance only, results are identical):
   "auto" (default; also CTRL_IGEMM_ORDER: grouped walk for row GEMMs whose weights exceed an XCD's L2 share), "legacy", or
   forced "m,G" / "n,G" (XCDs split the activation panels / the weight panels, weight panels walked in groups of G tiles).
   0 = accepted.
   ctrl_igemm_tile_of: the (tile_m, tile_n) workgroup `bid` of an ntm x ntn grid computes under (mode 0|1|2, group) --
   a diagnostic the host tests use to prove every order is a bijection. */
int ctrl_igemm_set_order(const char* spec);
/* Which problems run on the 8-phase wide-tile kernel (csrc/igemm.hip: igemm8_kernel, 256 x 256|320 x 64 tiles): 0 none (the round-3
   ring kernels), 1 wherever the grid fills the chip (default; also CTRL_IGEMM8=0|1|force), 2 every eligible problem whatever its
   size -- what the parity tests use to drive their small shapes through it.  -1 restores the default.  Performance only: both
   kernel families share one epilogue and the same accumulation order inside a k-tile; results may differ in the last fp32 bit
   between families (different k-tile depth).  0 = accepted. */
int ctrl_igemm_set_wide(int mode);
/* Grouped launches (csrc/ops.h: OpCollector): the sibling adapter blocks of one pyramid level -- equal shapes, independent
   (model/ctrl_adapter.py:181-191) -- are replayed in lock-step and every GEMM / GroupNorm / LayerNorm / attention of theirs leaves as
   ONE launch over 2-4 problems.  1 (default; also CTRL_GROUP) on: the GEMM dispatcher sizes its tile for the whole group (a 32^2 level
   reaches the wide tile), so results equal the one-by-one forward within the tile families' last-bit differences, like another batch
   size; 2 on with the tile every problem would get alone: bit-identical to the one-by-one forward; 0 off; -1 queries.  Returns the
   mode.  Every mode is bit-reproducible run to run. */
int ctrl_group_launches(int on);
/* Run-time policy table (csrc/policy.h): every CTRL_* environment variable the library understands is read ONCE, at the first query, into
   one table -- no dispatcher, plan or op calls getenv().  Test / experiment ABI: ctrl_policy_set(name, value) overrides an entry for the
   process (value NULL = unset; 0 = accepted, 1 = unknown name), ctrl_policy_get(name) returns the value in force (NULL = unset or unknown),
   ctrl_policy_count / ctrl_policy_name(i) list the names (a bench line can record what a run was taken with).  Not to be called
   concurrently with a forward. */
int ctrl_policy_set(const char* name, const char* value);
const char* ctrl_policy_get(const char* name);
int ctrl_policy_count(void);
const char* ctrl_policy_name(int i);
int ctrl_igemm_tile_of(int bid, int ntm, int ntn, int mode, int group, int* tile_m, int* tile_n);

typedef struct ctrl_attn_desc {
    const void* Q; int64_t ldq;      /* [B*Lq][ldq] fp16, head h at column h*D */
    const void* K; int64_t ldk;      /* [kvB*Lk][ldk] */
    const void* Vt; int32_t Lkpad; int32_t kvB;     /* [kvB][heads*D][Lkpad], Lkpad % 64 == 0; kvB = B, or 1 = K/V shared by all batches */
    void* O; int64_t ldo;            /* [B*Lq][ldo] */
    int32_t B, heads, D, Lq, Lk;
    float scale;
    int32_t k_prescaled;             /* K rows already hold k * scale * log2(e) (applied in fp32 by the producing GEMM's epilogue,
                                        before the one rounding to fp16): the kernel then folds the running maximum into the
                                        accumulator initialisation of the QK^T MFMA and exponentiates its output directly */
    int32_t pad0_;
} ctrl_attn_desc;
int ctrl_op_flash_attn(const ctrl_attn_desc* d, void* stream);
/* Instruction-selection variant of the head_dim-64 long-sequence kernel (csrc/attention_d64.hip): 0 = the round-2 kernel,
   > 0 = the round-3 family (what each number selects is listed in that file); performance only, every variant computes the
   same function.  -1 = back to the default (CTRL_ATTN_VARIANT or the best measured).  Used by tools/attn_bench.cpp. */
int ctrl_attn_set_variant(int v);
/* The attention kernels' workgroup -> (batch * heads + head, query tile) map, evaluated on the host (CPU tests: every (pair, tile) exactly once,
   whole pairs per XCD, the last partial round balanced over the eight XCDs -- csrc/ops.h attn_work_map).  Returns 1 and fills pair / qtile when
   workgroup `gbid` of a grid of 8 * ceil(pairs / 8) * qtiles has work, 0 when it exits at once. */
int ctrl_attn_work_map(int gbid, int qtiles, int pairs, int* pair, int* qtile);

typedef struct ctrl_tattn_desc {
    const void* Q; int64_t ld;       /* [(b*Fq+f)*HW + p][ld], head h at column h*64; unsharded: rows are q | k | v, 3*C wide */
    void* O; int64_t ldo;            /* [rows][C] */
    int32_t Bc, F, HW, heads;        /* clips, key frames per clip (<= 32), pixels, heads of 64 */
    float scale;
    int32_t Fq;                      /* query frames per clip held in Q / O (0 = F) */
    /* keys / values: NULL = the k | v columns of the Q rows.  Otherwise K|V rows [((r*Bc + b)*Fl + fl)*HW + p][ldkv]
       (k at column h*64, v at C + h*64) for key frame kf = r*Fl + fl: the all-gathered K|V of a clip whose frames are
       sharded over ranks, Fl frames per rank */
    const void* KV; int64_t ldkv;
    int32_t Fl;                      /* key frames per gathered shard (0 = F) */
    int32_t pad0_;
} ctrl_tattn_desc;
int ctrl_op_temporal_attn(const ctrl_tattn_desc* d, void* stream);

/* x_dtype: CTRL_F16 or CTRL_F32 (fp32 residual stream).  GroupNorm statistics are reduced in a fixed order (no
 * floating-point atomics): `stats` holds ctrl_op_gn_stats_floats(...) floats - [imgs][G][2] (sum, sumsq) results first,
 * then scratch - and must have been zeroed once; the kernel leaves its scratch tickets zero again. */
size_t ctrl_op_gn_stats_floats(int imgs, int rows_per_img, int C, int G);
int ctrl_op_gn_stats(const void* x, int x_dtype, float* stats, int imgs, int rows_per_img, int C, int G, void* stream);
int ctrl_op_gn_apply(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, void* y,
                     int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream);
/* split-operand variant: y rows are [hi | lo] (row stride ldy, lo at column offset lo_off), hi + lo = the fp32 result to ~2^-22 */
int ctrl_op_gn_apply_split(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, void* y,
                           int64_t ldy, int lo_off, int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream);
/* GroupNorm(32) of a small map in ONE launch (statistics + apply by the same workgroup; qualifies when the 80-channel slice of one
 * image is at most 512 KB: ctrl_op_gn_fused_applies): ldy = row stride of y (0 = C), lo_off > 0 = split [hi | lo] result as above.
 * The plans use it only with CTRL_GN_FUSED=1: measured inside the step it is slower than the two launches it replaces. */
int ctrl_op_gn_fused_applies(int x_dtype, int rows_per_img, int C, int G);
int ctrl_op_gn_fused(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int64_t ldy, int lo_off,
                     int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream);
int ctrl_op_layernorm(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                      int M, int C, float eps, void* stream);
int ctrl_op_nchw_to_nhwc(const void* x, int dtype, void* y, int N, int C, int HW, void* stream);
int ctrl_op_nhwc_to_nchw(const void* x, void* y, int dtype, int N, int C, int HW, float scale, void* stream);
int ctrl_avgpool_nchw(const void* x, void* y, int dtype, int NC, int Hin, int Win, int Hout, int Wout, void* stream);
int ctrl_op_timestep_sincos(const float* t, int t_count, float* out, int N, int dim, void* stream);
int ctrl_op_linear_small(const float* x, int64_t ldx, const void* w, const float* b, float* out, int64_t ldo,
                         int M, int N, int K, int in_silu, int out_silu, void* stream);
int ctrl_op_blend(const void* x_spatial, int xs_dtype, const void* x_temporal, int xt_dtype, const float* mix_factor,
                  void* y, int y_dtype, size_t n, void* stream);
int ctrl_op_add_rowvec(const void* x, int x_dtype, const float* v, int64_t ldv, void* y, int y_dtype, size_t M, int C,
                       int rows_per_img, int vmod, void* stream);
int ctrl_op_conv3x3_direct(const void* in, int in_dtype, int in_nchw, const float* w, const float* bias, void* out,
                           int N, int Cin, int Cout, int Hin, int Win, int stride, int silu, void* stream);
/* the same 3x3 convolution (pad 1, stride 1|2, + bias, optional SiLU) on the matrix cores for channels-last fp16 input with Cin, Cout in
 * {16, 32}: w fp16 [Cout][9][Cin] (ctrl_op_pack_conv_w with taps = 9) */
int ctrl_op_conv3x3_small_mfma(const void* x, const void* w, const float* bias, void* out, int N, int Cin, int Cout, int Hin, int Win,
                               int stride, int silu, void* stream);
/* Conditioning-image preparation feeding the path (model/ctrl_helper.py:268-296 prepare_images = per frame diffusers
 * VaeImageProcessor(do_convert_rgb, no normalisation).preprocess -> batch repeat -> CFG duplication), bit-exact with Pillow's
 * 8-bit Lanczos resampling: src uint8 RGB [F][Hin][Win][3]; per resampled axis the bounds int32 [out][2] and the 22-bit
 * fixed-point weights int32 [out][ks] of Pillow's precompute_coeffs (device; NULL when the axis keeps its size); tmp uint8
 * [F][Hin][W][3]; out [cfg][repeat*F][3][H][W] in out_dtype, out[g][r*F + f] = frame f / 255 */
int ctrl_prepare_images(const void* src_u8, int F, int Hin, int Win, const int32_t* hbounds, const int32_t* hk, int hks,
                        const int32_t* vbounds, const int32_t* vk, int vks, void* tmp_u8, void* out, int out_dtype,
                        int W, int H, int repeat, int cfg, void* stream);
/* load-time packers */
int ctrl_op_pack_conv_w(const void* w, int dtype, void* out, int Cout, int Cin, int taps, void* stream);
/* weights of a split-operand convolution: [Cout][taps][2*Cin], every tap's Cin weights twice (for the hi and the lo half) */
int ctrl_op_pack_conv_w_dup(const void* w, int dtype, void* out, int Cout, int Cin, int taps, void* stream);
int ctrl_op_pack_conv_w_direct(const void* w, int dtype, float* out, int Cout, int Cin, void* stream);
int ctrl_op_pack_linear_w(const void* w, int dtype, void* out, int N, int K, int geglu, void* stream);
int ctrl_op_pack_vec(const void* v, int dtype, float* out, int N, int geglu, void* stream);

/* ---------------------------------------------------------------- plan level */
/* A named view of one state-dict tensor (reference layout, device memory). */
typedef struct ctrl_tensor_ref {
    const char* name;
    const void* data;
    int32_t dtype;
    int32_t ndim;
    int64_t shape[6];
} ctrl_tensor_ref;

/* ---- ControlNet (SD-1.5 architecture family; controlnet/controlnet.py:179-438) ---- */
typedef struct ctrl_controlnet_config {
    int32_t in_channels;               /* 4 */
    int32_t conditioning_channels;     /* 3 */
    int32_t block_out_channels[4];     /* 320,640,1280,1280 */
    int32_t down_block_has_attn[4];    /* 1,1,1,0  (CrossAttnDownBlock2D x3, DownBlock2D) */
    int32_t layers_per_block;          /* 2 */
    int32_t num_attention_heads;       /* 8 (the reference's `attention_head_dim` naming quirk, :221-227) */
    int32_t cross_attention_dim;       /* 768 */
    int32_t cond_embed_channels[4];    /* 16,32,96,256 */
    float norm_eps;                    /* 1e-5 */
} ctrl_controlnet_config;

typedef struct ctrl_controlnet ctrl_controlnet;   /* opaque */

/* parameter inventory: names/shapes are the reference's state-dict keys (single source of truth for the
 * Python mirror).  Returns the number of parameters; fills name/shape for index i. */
int ctrl_controlnet_param_count(const ctrl_controlnet_config* cfg);
int ctrl_controlnet_param_spec(const ctrl_controlnet_config* cfg, int i, char* name, int name_len,
                               int64_t shape[6], int* ndim);
/* builds a plan: packs the weights (fp16, MFMA-friendly layouts) into memory owned by the plan */
int ctrl_controlnet_create(const ctrl_controlnet_config* cfg, const ctrl_tensor_ref* tensors, int n_tensors,
                           void* stream, ctrl_controlnet** out);
/* a second plan over the SAME packed weights (reference-counted) with its own workspace / streams / events: lets two forwards of one
   module be in flight at once (the mirror's batch lanes).  ctrl_controlnet_destroy frees it; the weights go with the last holder. */
int ctrl_controlnet_clone(ctrl_controlnet* h, ctrl_controlnet** out);
void ctrl_controlnet_destroy(ctrl_controlnet* h);

enum { CTRL_SKIP_CONV_IN = 1, CTRL_SKIP_TIME_EMB = 2, CTRL_GUESS_MODE = 4,
       /* step-invariant caching of the conditioning embedder (controlnet/controlnet.py:94-104): the condition image is the
        * same on every denoise step of a request.  KEEP = also store the embedder's last hidden map (256 ch at the latent
        * resolution) in plan-owned memory; REUSE = controlnet_cond is unchanged since the forward that stored it: start
        * from the stored map (only the final zero-initialised 3x3 conv runs).  Results are bit-identical either way. */
       CTRL_COND_KEEP = 8, CTRL_COND_REUSE = 16,
       /* everything on the launch stream, no auxiliary lane: for callers that already run this forward on a forked lane of their own
        * (MultiControlNetModel's per-net lanes) -- a fork inside a fork crashes hipGraph's end-of-capture on ROCm 7.2 */
       CTRL_NO_AUX_LANE = 32 };
/* sample [N][4][Hs][Ws], timesteps fp32 device [t_count] (1 or N), encoder_hidden_states [N][Lk][cross],
 * controlnet_cond [N][3][8*Hs][8*Ws]; outs[0..11] = down_block_res_samples, outs[12] = mid_block_res_sample,
 * all NCHW in out_dtype, already multiplied by conditioning_scale. */
int ctrl_controlnet_forward(ctrl_controlnet* h,
                            const void* sample, int sample_dtype, int N, int Hs, int Ws,
                            const float* timesteps, int t_count,
                            const void* encoder_hidden_states, int ehs_dtype, int Lk,
                            const void* controlnet_cond, int cond_dtype,
                            float conditioning_scale, int flags,
                            void* const* outs, int out_dtype, void* stream);

/* ---- Ctrl-Adapter (model/ctrl_adapter.py:17-116) ---- */
typedef struct ctrl_adapter_config {
    int32_t backbone_sdxl;             /* 1: up-sampling scale 2 (ctrl_adapter.py:61-66) */
    int32_t num_blocks;                /* layers per adapter block (configs use 1) */
    int32_t num_adapters_per_location; /* 1|2|3 */
    int32_t cross_attention_dim;       /* 2048 (sdxl) | 1024 (i2vgen-xl, svd) */
    int32_t add_spatial_resnet, add_temporal_resnet, add_spatial_transformer, add_temporal_transformer;
    int32_t loc_A, loc_B, loc_C, loc_D, loc_M;
} ctrl_adapter_config;

typedef struct ctrl_adapter ctrl_adapter;
int ctrl_adapter_param_count(const ctrl_adapter_config* cfg);
int ctrl_adapter_param_spec(const ctrl_adapter_config* cfg, int i, char* name, int name_len,
                            int64_t shape[6], int* ndim);
int ctrl_adapter_create(const ctrl_adapter_config* cfg, const ctrl_tensor_ref* tensors, int n_tensors,
                        void* stream, ctrl_adapter** out);
void ctrl_adapter_destroy(ctrl_adapter* h);
/* ins[0..11] down_block_res_samples, ins[12] mid (may be NULL) : NCHW [N][C_i][H_i][W_i] in in_dtype, with
 * (H_0,W_0) = (H0,W0) and the SD-1.5 pyramid below it.  outs[i] NCHW in out_dtype at the adapted
 * resolution (x2 for sdxl); slots without an adapter are zero-filled (zeros_like, ctrl_adapter.py:193).
 * outs[12] may be NULL.  encoder_hidden_states [ehs_batch][Lk][cross], ehs_batch = 1 (broadcast) or N. */
int ctrl_adapter_forward(ctrl_adapter* h,
                         const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                         const float* timesteps, int t_count,
                         const void* encoder_hidden_states, int ehs_dtype, int ehs_batch, int Lk,
                         void* const* outs, int out_dtype, void* stream);
/* The same forward with the pipelines' "sparse frames -> dense frames" scatter folded into the last epilogue of every
 * block (i2vgen_xl/pipelines/i2vgen_xl_controlnet_adapter_pipeline.py:1052-1071, svd/pipelines/
 * svd_controlnet_adapter_pipeline.py:719-741: torch.zeros + a python loop of per-frame copies): outs[i] hold N_out
 * frames, input frame j is written at frame frame_pos[j] (host array of N distinct positions < N_out), every other
 * frame is zero-filled.  The values are bit-identical to ctrl_adapter_forward's; only their location differs. */
int ctrl_adapter_forward_scatter(ctrl_adapter* h,
                                 const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                                 const float* timesteps, int t_count,
                                 const void* encoder_hidden_states, int ehs_dtype, int ehs_batch, int Lk,
                                 void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out, void* stream);

/* ---- Step-invariant text K/V cache: the to_k / to_v projections of encoder_hidden_states of every
 * cross-attention (Lk > 1) are kept in plan-owned buffers.  mode 1 (keep): the next forwards compute and store them (run
 * the first one eagerly, not under stream capture: it allocates); mode 2 (reuse): the projection GEMMs are skipped and
 * the stored K / V^T are read -- valid while encoder_hidden_states (values, batch, length) are unchanged, which the
 * caller guarantees; mode 0: off (default).  Results are bit-identical in all three modes. */
int ctrl_controlnet_text_cache(ctrl_controlnet* h, int mode);
int ctrl_adapter_text_cache(ctrl_adapter* h, int mode);

/* ---- Workspace hygiene.  A plan's workspace, text K/V buffers and conditioning cache grow by RETIRING the outgrown block
 * (never freeing it inside a forward: queued launches and captured hipGraphs keep valid addresses), so a server that sees
 * ever larger shapes keeps the sum of the earlier sizes.  ctrl_*_trim synchronises the device and frees the retired blocks;
 * call it when no captured graph that was recorded before the last growth will be replayed again.  A plan has no lock: trim
 * must not run concurrently with a forward of the same plan, and it is refused (non-zero) while a stream capture of that plan's
 * forward is still open (its device synchronisation would invalidate the capture). */
/* Which precision selection plan creation took for this checkpoint (a short key=value line): split-operand levels, whether outlier
   normalisation scales (max|gamma| / median|gamma| above the gate) switched it to the conservative selection, fp32 token-stream blocks of
   the adapter.  The Python mirrors expose it as `module.selection` and warn once when a conservative selection was taken. */
int ctrl_controlnet_selection(ctrl_controlnet* h, char* buf, int len);
int ctrl_adapter_selection(ctrl_adapter* h, char* buf, int len);
int ctrl_controlnet_trim(ctrl_controlnet* h);
int ctrl_adapter_trim(ctrl_adapter* h);

/* ---- One clip split across GPUs by frames (BASELINE.json config 4 with clips < GPUs).
 * Every rank holds Fl = F / world consecutive frames of every clip: `N` = clips * Fl LOCAL frames (frame-major, rank r
 * owns frames [r*Fl, (r+1)*Fl) of each clip), num_frames = Fl.  All ops of the adapter are per frame except three, and
 * each of those does exactly one exchange through the caller-supplied transport (stream-ordered on `stream`; byte
 * offsets into the exchange workspace `ws`, which the caller registered with its transport):
 *   temporal transformer (model/adapter_spatial_temporal.py:280) all_to_all: frame shards <-> pixel shards around the block
 *                       (everything inside it is per pixel), or -- when the transport has no all_to_all or the
 *                       pixels do not divide by the ranks -- all_gather of the K|V rows over the frame axis
 *   Conv3d (3,1,1)      (TemporalResnetBlock, :226)              halo_exchange of the first / last local frame
 *   temporal GroupNorm  (TemporalResnetBlock, :226)              all_reduce_sum_f32 of the (clip, group) sums
 * The transports used are torch.distributed over RCCL (ctrl-adapter_amd/clip_parallel.py); any transport with these
 * semantics works.  Callbacks return 0 on success.  If `ws_bytes` is too small the call fails with return code 2 and
 * `ws_needed` holds the size to retry with.  Runs on the caller's stream only (no stream lanes).
 * Results: the all_gather form is BIT-identical to the unsharded forward; the all_to_all form is tolerance-equal (<= 3.4e-4
 * rel-inf observed, 1e-3 asserted) -- its way back carries the temporal branch as fp16 and the AlphaBlender runs after it,
 * where the unsharded path blends in the fp32 epilogue of the block's last GEMM.  When the pixels of a frame do not divide by
 * the ranks the all_to_all form is dropped for that block (logged once on stderr). */
typedef struct ctrl_clip_comm {
    int32_t rank, world;
    void* ws; int64_t ws_bytes;
    /* recv[r*bytes .. (r+1)*bytes) = rank r's send[0 .. bytes) */
    int (*all_gather)(void* user, int64_t send_off, int64_t recv_off, int64_t bytes_per_rank, void* stream);
    /* count floats at off, summed over ranks in place */
    int (*all_reduce_sum_f32)(void* user, int64_t off, int64_t count, void* stream);
    /* send_prev -> rank-1's recv_next, send_next -> rank+1's recv_prev (edge ranks have no such neighbour: nothing is
       sent and the corresponding recv area is left untouched) */
    int (*halo_exchange)(void* user, int64_t send_prev_off, int64_t send_next_off, int64_t recv_prev_off, int64_t recv_next_off,
                         int64_t bytes, void* stream);
    void* user;
    int64_t ws_needed;
    /* optional (NULL: the K|V all_gather form is used): recv[r*bytes .. (r+1)*bytes) = rank r's send[me*bytes .. (me+1)*bytes),
       i.e. block r of my send area goes to rank r.  Swaps frame shards for pixel shards around the temporal transformer:
       2 + 1 KB per token and direction instead of the 2*C*2 B x world a rank RECEIVES per token in the all_gather form
        */
    int (*all_to_all)(void* user, int64_t send_off, int64_t recv_off, int64_t bytes_per_rank, void* stream);
    /* optional: further transports (own communicator, own workspace `ws`) for the adapter's stream lanes.  The blocks of the four
       pyramid levels are independent, and on whole clips they run on four HIP streams; with ONE transport the sharded forward stays
       on the caller's stream (the exchanges of one communicator must be issued and executed in one order on every rank), which
       alone costs ~17 % at world 1.  A chain of L transports lets lane l issue its exchanges on transport l: every rank walks the
       same program, so every communicator still sees its calls in one order, and the all-to-all of a 128^2 block runs under the
       other lanes' compute.  rank / world must agree along the chain; ws_needed is reported on the head for all of them. */
    struct ctrl_clip_comm* next_lane;
} ctrl_clip_comm;
int ctrl_adapter_forward_clip_sharded(ctrl_adapter* h,
                                      const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                                      const float* timesteps, int t_count,
                                      const void* encoder_hidden_states, int ehs_dtype, int ehs_batch, int Lk,
                                      void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                                      ctrl_clip_comm* comm, void* stream);

/* ---- The production transport, native: RCCL over xGMI, enqueued on the forward's stream from C++ (csrc/clip_rccl.cpp).  No host
 * callback into Python sits between launches, so ctrl_adapter_forward_clip_sharded is hipGraph-capturable with it (under capture the
 * forward uses the head of a next_lane chain only: RCCL refuses several communicators on forked streams inside one capture).  RCCL is
 * resolved at run time (dlopen; the copy already loaded into the process wins).  Rank 0 of the clip's group draws a unique id, the
 * caller ships its 128 bytes to the other ranks (any channel), every rank creates the communicator (collective), binds its
 * exchange workspace and passes the filled ctrl_clip_comm to the sharded forward. */
typedef struct ctrl_rccl_comm ctrl_rccl_comm;
int ctrl_rccl_unique_id(void* out128);
int ctrl_rccl_comm_create(const void* id128, int rank, int world, ctrl_rccl_comm** out);
void ctrl_rccl_comm_destroy(ctrl_rccl_comm* c);
int ctrl_rccl_comm_bind(ctrl_rccl_comm* c, void* ws, int64_t ws_bytes, int use_all_to_all, ctrl_clip_comm* cs);
int64_t ctrl_rccl_comm_bytes_sent(const ctrl_rccl_comm* c);

/* ---- Fused step: ctrl_controlnet_forward + ctrl_adapter_forward[_scatter] of one denoise step as one call (the two
 * back-to-back calls of the pipelines, sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:1323,1338 and
 * svd/...:684,709, i2vgen_xl/...:957,1042).  Same arguments as the two calls (the ControlNet outputs cn_outs[13] are
 * the adapter inputs; use_mid = pass cn_outs[12] as mid_block_res_sample; frame_pos may be NULL = no scatter); same
 * results bit for bit; the ControlNet runs on a plan-owned stream and every adapter block starts as soon as ITS input
 * is ready.  CTRL_STEP_OVERLAP=0 runs the two halves back to back. */
int ctrl_step_forward(ctrl_controlnet* cn, ctrl_adapter* ad,
                      const void* sample, int sample_dtype, int N, int Hs, int Ws,
                      const float* cn_timesteps, int cn_t_count,
                      const void* cn_encoder_hidden_states, int cn_ehs_dtype, int cn_Lk,
                      const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                      void* const* cn_outs, int cn_out_dtype,
                      int num_frames, const float* ad_timesteps, int ad_t_count,
                      const void* ad_encoder_hidden_states, int ad_ehs_dtype, int ad_ehs_batch, int ad_Lk,
                      int use_mid, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out, void* stream);

/* ---- Router (model/ctrl_router.py) ---- */
/* weights_out fp32 device [num_routers + (has_mid?1:0)][E]; wg fp32 device same shape (Linear(1,E).weight[:,0]
 * per router; ignored for equal_weights); mask host int[E] (NULL = all ones). */
int ctrl_router_weights(const float* wg, const int* mask_host, float* weights_out, int R, int E,
                        int equal_weights, void* stream);
/* out = sum_k weights[row*E + widx[k]] * experts[k]  over K active experts, element-wise on n elements */
int ctrl_router_merge(const void* const* experts_host, const float* weights_row, const int* widx_host, int K,
                      void* out, int dtype, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTRL_HIP_H */
