// Block builders (module tree -> parameter names, shared by the spec and the packer) and block runners
// (sequence of HIP kernel launches) used by both orchestrators.  Names of leaf modules follow
// diffusers v0.27.x (SURVEY.md section 8b "Weights / persistence").
#include "plan_common.h"

// ------------------------------------------------------------------------------------------ builders
int build_resnet(ParamSink& ps, const std::string& pre, int Cin, int Cout, bool force_shortcut, ResnetW* w, bool dup, int dup_shortcut) {
    w->Cin = Cin; w->Cout = Cout;
    TRY(ps.norm(pre + ".norm1", Cin, &w->norm1));
    TRY(ps.conv(pre + ".conv1", Cout, Cin, 3, false, &w->conv1, dup));
    // time_emb_proj is registered by the caller (batched across blocks) to keep state-dict order irrelevant
    TRY(ps.norm(pre + ".norm2", Cout, &w->norm2));
    TRY(ps.conv(pre + ".conv2", Cout, Cout, 3, false, &w->conv2, dup));
    w->has_shortcut = force_shortcut || (Cin != Cout);
    if (w->has_shortcut) TRY(ps.conv(pre + ".conv_shortcut", Cout, Cin, 1, false, &w->shortcut, dup_shortcut < 0 ? dup : dup_shortcut != 0));
    return 0;
}

int build_attn_self(ParamSink& ps, const std::string& pre, int dim, int heads, int D, AttnW* w) {
    w->heads = heads; w->D = D; w->inner = heads * D;
    TRY(ps.linear_cat({pre + ".to_q", pre + ".to_k", pre + ".to_v"}, {w->inner, w->inner, w->inner}, dim, false, &w->qkv));
    TRY(ps.linear(pre + ".to_out.0", dim, w->inner, true, false, &w->out));
    return 0;
}

int build_attn_cross(ParamSink& ps, const std::string& pre, int dim, int cross, int heads, int D, AttnW* w) {
    w->heads = heads; w->D = D; w->inner = heads * D;
    TRY(ps.linear(pre + ".to_q", w->inner, dim, false, false, &w->q));
    TRY(ps.linear_cat({pre + ".to_k", pre + ".to_v"}, {w->inner, w->inner}, cross, false, &w->kv));
    // the V rows of `kv` double as the stand-alone to_v operand of the single-key (Lk == 1) path
    w->v.w = w->kv.w ? w->kv.w + (size_t)w->inner * cross : nullptr;
    w->v.b = nullptr; w->v.N = w->inner; w->v.K = cross;
    TRY(ps.linear(pre + ".to_out.0", dim, w->inner, true, false, &w->out));
    return 0;
}

int build_basic_tb(ParamSink& ps, const std::string& pre, int dim, int heads, int D, int cross, BasicTBW* w) {
    w->dim = dim; w->cross = cross;
    TRY(ps.norm(pre + ".norm1", dim, &w->norm1));
    TRY(build_attn_self(ps, pre + ".attn1", dim, heads, D, &w->attn1));
    TRY(ps.norm(pre + ".norm2", dim, &w->norm2));
    TRY(build_attn_cross(ps, pre + ".attn2", dim, cross, heads, D, &w->attn2));
    TRY(ps.norm(pre + ".norm3", dim, &w->norm3));
    TRY(ps.linear(pre + ".ff.net.0.proj", 8 * dim, dim, true, true, &w->ff1));
    TRY(ps.linear(pre + ".ff.net.2", dim, 4 * dim, true, false, &w->ff2));
    TRY(ps.ffn_perm(&w->ff2));
    return 0;
}

int build_temporal_tb(ParamSink& ps, const std::string& pre, int dim, int heads, int D, int cross, TemporalTBW* w) {
    w->dim = dim; w->cross = cross;
    TRY(ps.norm(pre + ".norm_in", dim, &w->norm_in));
    TRY(ps.linear(pre + ".ff_in.net.0.proj", 8 * dim, dim, true, true, &w->ffin1));
    TRY(ps.linear(pre + ".ff_in.net.2", dim, 4 * dim, true, false, &w->ffin2));
    TRY(ps.ffn_perm(&w->ffin2));
    TRY(ps.norm(pre + ".norm1", dim, &w->norm1));
    TRY(build_attn_self(ps, pre + ".attn1", dim, heads, D, &w->attn1));
    TRY(ps.norm(pre + ".norm2", dim, &w->norm2));
    TRY(build_attn_cross(ps, pre + ".attn2", dim, cross, heads, D, &w->attn2));
    TRY(ps.norm(pre + ".norm3", dim, &w->norm3));
    TRY(ps.linear(pre + ".ff.net.0.proj", 8 * dim, dim, true, true, &w->ff1));
    TRY(ps.linear(pre + ".ff.net.2", dim, 4 * dim, true, false, &w->ff2));
    TRY(ps.ffn_perm(&w->ff2));
    return 0;
}

// ------------------------------------------------------------------------------------------ runners
int run_groupnorm(Ctx& cx, const Norm& n, const TV& x, half_t* y, int imgs, int rows, float eps, bool silu, bool split) {
    if (op_gn_fused_applies(x.dt, rows, n.C, 32)) {      // small map: statistics + apply in one launch
        RUN(cx, op_gn_fused(x.p, x.dt, n.g, n.b, y, imgs, rows, n.C, 32, eps, silu ? 1 : 0, cx.s, split ? 2 * n.C : n.C, split ? n.C : 0));
        return 0;
    }
    float* st = cx.stats(op_gn_stats_floats(imgs, rows, n.C, 32));
    RUN(cx, op_gn_stats(x.p, x.dt, st, imgs, rows, n.C, 32, cx.s));
    RUN(cx, op_gn_apply(x.p, x.dt, st, n.g, n.b, y, imgs, rows, n.C, 32, eps, silu ? 1 : 0, cx.s, split ? 2 * n.C : n.C, split ? n.C : 0));
    return 0;
}

int run_layernorm(Ctx& cx, const Norm& n, const TV& x, half_t* y, int M, int dim) {
    RUN(cx, op_layernorm(x.p, x.dt, dim, n.g, n.b, y, dim, M, dim, 1e-5f, cx.s));
    return 0;
}

int run_conv(Ctx& cx, const ConvW& c, const half_t* x, const TV& y, int N, int Hin, int Win, const ConvOpts& o) {
    const int k = c.taps == 9 ? 3 : 1, pad = c.taps == 9 ? 1 : 0;
    const int Hout = (Hin * o.up + 2 * pad - k) / o.stride + 1, Wout = (Win * o.up + 2 * pad - k) / o.stride + 1;
    IGemmArgs g = {};
    // a 1x1 convolution at stride 1 is a row GEMM over the pixels: the rows form of the kernels skips the per-k-tile gather arithmetic
    // of the conv2d form (M32768 N320 K640 split operand: 40 us against 62 us in the round-5 profile); same arithmetic, same results
    const bool as_rows = c.taps == 1 && o.stride == 1 && o.up == 1 && o.res_up != 2;
    g.A = x; g.lda = o.lda ? o.lda : c.Cin; g.mode = as_rows ? IG_ROWS : IG_CONV2D; g.Cin = c.Cin; g.taps = c.taps;
    g.Hin = Hin; g.Win = Win; g.Hout = Hout; g.Wout = Wout; g.stride = o.stride; g.up = o.up;
    g.W = c.w; g.M = N * Hout * Wout; g.Nout = c.Cout; g.Ktot = c.taps * c.Cin;
    g.bias = c.b; g.rowvec = o.rowvec; g.rowvec_ld = o.rowvec_ld; g.rows_per_img = Hout * Wout;
    g.scale = 1.f; g.act = o.act; g.a_split = c.paired ? 2 : (c.dup ? 1 : 0);
    set_res(g, o.res, c.Cout);
    g.res_up = o.res_up;
    set_out(g, y, c.Cout, c.Cout);
    const size_t mk = cx.mark();
    const int sk = igemm_splitk_factor(g);
    if (sk > 1) {
        g.splitk_ws_bytes = (int64_t)igemm_splitk_ws_bytes(g, sk);
        g.splitk_ws = cx.alloc((size_t)g.splitk_ws_bytes);
        // ticket words of the in-launch reduction: a slice of the pool the forward zeroes ONCE (with the GroupNorm tickets) instead
        // of a fill launch in front of every split-K convolution
        const size_t tw = igemm_splitk_ticket_words(g, sk);
        if (tw) g.splitk_tickets = (int32_t*)cx.stats(tw);
    }
    RUN(cx, op_igemm(g, cx.s));
    cx.release(mk);
    return 0;
}

int run_linear(Ctx& cx, const Lin& l, const half_t* x, long ldx, const TV& y, long ldy, int M, const TV& res, long ldres,
               const float* rowvec, int rowvec_ld, int rows_per_vec, const float* blend_mix, const TV& blend_other,
               const Norm* ln, half_t* ln_out) {
    IGemmArgs g = {};
    g.A = x; g.lda = ldx; g.mode = IG_ROWS; g.Cin = l.K; g.taps = 1;
    g.W = l.w; g.M = M; g.Nout = l.N; g.Ktot = l.K;
    g.bias = l.b; g.scale = 1.f; g.geglu = l.geglu ? 1 : 0;
    g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rows_per_img = rows_per_vec > 0 ? rows_per_vec : 1;
    set_res(g, res, ldres);
    set_out(g, y, ldy, l.geglu ? l.N / 2 : l.N);
    set_blend(g, blend_mix, blend_other, ldy);
    RUN(cx, op_igemm(g, cx.s));
    if (ln && ln_out) TRY(run_layernorm(cx, *ln, y, ln_out, M, l.N));
    return 0;
}

int run_resnet(Ctx& cx, const ResnetW& w, const TV& x, const TV& out, int N, int H, int W, int up,
               const float* temb_proj, int temb_ld, float eps) {
    const int Ho = H * up, Wo = W * up;
    const size_t m = cx.mark();
    half_t* a = cx.h((size_t)N * H * W * w.conv1.Cin);          // 2 x Cin wide for a split-operand conv1
    TRY(run_groupnorm(cx, w.norm1, x, a, N, H * W, eps, true, w.conv1.dup));
    // conv1 output feeds only GroupNorm: keep it in the stream dtype (its statistics are taken from this copy)
    TV h1 = cx.h1_f16 ? tv16(cx.h((size_t)N * Ho * Wo * w.Cout)) : stream_alloc(cx, (size_t)N * Ho * Wo * w.Cout, false);
    ConvOpts o1; o1.up = up; o1.rowvec = temb_proj; o1.rowvec_ld = temb_ld;
    TRY(run_conv(cx, w.conv1, a, h1, N, H, W, o1));
    half_t* b = cx.h((size_t)N * Ho * Wo * w.conv2.Cin);
    TRY(run_groupnorm(cx, w.norm2, h1, b, N, Ho * Wo, eps, true, w.conv2.dup));
    TV sc = x;
    int sc_up = 0;
    if (w.has_shortcut) {
        CTRL_CHECK(cx.dry || x.m16 != nullptr, "resnet: the shortcut conv needs an fp16 copy of its input");
        CTRL_CHECK(x.lo_off > 0 || !w.shortcut.dup, "resnet: a split-operand shortcut conv needs a split [hi | lo] input mirror");
        // a 1x1 conv commutes with the nearest up-sampling in front of it: with up = 2 the shortcut runs on the INPUT grid (a
        // quarter of the rows) and conv2's epilogue reads it through the up-sampling (ctrl_igemm_desc::res_up) -- bit-identical
        const bool low = (up == 2) && (Ho % 2 == 0) && (Wo % 2 == 0);
        TV s2 = stream_alloc(cx, (size_t)N * (low ? H * W : Ho * Wo) * w.Cout, false);
        ConvOpts os; os.up = low ? 1 : up;
        if (x.lo_off > 0 && !w.shortcut.dup) os.lda = 2 * w.Cin;      // plain conv on a split mirror: the hi half of every row
        TRY(run_conv(cx, w.shortcut, x.m16, s2, N, H, W, os));
        sc = s2;
        sc_up = low ? 2 : 0;
    }
    ConvOpts o2; o2.res = sc; o2.res_up = sc_up;
    TRY(run_conv(cx, w.conv2, b, out, N, Ho, Wo, o2));
    cx.release(m);
    return 0;
}

// softmax scale in the exp2 domain: the K projections leave their GEMM epilogue multiplied by it (one fp32 multiply before
// the rounding to fp16 that happens anyway), see ctrl_attn_desc::k_prescaled
static inline float attn_k_scale(int D) { return 1.4426950408889634f / sqrtf((float)D); }

static int run_attention(Ctx& cx, const half_t* Q, long ldq, const half_t* K, long ldk, const half_t* Vt, int Lkpad,
                         half_t* O, long ldo, int B, int kvB, int heads, int D, int Lq, int Lk) {
    AttnArgs a = {};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.Vt = Vt; a.Lkpad = Lkpad; a.kvB = kvB;
    a.O = O; a.ldo = ldo; a.B = B; a.heads = heads; a.D = D; a.Lq = Lq; a.Lk = Lk;
    a.k_prescaled = 1;
    a.scale = 1.0f / sqrtf((float)D);
    RUN(cx, op_flash_attn(a, cx.s));
    return 0;
}

// self attention on LN'd tokens; out = attn_out_proj(attn) + bias + resid
// `addvec` (optional): a per-image vector added to the block output by the out-projection's epilogue -- the
// single-key cross-attention that follows the self-attention (note N5) costs no pass of its own
static int run_self_attn(Ctx& cx, const AttnW& w, const half_t* xn, int dim, const TV& resid, const TV& out, int B, int L,
                         const float* addvec = nullptr, int addvec_rows = 0, const Norm* ln_next = nullptr, half_t* ln_out = nullptr) {
    const size_t mk = cx.mark();
    const int M = B * L, Ci = w.inner;
    const int Lpad = (L + 63) / 64 * 64;
    half_t* qk = cx.h((size_t)M * 2 * Ci);
    half_t* vt = cx.h((size_t)B * Ci * Lpad);
    if (L % 8) RUN(cx, op_fill_zero(vt, (size_t)B * Ci * Lpad * sizeof(half_t), cx.s));   // attention contract: finite pad columns
    // Q | K | V in ONE launch (round 5): Q|K leave row-major through the coalesced vector epilogue, V transposed ([batch][C][tokens],
    // the attention kernel's A operand) through the LDS-transposed one -- a tile lies in one segment because the dispatcher only
    // picks tile widths that divide 2 * Ci (ctrl_igemm_desc::seg), so the activation panel is read once instead of twice and the
    // grid has 1.5 x the tiles of the Q|K launch.  CTRL_QKV_ONE=0: the two launches of rounds 1-4 (bit-identical results).
    const bool qkv_one = !policy_is0(P_QKV_ONE);
    IGemmArgs g = {};
    g.A = xn; g.lda = dim; g.mode = IG_ROWS; g.Cin = dim; g.taps = 1;
    g.W = w.qkv.w; g.M = M; g.Nout = 2 * Ci; g.Ktot = dim; g.scale = 1.f;
    g.scale2_from = Ci; g.scale2_to = 2 * Ci; g.scale2 = attn_k_scale(w.D);      // the K third leaves pre-scaled for the attention kernel
    g.nseg = 1;
    g.seg[0] = IGemmSeg{qk, 2 * Ci, 0, 2 * Ci, SEG_ROW, DT_F16, 1, 0};
    if (qkv_one && (2 * Ci) % 64 == 0 && L % 8 == 0 && Lpad % 8 == 0 && M % 8 == 0) {
        g.Nout = 3 * Ci;
        g.nseg = 2;
        g.seg[1] = IGemmSeg{vt, Lpad, 2 * Ci, Ci, SEG_TRANSPOSED, DT_F16, L, 0};
        RUN(cx, op_igemm(g, cx.s));
    } else {
        RUN(cx, op_igemm(g, cx.s));
        IGemmArgs gv = g;
        gv.scale2_from = 0; gv.scale2_to = 0; gv.scale2 = 0.f;
        gv.W = w.qkv.w + (size_t)2 * Ci * dim; gv.Nout = Ci;
        gv.seg[0] = IGemmSeg{vt, Lpad, 0, Ci, SEG_TRANSPOSED, DT_F16, L, 0};
        RUN(cx, op_igemm(gv, cx.s));
    }
    half_t* o = cx.h((size_t)M * Ci);
    TRY(run_attention(cx, qk, 2 * Ci, qk + Ci, 2 * Ci, vt, Lpad, o, Ci, B, B, w.heads, w.D, L, L));
    TRY(run_linear(cx, w.out, o, Ci, out, dim, M, resid, dim, addvec, dim, addvec_rows, nullptr, TV(), ln_next, ln_out));
    cx.release(mk);
    return 0;
}

// out-projection of the single value vector of a one-key cross-attention: softmax over one key == 1, so the attention
// output is to_out(to_v(ctx)) for every query of the image (SURVEY.md note N5): [e.batch][dim] fp32
int single_key_vector(Ctx& cx, const AttnW& w, int dim, const EhsCtx& e, float** out) {
    float* v = cx.f((size_t)e.batch * w.inner);
    RUN(cx, op_linear_small(e.f32, e.cross, w.v.w, nullptr, v, w.inner, e.batch, w.inner, e.cross, 0, 0, cx.s));
    float* o = cx.f((size_t)e.batch * dim);
    RUN(cx, op_linear_small(v, w.inner, w.out.w, w.out.b, o, dim, e.batch, dim, w.inner, 0, 0, cx.s));
    *out = o;
    return 0;
}

// cross attention; Lk == 1 is the degenerate query-independent case (SURVEY.md note N5)
// K and V^T of the text states (step-invariant: optionally kept in / taken from the plan's cache)
int project_text_kv(Ctx& cx, const AttnW& w, const EhsCtx& e, PreKV* out) {
    const int Ci = w.inner;
    const int Lkpad = (e.Lk + 63) / 64 * 64;
    const int Mk = e.batch * e.Lk;
    const bool cached = cx.kvc && cx.kvc->mode != KvCache::OFF;
    half_t* k = nullptr; half_t* vt = nullptr;
    if (cached) {
        KvCache::Slot* sl = nullptr;
        TRY(cx.kvc->get((size_t)Mk * Ci, (size_t)e.batch * Ci * Lkpad, cx.dry, &sl, cx.capturing));
        k = sl->k; vt = sl->vt;
    } else {
        k = cx.h((size_t)Mk * Ci);
        vt = cx.h((size_t)e.batch * Ci * Lkpad);
    }
    if (!(cached && cx.kvc->mode == KvCache::REUSE)) {
        if (e.Lk % 8) RUN(cx, op_fill_zero(vt, (size_t)e.batch * Ci * Lkpad * sizeof(half_t), cx.s));   // finite pad columns
        // K | V^T of the text states in ONE launch (round 5; 77 tokens per prompt are not a multiple of 8, so the mixed segment list
        // takes the scalar epilogue -- as the V launch always did; these GEMMs have 616 rows): K pre-scaled, V plain
        const bool kv_one = !policy_is0(P_QKV_ONE);
        IGemmArgs g = {};
        g.A = e.h16; g.lda = e.cross; g.mode = IG_ROWS; g.Cin = e.cross; g.taps = 1;
        g.W = w.kv.w; g.M = Mk; g.Nout = Ci; g.Ktot = e.cross; g.scale = attn_k_scale(w.D);     // pre-scaled K
        g.nseg = 1;
        g.seg[0] = IGemmSeg{k, Ci, 0, Ci, SEG_ROW, DT_F16, 1, 0};
        if (kv_one && Ci % 64 == 0) {
            g.Nout = 2 * Ci;
            g.scale2_from = Ci; g.scale2_to = 0; g.scale2 = 1.f;
            g.nseg = 2;
            g.seg[1] = IGemmSeg{vt, Lkpad, Ci, Ci, SEG_TRANSPOSED, DT_F16, e.Lk, 0};
            RUN(cx, op_igemm(g, cx.s));
        } else {
            RUN(cx, op_igemm(g, cx.s));
            IGemmArgs gv = g;
            gv.scale = 1.f;
            gv.W = w.kv.w + (size_t)Ci * e.cross;
            gv.seg[0] = IGemmSeg{vt, Lkpad, 0, Ci, SEG_TRANSPOSED, DT_F16, e.Lk, 0};
            RUN(cx, op_igemm(gv, cx.s));
        }
    }
    out->k = k; out->vt = vt;
    return 0;
}

// xn_pre (optional): LayerNorm(ln)(x) already computed by x's producer; ln_next / ln_out: the LayerNorm applied to `out` next
static int run_cross_attn(Ctx& cx, const AttnW& w, const Norm& ln, const TV& x, int dim, const TV& out, int B, int L,
                          const EhsCtx& e, const PreKV* pre = nullptr, const half_t* xn_pre = nullptr,
                          const Norm* ln_next = nullptr, half_t* ln_out = nullptr) {
    const size_t mk = cx.mark();
    const int M = B * L, Ci = w.inner;
    if (e.Lk == 1) {
        // softmax over one key == 1  =>  out = to_out(to_v(ctx)) for every query of the image
        float* o = nullptr;
        TRY(single_key_vector(cx, w, dim, e, &o));
        RUN(cx, op_add_rowvec(x.p, x.dt, o, dim, out.p, out.dt, (size_t)M, dim, L, e.batch, cx.s));
        if (ln_next && ln_out) TRY(run_layernorm(cx, *ln_next, out, ln_out, M, dim));
        cx.release(mk);
        return 0;
    }
    const half_t* xn = xn_pre;
    if (!xn) {
        half_t* xb = cx.h((size_t)M * dim);
        TRY(run_layernorm(cx, ln, x, xb, M, dim));
        xn = xb;
    }
    half_t* q = cx.h((size_t)M * Ci);
    TRY(run_linear(cx, w.q, xn, dim, tv16(q), Ci, M, TV(), 0));
    const int Lkpad = (e.Lk + 63) / 64 * 64;
    PreKV kv;
    if (pre) kv = *pre;
    else TRY(project_text_kv(cx, w, e, &kv));
    half_t* k = kv.k; half_t* vt = kv.vt;
    half_t* o = cx.h((size_t)M * Ci);
    TRY(run_attention(cx, q, Ci, k, Ci, vt, Lkpad, o, Ci, B, e.batch == 1 ? 1 : B, w.heads, w.D, L, e.Lk));
    TRY(run_linear(cx, w.out, o, Ci, out, dim, M, x, dim, nullptr, 0, 0, nullptr, TV(), ln_next, ln_out));
    cx.release(mk);
    return 0;
}

// xn_pre (optional): LayerNorm(ln)(x) already computed by x's producer
static int run_ff(Ctx& cx, const Norm& ln, const Lin& ff1, const Lin& ff2, const TV& x, int dim, const TV& out, int M,
                  const half_t* xn_pre = nullptr) {
    const size_t mk = cx.mark();
    const half_t* xn = xn_pre;
    if (!xn) {
        half_t* xb = cx.h((size_t)M * dim);
        TRY(run_layernorm(cx, ln, x, xb, M, dim));
        xn = xb;
    }
    TRY(run_ffn(cx, ff1, ff2, xn, dim, out, M, x));
    cx.release(mk);
    return 0;
}

int run_ffn(Ctx& cx, const Lin& ff1, const Lin& ff2, const half_t* xn, int dim, const TV& out, int M, const TV& res,
            const float* blend_mix, const TV& blend_other, const Norm* ln, half_t* ln_out) {
    const int inner = ff1.N / 2;
    const bool fused = ff1.geglu && op_ffn_fused_shape_ok(dim, inner) && ff2.N == dim && ff2.K == inner && ff2.wperm &&
                       ff1.b && M >= kFfnFusedMinM && !policy_is0(P_FF_FUSED);
    if (fused) {
        FfnArgs f = {};
        f.X = xn; f.ldx = dim; f.W1 = ff1.w; f.b1 = ff1.b; f.W2p = ff2.wperm;
        IGemmArgs& g = f.out;
        g.mode = IG_ROWS; g.Cin = inner; g.taps = 1; g.M = M; g.Nout = dim; g.Ktot = inner;
        g.bias = ff2.b; g.scale = 1.f; g.rows_per_img = 1;
        set_res(g, res, dim);
        set_out(g, out, dim, dim);
        set_blend(g, blend_mix, blend_other, dim);
        RUN(cx, op_ffn_fused(f, cx.s));
        if (ln && ln_out) TRY(run_layernorm(cx, *ln, out, ln_out, M, dim));
        return 0;
    }
    const size_t mk = cx.mark();
    half_t* hmid = cx.h((size_t)M * inner);
    TRY(run_linear(cx, ff1, xn, dim, tv16(hmid), inner, M, TV(), 0));
    TRY(run_linear(cx, ff2, hmid, inner, out, ff2.N, M, res, dim, nullptr, 0, 0, blend_mix, blend_other, ln, ln_out));
    cx.release(mk);
    return 0;
}

int run_basic_tb(Ctx& cx, const BasicTBW& w, const TV& X, const TV& out, int B, int L, const EhsCtx& e, const float* ov_pre,
                 const PreKV* kv_pre, const half_t* X_ln) {
    CTRL_CHECK(e.batch == 1 || e.batch == B, "encoder_hidden_states batch must be 1 or equal to the sample batch");
    const size_t mk = cx.mark();
    const int M = B * L, dim = w.dim;
    const half_t* xn = X_ln;
    if (!xn) {
        half_t* xb = cx.h((size_t)M * dim);
        TRY(run_layernorm(cx, w.norm1, X, xb, M, dim));
        xn = xb;
    }
    // every LayerNorm of the block is handed to the GEMM that produces its input (run_linear runs the stand-alone kernel right behind
    // it; the fused-epilogue form of round 3 lost and was removed)
    half_t* xn2 = cx.h((size_t)M * dim);        // LayerNorm(norm2)(x1), then re-used for LayerNorm(norm3)(x2)
    TV x2 = stream_alloc(cx, (size_t)M * dim, false);
    if (e.Lk == 1) {
        // x2 = X + attn1(norm1 X) + to_out(to_v(ctx)): the query-independent cross-attention term rides on the self-
        // attention's out-projection epilogue (per-image vector; one vector for all rows when the context is broadcast)
        float* ov = const_cast<float*>(ov_pre);
        if (!ov) TRY(single_key_vector(cx, w.attn2, dim, e, &ov));
        TRY(run_self_attn(cx, w.attn1, xn, dim, X, x2, B, L, ov, e.batch == 1 ? M : L, &w.norm3, xn2));
        TRY(run_ff(cx, w.norm3, w.ff1, w.ff2, x2, dim, out, M, xn2));
    } else {
        TV x1 = stream_alloc(cx, (size_t)M * dim, false);
        half_t* xn3 = cx.h((size_t)M * dim);
        TRY(run_self_attn(cx, w.attn1, xn, dim, X, x1, B, L, nullptr, 0, &w.norm2, xn2));
        TRY(run_cross_attn(cx, w.attn2, w.norm2, x1, dim, x2, B, L, e, kv_pre, xn2, &w.norm3, xn3));
        TRY(run_ff(cx, w.norm3, w.ff1, w.ff2, x2, dim, out, M, xn3));
    }
    cx.release(mk);
    return 0;
}
