// Ctrl-Adapter forward orchestrator (C++ host code, enqueues the gfx950 kernels of libctrlhip).
//
// Restates the control flow of the reference:
//   ControlNetAdapter        model/ctrl_adapter.py:17-116 (which slots get a block :119-168), forward :171-224
//   AdapterSpatioTemporal    model/adapter_spatial_temporal.py:11-171 (module tree), forward :175-292:
//     timestep normalisation :190-198 -> resnet time embedding :206-209 -> spatial ResNet (+ x2 nearest
//     up-sampling of both branches, model/resnet_block_2d.py:174-184) :213-219 -> temporal ResNet + AlphaBlender
//     :223-231 -> GroupNorm / proj_in :252-257 -> frame-index embedding :259-266 -> spatial transformer :270-273 ->
//     temporal transformer + AlphaBlender :278-282 -> proj_out + residual :286-289
// Frames of a clip are kept frame-major ([(b f) h w][C], the reference's own (bf) c h w order), so the temporal
// ops address the frame axis with a stride instead of materialising the b c f h w permutes.
#include "plan_common.h"
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

namespace {

struct TResnetW {             // diffusers TemporalResnetBlock
    Norm norm1, norm2;
    ConvW conv1, conv2;
    Lin temb;                 // time_emb_proj
};

struct AdapterLayerW {
    ResnetW sres; Lin sres_temb;
    TResnetW tres;
    float* res_mix = nullptr;
    BasicTBW stb;
    TemporalTBW ttb;
    float* tr_mix = nullptr;
};

struct AdapterBlockW {
    int C = 0, up = 1, heads = 0;
    bool tok_f32 = false;     // outlier channels in this block's normalisation scales: its token stream stays fp32 (ParamSink::norm_scale_spread)
    Lin rte1, rte2;           // resnet_time_embedding
    Norm norm;
    Lin proj_in, proj_out;
    Lin tte1, tte2;           // transformer_time_embedding
    std::vector<AdapterLayerW> layers;
};

struct AdapterW {
    ctrl_adapter_config cfg;
    std::vector<int> slot_ids;          // which of the 12 ControlNet outputs get a block (ctrl_adapter.py:119-139)
    std::vector<AdapterBlockW> blocks;  // in slot order
    bool has_mid = false;
    AdapterBlockW mid;
};

const int INNER = 512;   // num_attention_heads(8) * attention_head_dim(64): adapter_spatial_temporal.py:62

int build_block(ParamSink& ps, const std::string& pre, int C, const ctrl_adapter_config& c, int up, AdapterBlockW* b) {
    b->C = C; b->up = up; b->heads = C / 64;      // :42 -- head count = C / attention_head_dim
    b->tok_f32 = ps.norm_scale_spread(pre + ".") > kNormSpreadGate;
    const bool sr = c.add_spatial_resnet, tr = c.add_temporal_resnet, st = c.add_spatial_transformer, tt = c.add_temporal_transformer;
    if (sr || tr) {
        TRY(ps.linear(pre + ".resnet_time_embedding.linear_1", C, C, true, false, &b->rte1));
        TRY(ps.linear(pre + ".resnet_time_embedding.linear_2", C, C, true, false, &b->rte2));
    }
    if (st || tt) {
        TRY(ps.norm(pre + ".norm", C, &b->norm));
        if (tt) {
            TRY(ps.linear(pre + ".transformer_time_embedding.linear_1", INNER, C, true, false, &b->tte1));
            TRY(ps.linear(pre + ".transformer_time_embedding.linear_2", INNER, INNER, true, false, &b->tte2));
        }
        TRY(ps.linear(pre + ".proj_in", INNER, C, true, false, &b->proj_in));
        TRY(ps.linear(pre + ".proj_out", C, INNER, true, false, &b->proj_out));
    }
    b->layers.resize(c.num_blocks);
    for (int i = 0; i < c.num_blocks; ++i) {
        AdapterLayerW& L = b->layers[i];
        const std::string si = std::to_string(i);
        if (sr) {
            TRY(build_resnet(ps, pre + ".spatial_resnets." + si, C, C, true, &L.sres));
            TRY(ps.linear(pre + ".spatial_resnets." + si + ".time_emb_proj", C, C, true, false, &L.sres_temb));
        }
        if (tr) {
            const std::string p = pre + ".temporal_resnets." + si;
            TRY(ps.norm(p + ".norm1", C, &L.tres.norm1));
            TRY(ps.conv(p + ".conv1", C, C, 3, true, &L.tres.conv1));
            TRY(ps.linear(p + ".time_emb_proj", C, C, true, false, &L.tres.temb));
            TRY(ps.norm(p + ".norm2", C, &L.tres.norm2));
            TRY(ps.conv(p + ".conv2", C, C, 3, true, &L.tres.conv2));
        }
        if (st) TRY(build_basic_tb(ps, pre + ".spatial_attentions." + si, INNER, b->heads, 64, c.cross_attention_dim, &L.stb));
        if (tt) TRY(build_temporal_tb(ps, pre + ".temporal_attentions." + si, INNER, b->heads, 64, c.cross_attention_dim, &L.ttb));
        if (sr && tr) TRY(ps.scalar(pre + ".resnets_time_mixer." + si + ".mix_factor", &L.res_mix));
        if (st && tt) TRY(ps.scalar(pre + ".transformers_time_mixer." + si + ".mix_factor", &L.tr_mix));
    }
    return 0;
}

int build_adapter(ParamSink& ps, const ctrl_adapter_config& c, AdapterW* w) {
    w->cfg = c;
    const int n = c.num_adapters_per_location;
    CTRL_CHECK(n >= 1 && n <= 3, "adapter: num_adapters_per_location must be 1..3");
    CTRL_CHECK(c.num_blocks >= 1 && c.num_blocks <= 8, "adapter: num_blocks out of range");
    CTRL_CHECK(c.cross_attention_dim % 64 == 0, "adapter: cross_attention_dim must be a multiple of 64");
    // model/ctrl_adapter.py:119-168
    static const int sel[4][4][3] = {{{0}, {2}, {0, 2}, {0, 1, 2}}, {{0}, {5}, {3, 5}, {3, 4, 5}},
                                     {{0}, {8}, {6, 8}, {6, 7, 8}}, {{0}, {11}, {9, 11}, {9, 10, 11}}};
    static const int chn[4][4][3] = {{{0}, {320}, {320, 320}, {320, 320, 320}}, {{0}, {640}, {320, 640}, {320, 640, 640}},
                                     {{0}, {1280}, {640, 1280}, {640, 1280, 1280}}, {{0}, {1280}, {1280, 1280}, {1280, 1280, 1280}}};
    const int on[4] = {c.loc_A, c.loc_B, c.loc_C, c.loc_D};
    std::vector<int> chans;
    for (int l = 0; l < 4; ++l)
        if (on[l])
            for (int k = 0; k < n; ++k) { w->slot_ids.push_back(sel[l][n][k]); chans.push_back(chn[l][n][k]); }
    const int up = c.backbone_sdxl ? 2 : 1;
    w->blocks.resize(chans.size());
    for (size_t i = 0; i < chans.size(); ++i)
        TRY(build_block(ps, "down_blocks_adapter." + std::to_string(i), chans[i], c, up, &w->blocks[i]));
    w->has_mid = c.loc_M != 0;
    if (w->has_mid) TRY(build_block(ps, "mid_block_adapter", 1280, c, up, &w->mid));
    return 0;
}

struct AFwd {
    int N, F, B;                 // frames total, frames per clip, clips
    const float* t; int t_count;
    EhsCtx e;                    // spatial cross-attention context
    EhsCtx e_first;              // temporal cross-attention context (first frame of each clip), Lk == 1 only
    int in_dt, out_dt;
    const int* out_map = nullptr;   // device: input frame -> output frame (ctrl_adapter_forward_scatter), or null
    // one clip's frames sharded over ranks: F / N / B above are LOCAL, Fg = frames of the whole clip
    ctrl_clip_comm* comm = nullptr;
    int Fg = 0;
    size_t* ws_peak = nullptr;      // exchange-workspace bytes the forward needs (recorded by every exchange)
};

// ---- exchanges of the frame-sharded clip: every one lays its buffers out from offset 0 of the exchange workspace; reuse
// is safe because everything (copies, transport calls, consumers) is ordered on the one stream the forward runs on ----
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline void ws_need(const AFwd& a, size_t bytes) { if (*a.ws_peak < bytes) *a.ws_peak = bytes; }
#define COMM_TRY(expr) do { if ((expr) != 0) CTRL_FAIL("clip-sharded exchange failed: " #expr); } while (0)

// clip-wide GroupNorm (statistics over all frames of a clip) on the local frames: local sums -> all-reduce over the
// ranks -> apply with the global element count.  y is the padded conv operand [clip][Fl + 2][HW][C] (frames 1 .. Fl).
int gn_clip_sharded(Ctx& cx, const Norm& n, const TV& x, half_t* y_padded, const AFwd& a, int HW, float eps) {
    const int C = n.C, B = a.B, Fl = a.F;
    float* st = cx.stats(op_gn_stats_floats(B, Fl * HW, C, 32));
    RUN(cx, op_gn_stats(x.p, x.dt, st, B, Fl * HW, C, 32, cx.s));
    const size_t nst = (size_t)B * 32 * 2;
    ws_need(a, nst * sizeof(float));
    if (!cx.dry) {
        HIP_TRY(hipMemcpyAsync(a.comm->ws, st, nst * sizeof(float), hipMemcpyDeviceToDevice, cx.s));
        COMM_TRY(a.comm->all_reduce_sum_f32(a.comm->user, 0, (int64_t)nst, cx.s));
    }
    RUN(cx, op_gn_apply(x.p, x.dt, (const float*)a.comm->ws, n.g, n.b, y_padded, B, Fl * HW, C, 32, eps, 1, cx.s,
                        C, 0, (long)a.Fg * HW, (long)(Fl + 2) * HW, HW));
    return 0;
}

// +-1-frame halo of the Conv3d operand: frame slots 0 / Fl + 1 of every clip <- the neighbour ranks' last / first frame
int halo_exchange_frames(Ctx& cx, half_t* np, const AFwd& a, int HW, int C) {
    const int B = a.B, Fl = a.F;
    const size_t frame_b = (size_t)HW * C * sizeof(half_t), pitch = (size_t)(Fl + 2) * frame_b;
    const size_t blk = al256((size_t)B * frame_b);
    ws_need(a, 4 * blk);
    if (cx.dry) return 0;
    char* ws = (char*)a.comm->ws;
    char* p = (char*)np;
    HIP_TRY(hipMemcpy2DAsync(ws + 0 * blk, frame_b, p + 1 * frame_b, pitch, frame_b, B, hipMemcpyDeviceToDevice, cx.s));
    HIP_TRY(hipMemcpy2DAsync(ws + 1 * blk, frame_b, p + (size_t)Fl * frame_b, pitch, frame_b, B, hipMemcpyDeviceToDevice, cx.s));
    COMM_TRY(a.comm->halo_exchange(a.comm->user, 0, (int64_t)blk, (int64_t)(2 * blk), (int64_t)(3 * blk), (int64_t)((size_t)B * frame_b), cx.s));
    if (a.comm->rank > 0) HIP_TRY(hipMemcpy2DAsync(p, pitch, ws + 2 * blk, frame_b, frame_b, B, hipMemcpyDeviceToDevice, cx.s));
    else HIP_TRY(hipMemset2DAsync(p, pitch, 0, frame_b, B, cx.s));                       // first frames of the clip: zero padding
    char* last = p + (size_t)(Fl + 1) * frame_b;
    if (a.comm->rank < a.comm->world - 1) HIP_TRY(hipMemcpy2DAsync(last, pitch, ws + 3 * blk, frame_b, frame_b, B, hipMemcpyDeviceToDevice, cx.s));
    else HIP_TRY(hipMemset2DAsync(last, pitch, 0, frame_b, B, cx.s));
    return 0;
}

// Everything of a block that depends on (timestep, encoder states, frame index) only -- the time / frame-index embedding
// MLPs, the per-layer time projections, the single-key cross-attention vectors -- is computed for ALL blocks at the start of
// the forward, one grouped launch per dependency level (instead of ~10 small-M launches inside every block)
struct BlockPre {
    float* temb = nullptr;                 // SiLU(resnet_time_embedding(t)) [N][C]
    float* femb = nullptr;                 // transformer_time_embedding(frame index) [F][512]
    std::vector<float*> sres_tp, tres_tp;  // per layer: time_emb_proj(SiLU(temb)) [N][C]
    std::vector<float*> stb_ov, ttb_ov;    // per layer: to_out(to_v(context)) of the one-key cross-attentions
};

int precompute_small(Ctx& cx, const ctrl_adapter_config& c, const std::vector<const AdapterBlockW*>& blocks, const AFwd& a,
                     std::vector<BlockPre>* pre) {
    const bool sr = c.add_spatial_resnet, tr = c.add_temporal_resnet, st = c.add_spatial_transformer, tt = c.add_temporal_transformer;
    const int N = a.N, Fe = a.comm ? a.Fg : a.F;
    pre->assign(blocks.size(), BlockPre());
    std::vector<SmallLin> L1, L2, L3;
    float* ts_by_c[3] = {nullptr, nullptr, nullptr};     // sinusoids shared by the blocks of one width (320 / 640 / 1280)
    float* fs_by_c[3] = {nullptr, nullptr, nullptr};
    auto cidx = [](int C) { return C == 320 ? 0 : (C == 640 ? 1 : 2); };
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        const AdapterBlockW& b = *blocks[bi];
        BlockPre& P = (*pre)[bi];
        const int C = b.C, ci = cidx(C);
        if (sr || tr) {
            if (!ts_by_c[ci]) {
                ts_by_c[ci] = cx.f((size_t)N * C);
                RUN(cx, op_timestep_sincos(a.t, a.t_count, ts_by_c[ci], N, C, cx.s));
            }
            float* t1 = cx.f((size_t)N * C);
            P.temb = cx.f((size_t)N * C);
            L1.push_back({ts_by_c[ci], C, b.rte1.w, b.rte1.b, t1, C, N, C, C, 0, 1});
            // (SiLU on the way OUT: the embedding's only consumers are the time_emb_proj linears of the block's resnets, each of which starts
            //  with SiLU(temb) -- model/resnet_block_2d.py:191-192, TemporalResnetBlock likewise; applied once here instead of once per output
            //  column of every consumer: the same fp32 values, bit-identical, and the level-3 launch of a video forward no longer spends its time in exp / rcp)
            L2.push_back({t1, C, b.rte2.w, b.rte2.b, P.temb, C, N, C, C, 0, 1});
        }
        if (tt) {
            if (!fs_by_c[ci]) {
                fs_by_c[ci] = cx.f((size_t)Fe * C);
                RUN(cx, op_frameidx_sincos(fs_by_c[ci], Fe, Fe, C, cx.s));
            }
            float* f1 = cx.f((size_t)Fe * INNER);
            P.femb = cx.f((size_t)Fe * INNER);
            L1.push_back({fs_by_c[ci], C, b.tte1.w, b.tte1.b, f1, INNER, Fe, INNER, C, 0, 1});
            L2.push_back({f1, INNER, b.tte2.w, b.tte2.b, P.femb, INNER, Fe, INNER, INNER, 0, 0});
        }
        for (const AdapterLayerW& Lw : b.layers) {
            if (sr) {
                float* tp = cx.f((size_t)N * C);
                P.sres_tp.push_back(tp);
                L3.push_back({P.temb, C, Lw.sres_temb.w, Lw.sres_temb.b, tp, C, N, C, C, 0, 0});
            }
            if (tr) {
                float* tp = cx.f((size_t)N * C);
                P.tres_tp.push_back(tp);
                L3.push_back({P.temb, C, Lw.tres.temb.w, Lw.tres.temb.b, tp, C, N, C, C, 0, 0});
            }
            auto one_key = [&](const AttnW& w, const EhsCtx& e, std::vector<float*>* dst) {
                float* v = cx.f((size_t)e.batch * w.inner);
                float* o = cx.f((size_t)e.batch * INNER);
                L1.push_back({e.f32, e.row_ld ? e.row_ld : (long)e.cross, w.v.w, nullptr, v, w.inner, e.batch, w.inner, e.cross, 0, 0});
                L2.push_back({v, w.inner, w.out.w, w.out.b, o, INNER, e.batch, INNER, w.inner, 0, 0});
                dst->push_back(o);
            };
            if (st && a.e.Lk == 1) one_key(Lw.stb.attn2, a.e, &P.stb_ov);
            if (tt) one_key(Lw.ttb.attn2, a.e_first, &P.ttb_ov);
        }
    }
    if (!L1.empty()) RUN(cx, op_linear_small_group(L1.data(), (int)L1.size(), cx.s));
    if (!L2.empty()) RUN(cx, op_linear_small_group(L2.data(), (int)L2.size(), cx.s));
    if (!L3.empty()) RUN(cx, op_linear_small_group(L3.data(), (int)L3.size(), cx.s));
    return 0;
}

// temporal ResNet on frame-major rows [(b f) hw][C]
// blend_mix (optional): AlphaBlender fold -- out = a*x + (1-a)*TemporalResnet(x), a = sigmoid(*blend_mix); the spatial
// branch of the blender IS this block's input (adapter_spatial_temporal.py:226-229), so the last conv's epilogue does it
int run_temporal_resnet(Ctx& cx, const TResnetW& w, const TV& x, const TV& out, const AFwd& a, int HW, int C,
                        const float* tp /*[N][C]: time_emb_proj(SiLU(temb)), precompute_small*/, const float* blend_mix) {
    const size_t mk = cx.mark();
    const int N = a.N, M = N * HW;
    // frame-sharded clip: the conv operand is padded by one halo frame slot on each side of every clip's local frames
    half_t* n1 = cx.h(a.comm ? (size_t)a.B * (a.F + 2) * HW * C : (size_t)M * C);
    if (a.comm) {
        TRY(gn_clip_sharded(cx, w.norm1, x, n1, a, HW, 1e-6f));
        TRY(halo_exchange_frames(cx, n1, a, HW, C));
    } else {
        TRY(run_groupnorm(cx, w.norm1, x, n1, a.B, a.F * HW, 1e-6f, true));     // statistics span the clip's frames
    }
    TV h1 = stream_alloc(cx, (size_t)M * C, false);
    IGemmArgs g = {};
    g.A = n1; g.lda = C; g.mode = IG_TEMPORAL; g.Cin = C; g.taps = 3; g.F = a.F; g.HW = HW; g.t_pad = a.comm ? 1 : 0;
    g.W = w.conv1.w; g.M = M; g.Nout = C; g.Ktot = 3 * C; g.bias = w.conv1.b;
    g.rowvec = tp; g.rowvec_ld = C; g.rows_per_img = HW; g.scale = 1.f;
    set_out(g, h1, C, C);
    RUN(cx, op_igemm(g, cx.s));
    half_t* n2 = n1;
    if (a.comm) {
        TRY(gn_clip_sharded(cx, w.norm2, h1, n2, a, HW, 1e-6f));
        TRY(halo_exchange_frames(cx, n2, a, HW, C));
    } else {
        TRY(run_groupnorm(cx, w.norm2, h1, n2, a.B, a.F * HW, 1e-6f, true));
    }
    IGemmArgs g2 = g;
    g2.A = n2; g2.W = w.conv2.w; g2.bias = w.conv2.b; g2.rowvec = nullptr;
    set_res(g2, x, C);
    set_out(g2, out, C, C);
    set_blend(g2, blend_mix, x, C);
    RUN(cx, op_igemm(g2, cx.s));
    cx.release(mk);
    return 0;
}

// TemporalBasicTransformerBlock on frame-major tokens X [(b f) L][512]; every op but the attention is per token
// blend_mix / blend_other (optional): out = a*blend_other + (1-a)*block(X), folded into the last GEMM's epilogue
// X_ln (optional): norm_in(X) already computed by X's producer (the fused frame-embedding add + LayerNorm pass); the buffer is
// then re-used for the block's other LayerNorm outputs
int run_temporal_tb(Ctx& cx, const TemporalTBW& w, const TV& X, const TV& out, const AFwd& a, int L,
                    const float* blend_mix, const TV& blend_other, const float* ov /*one-key cross-attention vector*/,
                    half_t* X_ln = nullptr) {
    const size_t mk = cx.mark();
    const int M = a.N * L, dim = w.dim, Ci = w.attn1.inner;
    // x = ff_in(norm_in(x)) + x
    half_t* xn = X_ln;
    if (!xn) {
        xn = cx.h((size_t)M * dim);
        TRY(run_layernorm(cx, w.norm_in, X, xn, M, dim));
    }
    TV x0 = stream_alloc(cx, (size_t)M * dim, false);
    // (every LayerNorm below rides on the epilogue of the GEMM that produces its input: xn is free again once ff_in has
    // read it, and this stream is in order)
    TRY(run_ffn(cx, w.ffin1, w.ffin2, xn, dim, x0, M, X, nullptr, TV(), &w.norm1, xn));
    // x = attn1(norm1(x)) + x  (sequence = frames)
    half_t* o = nullptr;
    if (!a.comm) {
        half_t* qkv = cx.h((size_t)M * 3 * Ci);
        TRY(run_linear(cx, w.attn1.qkv, xn, dim, tv16(qkv), 3 * Ci, M, TV(), 0));
        o = cx.h((size_t)M * Ci);
        TAttnArgs ta = {};
        ta.Q = qkv; ta.ld = 3 * Ci; ta.O = o; ta.ldo = Ci; ta.Bc = a.B; ta.F = a.F; ta.HW = L; ta.heads = w.attn1.heads;
        ta.scale = 0.125f;
        RUN(cx, op_temporal_attn(ta, cx.s));
    } else {
        // frames sharded over ranks: Q stays local, the K|V rows of the local frames go straight from the QKV GEMM's
        // epilogue into the exchange workspace and are all-gathered over the frame axis (one exchange per block)
        half_t* q = cx.h((size_t)M * Ci);
        const size_t per_rank = al256((size_t)M * 2 * Ci * sizeof(half_t));
        ws_need(a, per_rank * (1 + (size_t)a.comm->world));
        char* ws = cx.dry ? (char*)(uintptr_t)0x1000 : (char*)a.comm->ws;
        IGemmArgs g = {};
        g.A = xn; g.lda = dim; g.mode = IG_ROWS; g.Cin = dim; g.taps = 1;
        g.W = w.attn1.qkv.w; g.M = M; g.Nout = 3 * Ci; g.Ktot = dim; g.scale = 1.f;
        g.nseg = 2;
        g.seg[0] = IGemmSeg{q, Ci, 0, Ci, SEG_ROW, DT_F16, 1, 0};
        g.seg[1] = IGemmSeg{ws, 2 * Ci, Ci, 2 * Ci, SEG_ROW, DT_F16, 1, 0};
        RUN(cx, op_igemm(g, cx.s));
        if (!cx.dry) COMM_TRY(a.comm->all_gather(a.comm->user, 0, (int64_t)per_rank, (int64_t)per_rank, cx.s));
        o = cx.h((size_t)M * Ci);
        TAttnArgs ta = {};
        ta.Q = q; ta.ld = Ci; ta.O = o; ta.ldo = Ci; ta.Bc = a.B; ta.F = a.Fg; ta.Fq = a.F; ta.Fl = a.F; ta.HW = L;
        ta.heads = w.attn1.heads; ta.scale = 0.125f;
        ta.KV = ws + per_rank; ta.ldkv = 2 * Ci;
        RUN(cx, op_temporal_attn(ta, cx.s));
    }
    // x = attn2(norm2(x), first-frame context) + attn1(...) + x : one key => the cross-attention term is one vector
    // (note N5), added by the out-projection's epilogue.  Rows are (b f p) and the context is the broadcast vector or
    // the first frame of the only clip: the same vector for every row.
    TV x2 = stream_alloc(cx, (size_t)M * dim, false);
    if (a.e_first.batch > 1) {
        // one context per clip: the vector of row (b f p) is the one of clip (b*L + p) % B (the reference's pairing, see
        // add_rowvec_clip_kernel); it joins the residual before the out-projection adds it
        RUN(cx, op_add_rowvec_clip(x0.p, x0.dt, ov, dim, x0.p, x0.dt, a.B, a.F, L, dim, cx.s));
        ov = nullptr;
    }
    TRY(run_linear(cx, w.attn1.out, o, Ci, x2, dim, M, x0, dim, ov, ov ? dim : 0, ov ? M : 0, nullptr, TV(), &w.norm3, xn));
    // x = ff(norm3(x)) + x
    TRY(run_ffn(cx, w.ff1, w.ff2, xn, dim, out, M, x2, blend_mix, blend_other));
    cx.release(mk);
    return 0;
}

// The temporal transformer of a frame-sharded clip, all-to-all form: everything inside the block is per
// pixel, so the ranks swap their frame shards [(b f_local) L][512] for pixel shards [(b F) L/G][512] (one all_to_all of the
// fp32 token stream), run the UNSHARDED block on their pixels -- all frames present, no exchange inside -- and swap the
// fp16 result back.  A rank sends and receives (G-1)/G x (4 + 2) x 512 B per token instead of RECEIVING G x 2C x 2 B per
// token in the K|V all_gather form (C = 320 .. 1280): 4x .. 27x fewer bytes at G = 8.  The AlphaBlender of the block's
// output with the spatial branch (:282) is applied after the way back (op_blend).
bool clip_a2a_enabled() {
    return !policy_is0(P_CLIP_A2A);
}
int run_temporal_tb_pixel_sharded(Ctx& cx, const TemporalTBW& w, const TV& X, const TV& out, const AFwd& a, int L,
                                  const float* blend_mix, const TV& blend_other, const float* ov) {
    const size_t mk = cx.mark();
    const int G = a.comm->world, B = a.B, Fl = a.F, Fg = a.Fg, Lp = L / G, dim = w.dim;
    const size_t esz = X.dt == DT_F32 ? 4 : 2;
    const size_t row_in = (size_t)Lp * dim * esz, row_out = (size_t)Lp * dim * sizeof(half_t);
    const size_t blk_in = al256((size_t)B * Fl * row_in), blk_out = al256((size_t)B * Fl * row_out);
    ws_need(a, 2 * (size_t)G * blk_in);
    char* ws = cx.dry ? nullptr : (char*)a.comm->ws;
    const int Mp = B * Fg * Lp;                       // rows of the pixel shard: all frames of every clip, L/G pixels
    void* xp = cx.alloc((size_t)Mp * dim * esz);
    half_t* yp = cx.h((size_t)Mp * dim);
    half_t* y = blend_mix ? cx.h((size_t)B * Fl * L * dim) : out.m16;
    CTRL_CHECK(cx.dry || y != nullptr, "clip-sharded temporal transformer: the output needs an fp16 tensor");
    if (!cx.dry) {
        // frame shard -> G pixel-shard blocks: block r = pixels [r*Lp, (r+1)*Lp) of every local (clip, frame) row
        for (int r = 0; r < G; ++r)
            HIP_TRY(hipMemcpy2DAsync(ws + (size_t)r * blk_in, row_in, (const char*)X.p + (size_t)r * row_in, (size_t)L * dim * esz,
                                     row_in, (size_t)B * Fl, hipMemcpyDeviceToDevice, cx.s));
        COMM_TRY(a.comm->all_to_all(a.comm->user, 0, (int64_t)((size_t)G * blk_in), (int64_t)blk_in, cx.s));
        // block r of the receive area = rank r's frames [r*Fl, (r+1)*Fl) of my pixels: -> [(b F) Lp][dim]
        for (int r = 0; r < G; ++r)
            HIP_TRY(hipMemcpy2DAsync((char*)xp + (size_t)r * Fl * row_in, (size_t)Fg * row_in, ws + (size_t)(G + r) * blk_in,
                                     (size_t)Fl * row_in, (size_t)Fl * row_in, (size_t)B, hipMemcpyDeviceToDevice, cx.s));
    }
    AFwd a2 = a;
    a2.comm = nullptr; a2.F = Fg; a2.N = B * Fg;
    TV Xp; Xp.p = xp; Xp.dt = X.dt; Xp.m16 = X.dt == DT_F16 ? (half_t*)xp : nullptr;
    TRY(run_temporal_tb(cx, w, Xp, tv16(yp), a2, Lp, nullptr, TV(), ov));
    if (!cx.dry) {
        for (int r = 0; r < G; ++r)       // block r = frames [r*Fl, (r+1)*Fl) of my pixel shard, for rank r
            HIP_TRY(hipMemcpy2DAsync(ws + (size_t)r * blk_out, (size_t)Fl * row_out, (const char*)yp + (size_t)r * Fl * row_out,
                                     (size_t)Fg * row_out, (size_t)Fl * row_out, (size_t)B, hipMemcpyDeviceToDevice, cx.s));
        COMM_TRY(a.comm->all_to_all(a.comm->user, 0, (int64_t)((size_t)G * blk_out), (int64_t)blk_out, cx.s));
        for (int r = 0; r < G; ++r)       // block r of the receive area = rank r's pixels of my frames
            HIP_TRY(hipMemcpy2DAsync((char*)y + (size_t)r * row_out, (size_t)L * dim * sizeof(half_t), ws + (size_t)(G + r) * blk_out,
                                     row_out, row_out, (size_t)B * Fl, hipMemcpyDeviceToDevice, cx.s));
    }
    if (blend_mix)      // AlphaBlender: a * spatial + (1 - a) * temporal
        RUN(cx, op_blend(blend_other.p, blend_other.dt, y, DT_F16, blend_mix, out.p, out.dt, (size_t)B * Fl * L * dim, cx.s));
    cx.release(mk);
    return 0;
}

// one AdapterSpatioTemporal block: in NCHW [N][C][h][w] (in_dt) -> out NCHW [N][C][h*up][w*up] (out_dt)
int run_block(Ctx& cx, const AdapterBlockW& b, const ctrl_adapter_config& c, const AFwd& a, const BlockPre& pre,
              const void* in, void* out, int h, int w) {
    const size_t mk = cx.mark();
    const int N = a.N, C = b.C;
    const bool sr = c.add_spatial_resnet, tr = c.add_temporal_resnet, st = c.add_spatial_transformer, tt = c.add_temporal_transformer;
    const bool has_tf = st || tt;
    half_t* x0 = cx.h((size_t)N * h * w * C);
    RUN(cx, op_nchw_to_nhwc(in, a.in_dt, x0, N, C, h * w, cx.s));
    TV x = tv16(x0);
    int H = h, W = w;
    // resnet time embedding (:206-209) and frame-index embedding (:259-266): precompute_small
    const float* temb = pre.temb;
    const float* femb = pre.femb;
    if (femb && a.comm) femb += (size_t)a.comm->rank * a.F * INNER;      // a sharded rank uses its slice of the global frames
    (void)temb;
    const size_t nl = b.layers.size();
    for (size_t i = 0; i < nl; ++i) {
        const AdapterLayerW& Lw = b.layers[i];
        const int up = (i == 0) ? b.up : 1;
        const bool last = (i + 1 == nl);
        // an fp16 copy of the layer's resnet output is needed only when nothing but a layout change follows it
        const bool need16 = !has_tf;
        if (sr) {
            const float* tp = pre.sres_tp[i];
            TV y = stream_alloc(cx, (size_t)N * H * up * W * up * C, need16 && !tr);
            TRY(run_resnet(cx, Lw.sres, x, y, N, H, W, up, tp, C, 1e-6f));
            x = y; H *= up; W *= up;
        } else if (up > 1 && !tr) {
            // no ResNet to fold the up-sampling into: F.interpolate(scale_factor=2, mode="nearest") (:235-237)
            CTRL_CHECK(cx.dry || x.m16 != nullptr, "adapter: the up-sampling path needs the fp16 map");
            half_t* xu = cx.h((size_t)N * H * 2 * W * 2 * C);
            RUN(cx, op_upsample2x_nhwc(x.m16, xu, N, H, W, C, cx.s));
            x = tv16(xu); H *= 2; W *= 2;
        } else if (up > 1) {
            // the reference up-samples only `if not add_spatial_resnet and not add_temporal_resnet`: with a temporal ResNet
            // alone the block keeps the input size while the zero slots ... are input-sized too; the output tensor the
            // mirror allocates is up-sampled, so refuse rather than return a partly written tensor
            CTRL_FAIL("adapter: sdxl backbone with temporal but no spatial ResNet leaves the block at the input size (:235); not supported");
        }
        if (tr) {
            // with a spatial ResNet in front the AlphaBlender (:229) is folded into the temporal block's last conv
            TV yt = stream_alloc(cx, (size_t)N * H * W * C, false);
            if (need16) yt = tv16(cx.h((size_t)N * H * W * C));       // only a layout change follows: fp16 is enough
            TRY(run_temporal_resnet(cx, Lw.tres, x, yt, a, H * W, C, pre.tres_tp[i], sr ? Lw.res_mix : nullptr));
            x = yt;
        }
        if (has_tf) {
            const int Lt = H * W, M = N * Lt;
            half_t* n = cx.h((size_t)M * C);
            TRY(run_groupnorm(cx, b.norm, x, n, N, Lt, 1e-6f, false));
            // the spatial transformer's token stream: three residual updates on a stream that proj_in starts afresh, so it can be
            // kept in fp16 where the error budget allows (adapter_tok_f16(): measured per workload)
            // Round 5 narrowed it to the blocks it was measured on and pays on -- ONE layer, no temporal transformer (the SDXL adapters):
            // with a temporal transformer behind it or a second layer on top, the fp16 stream costs 15-30 % of the error budget of the
            // small-grid chains (config-5 miniature 1.00e-3 -> 7.1e-4, two-layer SDXL variant 8.5e-4 -> 5.9e-4 with the fp32 stream:
            // profiles/r05_margin_sweep.txt); CTRL_ADAPTER_TOK_F16=force applies it everywhere as rounds 4 did.
            const bool tok16 = st && adapter_tok_f16() && cx.f32stream && !b.tok_f32 && (adapter_tok_f16_forced() || (!tt && nl == 1));
            if (tok16) cx.f32stream = false;
            TV tok = stream_alloc(cx, (size_t)M * INNER, false);
            // the first LayerNorm of the spatial transformer rides on proj_in's epilogue
            half_t* tok_ln = st ? cx.h((size_t)M * INNER) : nullptr;
            TRY(run_linear(cx, b.proj_in, n, C, tok, INNER, M, TV(), 0, nullptr, 0, 0, nullptr, TV(), st ? &Lw.stb.norm1 : nullptr, tok_ln));
            TV smix;
            if (st) {
                // no temporal block: the spatial transformer's result is read once more, as proj_out's fp16 operand -- the last
                // GEMM writes just that (the same rounding of the same fp32 value the mirror of an fp32 master would hold: bit-
                // identical results, one 4-byte-per-element store less); with a temporal block it stays an fp32 stream
                if (tok16 && tt) cx.f32stream = true;      // (what leaves the block towards the temporal branch stays fp32)
                TV t2 = tt ? stream_alloc(cx, (size_t)M * INNER, false) : tv16(cx.h((size_t)M * INNER));
                if (tok16) cx.f32stream = false;
                const int rc_tb = run_basic_tb(cx, Lw.stb, tok, t2, N, Lt, a.e, a.e.Lk == 1 ? pre.stb_ov[i] : nullptr, nullptr, tok_ln);
                if (tok16) cx.f32stream = true;
                if (rc_tb) return rc_tb;
                tok = t2; smix = t2;
            }
            if (tok16) cx.f32stream = true;
            if (tt) {
                TV t3 = stream_alloc(cx, (size_t)M * INNER, false);
                // frame-sharded clip: pixel shards around the temporal block when the transport can and the pixels divide
                const bool a2a = a.comm && a.comm->all_to_all && clip_a2a_enabled() && (Lt % a.comm->world == 0);
                if (a.comm && a.comm->all_to_all && clip_a2a_enabled() && !a2a) {
                    static bool told = false;          // (the K|V all-gather form moves 4-27x the bytes: say so once instead of silently)
                    if (!told) { told = true; fprintf(stderr, "ctrl: clip split: %d pixels per frame do not divide by %d ranks -- temporal transformer falls back to the K|V all-gather form\n", Lt, a.comm->world); }
                }
                // frame-index embedding add (:279) and the temporal block's first LayerNorm in one pass over the tokens (the
                // pixel-sharded form normalises after its exchange instead)
                half_t* t3_ln = a2a ? nullptr : cx.h((size_t)M * INNER);
                if (t3_ln && tok.dt == t3.dt)
                    RUN(cx, op_layernorm_add(tok.p, tok.dt, INNER, femb, INNER, Lt, a.F, t3.p, Lw.ttb.norm_in.g, Lw.ttb.norm_in.b,
                                             t3_ln, INNER, M, INNER, 1e-5f, cx.s));
                else {
                    RUN(cx, op_add_rowvec(tok.p, tok.dt, femb, INNER, t3.p, t3.dt, (size_t)M, INNER, Lt, a.F, cx.s));
                    t3_ln = nullptr;
                }
                if (st) {
                    // AlphaBlender (:282) folded into the temporal block's last GEMM; the blended tokens are consumed
                    // only as proj_out's operand: fp16
                    TV t5 = tv16(cx.h((size_t)M * INNER));
                    if (a2a) TRY(run_temporal_tb_pixel_sharded(cx, Lw.ttb, t3, t5, a, Lt, Lw.tr_mix, smix, pre.ttb_ov[i]));
                    else TRY(run_temporal_tb(cx, Lw.ttb, t3, t5, a, Lt, Lw.tr_mix, smix, pre.ttb_ov[i], t3_ln));
                    tok = t5;
                } else {
                    TV t4 = stream_alloc(cx, (size_t)M * INNER, true);
                    if (a2a) TRY(run_temporal_tb_pixel_sharded(cx, Lw.ttb, t3, t4, a, Lt, nullptr, TV(), pre.ttb_ov[i]));
                    else TRY(run_temporal_tb(cx, Lw.ttb, t3, t4, a, Lt, nullptr, TV(), pre.ttb_ov[i], t3_ln));
                    tok = t4;
                }
            }
            // proj_out + residual (:286-289); the last layer writes the NCHW result directly
            IGemmArgs g = {};
            g.A = tok.m16; g.lda = INNER; g.mode = IG_ROWS; g.Cin = INNER; g.taps = 1;
            g.W = b.proj_out.w; g.M = M; g.Nout = C; g.Ktot = INNER; g.bias = b.proj_out.b;
            g.scale = 1.f;
            set_res(g, x, C);
            if (last) {
                g.nseg = 1;
                g.seg[0] = IGemmSeg{out, Lt, 0, C, SEG_TRANSPOSED, a.out_dt, Lt, 0, a.out_map};
                RUN(cx, op_igemm(g, cx.s));
            } else {
                TV y = stream_alloc(cx, (size_t)M * C, true);      // next layer's shortcut conv reads the fp16 copy
                set_out(g, y, C, C);
                RUN(cx, op_igemm(g, cx.s));
                x = y;
            }
        } else if (last) {
            RUN(cx, op_nhwc_to_nchw(x.m16, out, a.out_dt, N, C, H * W, 1.f, cx.s, a.out_map));
        }
    }
    cx.release(mk);
    return 0;
}

}  // namespace

struct ctrl_adapter : PlanBase {
    AdapterW w;
    std::unique_ptr<Packer> packer;
    Arena arena;
    KvCache kvc;                         // text K/V cache (ctrl_*_text_cache)
    // frame scatter map: pinned staging + device copy, re-uploaded (on the caller's stream) only when it changes
    static constexpr int kMaxMap = 1024;
    int* map_host = nullptr;
    int* map_dev = nullptr;
    std::vector<int> map_cur;
    // Lanes: the adapter blocks are independent of each other (ctrl_adapter.py:181-205), so blocks of different
    // resolutions run on different HIP streams (lane 0 = the caller's stream) and the small low-resolution blocks fill
    // the CUs the big ones leave idle at their tile-wave tails.  Forked / joined with events: hipGraph-capturable.
    static constexpr int kLanes = 6;             // 4 pyramid levels (+ 2 when the three top-level blocks get a lane each)
    static constexpr int kLevelLanes = 4;
    hipStream_t side[kLanes - 1] = {};
    hipEvent_t fork_ev = nullptr, join_ev[kLanes - 1] = {};
    size_t lane_need[kLanes] = {};
    size_t slot_need[13] = {};                   // workspace of the block of slot i (dry pass): siblings of a group lie side by side
    int init_lanes() {
        for (int i = 0; i < kLanes - 1; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        return 0;
    }
    ~ctrl_adapter() {
        for (int i = 0; i < kLanes - 1; ++i) {
            if (side[i]) (void)hipStreamDestroy(side[i]);
            if (join_ev[i]) (void)hipEventDestroy(join_ev[i]);
        }
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        if (packer) packer->release_all();
        if (map_host) (void)hipHostFree(map_host);
        if (map_dev) (void)hipFree(map_dev);
    }
};

namespace {

struct AdapterCall {
    const void* const* ins; int in_dt; int N, H0, W0, F;
    const float* t; int t_count;
    const void* ehs; int ehs_dt; int ehs_batch; int Lk;
    void* const* outs; int out_dt;
    const int* map_dev; const int32_t* map_host; int N_out;     // frame scatter (null / null / N when off)
    ctrl_adapter* plan; int nlanes;                             // stream lanes (1 = everything on the caller's stream)
    const hipEvent_t* in_ev;                                    // optional [13]: input slot i is ready when in_ev[i] fires
    ctrl_clip_comm* comm;                                       // frame-sharded clip (null = whole clips); F / N are then local
    size_t* ws_peak;
};

size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

int adapter_run(Ctx& cx, const AdapterW& w, const AdapterCall& k) {
    TRY(begin_forward(cx));
    const ctrl_adapter_config& c = w.cfg;
    AFwd a;
    a.N = k.N; a.F = k.F; a.B = k.N / k.F; a.t = k.t; a.t_count = k.t_count; a.in_dt = k.in_dt; a.out_dt = k.out_dt;
    a.out_map = k.map_dev;
    a.comm = k.comm; a.Fg = k.comm ? k.F * k.comm->world : k.F; a.ws_peak = k.ws_peak;
    // frames of the dense output that no input frame lands on (zero-filled per slot below), as [begin, end) runs
    std::vector<std::pair<int, int>> holes;
    if (k.map_host) {
        std::vector<char> hit(k.N_out, 0);
        for (int j = 0; j < k.N; ++j) hit[k.map_host[j]] = 1;
        for (int f = 0; f < k.N_out;) {
            if (hit[f]) { ++f; continue; }
            int e = f;
            while (e < k.N_out && !hit[e]) ++e;
            holes.push_back({f, e});
            f = e;
        }
    }
    auto fill_holes = [&](void* out, size_t frame_elems) -> int {
        for (const auto& r : holes)
            RUN(cx, op_fill_zero((char*)out + (size_t)r.first * frame_elems * dt_size(k.out_dt),
                                 (size_t)(r.second - r.first) * frame_elems * dt_size(k.out_dt), cx.s));
        return 0;
    };
    const int cross = c.cross_attention_dim;
    // encoder hidden states: fp16 copy for the K/V projection GEMM, fp32 copy for the single-key path
    a.e.batch = k.ehs_batch; a.e.Lk = k.Lk; a.e.cross = cross;
    const size_t ne = (size_t)k.ehs_batch * k.Lk * cross;
    if (c.add_spatial_transformer || c.add_temporal_transformer) {
        half_t* e16 = cx.h(ne);
        RUN(cx, op_nchw_to_nhwc(k.ehs, k.ehs_dt, e16, 1, 1, (int)ne, cx.s));
        a.e.h16 = e16;
        if (k.Lk == 1) {
            float* e32 = cx.f(ne);
            RUN(cx, op_nhwc_to_nchw(e16, e32, DT_F32, 1, 1, (int)ne, 1.f, cx.s));
            a.e.f32 = e32;
        }
    }
    if (c.add_temporal_transformer) {
        // time_context = first frame of each clip (:246-249).  With broadcast encoder states (batch 1, what the
        // pipelines pass) every clip shares it.  Per-sample states: one context row per clip (row b*F of the states); with
        // more than one clip the reference hands them to the block ordered (pixel, clip) while the block's rows are
        // (clip, pixel) -- reproduced by op_add_rowvec_clip in run_temporal_tb (DESIGN.md "known quirks").
        CTRL_CHECK(k.Lk == 1, "adapter: the temporal transformer path requires single-token encoder_hidden_states");
        CTRL_CHECK(k.ehs_batch == 1 || !k.comm,
                   "adapter: a frame-sharded clip needs broadcast encoder_hidden_states ([1,1,C]): the first frame's context lives on one rank");
        a.e_first = a.e;
        a.e_first.batch = k.ehs_batch == 1 ? 1 : a.B;      // row b = first frame of clip b, or the broadcast vector
        a.e_first.row_ld = (long)a.F * cross;
    }
    // SD-1.5 pyramid below (H0, W0)
    static const int slot_c[12] = {320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280};
    static const int slot_f[12] = {1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8};
    const int up = c.backbone_sdxl ? 2 : 1;
    // ---- everything that depends on (timestep, context, frame index) only, for all blocks, before the lanes fork ----
    const bool run_mid = w.has_mid && k.ins[12] && k.outs[12];
    std::vector<const AdapterBlockW*> all_blocks;
    for (const AdapterBlockW& b : w.blocks) all_blocks.push_back(&b);
    if (run_mid) all_blocks.push_back(&w.mid);
    std::vector<BlockPre> pre;
    TRY(precompute_small(cx, c, all_blocks, a, &pre));
    // ---- lanes: one per pyramid level (slot_f = 1, 2, 4, 8 + mid); each lane owns a disjoint workspace region ----
    const int nl = k.nlanes;
    ctrl_adapter* P = k.plan;
    hipStream_t const main_s = cx.s;
    const size_t common_end = (cx.mark() + 255) & ~(size_t)255;
    size_t lane_base[ctrl_adapter::kLanes];
    {
        size_t off = common_end;
        for (int l = 0; l < ctrl_adapter::kLanes; ++l) { lane_base[l] = off; off += cx.dry ? 0 : P->lane_need[l]; }
        if (cx.dry) for (int l = 0; l < ctrl_adapter::kLanes; ++l) P->lane_need[l] = 0;
    }
    if (!cx.dry && nl > 1) {
        HIP_TRY(hipEventRecord(P->fork_ev, main_s));
        for (int l = 1; l < nl; ++l) HIP_TRY(hipStreamWaitEvent(P->side[l - 1], P->fork_ev, 0));
    }
    // one lane per pyramid level; with more than kLevelLanes lanes (CTRL_ADAPTER_SPLIT_TOP) the second and third block of the
    // top level (the three largest, equal-shaped chains of the step) get a lane of their own
    int top_seen = 0;
    auto lane_of = [&](int f) {
        int l = f == 1 ? 0 : (f == 2 ? 1 : (f == 4 ? 2 : 3));
        if (f == 1 && nl > ctrl_adapter::kLevelLanes) { l = top_seen == 0 ? 0 : ctrl_adapter::kLevelLanes - 1 + top_seen; ++top_seen; }
        return l % nl;
    };
    // frame-sharded clip: lane l exchanges through the l-th transport of the chain (ctrl_clip_comm::next_lane)
    ctrl_clip_comm* lane_comm[ctrl_adapter::kLanes] = {};
    {
        ctrl_clip_comm* cc = k.comm;
        for (int l = 0; l < ctrl_adapter::kLanes; ++l) { lane_comm[l] = cc ? cc : k.comm; if (cc) cc = cc->next_lane; }
    }
    // ---- jobs: one per slot that has a block, in slot order; consecutive jobs of one lane with the same (C, h, w) are SIBLINGS --
    //      the adapters of one location (ctrl_adapter.py:119-139: three per location, equal shapes for the last two or all three;
    //      the mid block joins the 8^2 ones).  Siblings are recorded and replayed in lock-step, so every GEMM / norm / attention of
    //      theirs leaves as ONE grouped launch (ops.h: OpCollector): 2-4 x the tiles per launch at the 64^2 .. 8^2 levels, where
    //      a single block fills less than the chip, and 40 % fewer launches per step.  Each sibling owns a disjoint workspace
    //      region (sized by the dry pass: slot_need).  Every problem is computed as it would be alone; with CTRL_GROUP=2 the GEMM dispatcher also
    //      picks the tile a lone problem would get and the forward is bit-identical to the one-by-one forward (CTRL_GROUP=0).  In the default
    //      mode (1) the dispatcher sizes the tile for the whole group -- another tile family for some GEMMs, last-bit differences
    //      (<= 3e-4 rel-inf asserted by tests/test_gpu_e2e.py::test_grouped_launches_are_bit_identical_and_fewer), like another batch size.
    struct Job { int slot, lane, h, wd, C; const AdapterBlockW* bw; const BlockPre* bp; const void* in; void* out; size_t frame_elems; };
    std::vector<Job> jobs;
    size_t bi = 0;
    for (int i = 0; i < 12; ++i) {
        const int h = std::max(k.H0 / slot_f[i], 1), wd = std::max(k.W0 / slot_f[i], 1);
        const bool has = bi < w.slot_ids.size() && w.slot_ids[bi] == i;
        if (has) {
            jobs.push_back({i, lane_of(slot_f[i]), h, wd, slot_c[i], &w.blocks[bi], &pre[bi], k.ins[i], k.outs[i], (size_t)slot_c[i] * h * up * wd * up});
            ++bi;
        } else {
            // torch.zeros_like(down_block_res_samples[i])  (ctrl_adapter.py:193): input-sized, not up-sampled
            RUN(cx, op_fill_zero(k.outs[i], (size_t)k.N_out * slot_c[i] * h * wd * dt_size(k.out_dt), cx.s));
        }
    }
    if (run_mid) {
        const int h = std::max(k.H0 / 8, 1), wd = std::max(k.W0 / 8, 1);
        jobs.push_back({12, lane_of(8), h, wd, 1280, &w.mid, &pre.back(), k.ins[12], k.outs[12], (size_t)1280 * h * up * wd * up});
    }
    const bool grouping = group_launches_enabled() && !k.comm;
    size_t lane_off[ctrl_adapter::kLanes];      // dry pass: how far the chains of a lane have got inside its region (they run one after the other: max)
    for (int l = 0; l < ctrl_adapter::kLanes; ++l) lane_off[l] = 0;
    for (size_t j0 = 0; j0 < jobs.size();) {
        size_t j1 = j0 + 1;
        while (grouping && j1 < jobs.size() && j1 - j0 < (size_t)kMaxGroup && jobs[j1].lane == jobs[j0].lane && jobs[j1].C == jobs[j0].C &&
               jobs[j1].h == jobs[j0].h && jobs[j1].wd == jobs[j0].wd) ++j1;
        const int ng = (int)(j1 - j0), lane = jobs[j0].lane;
        cx.s = lane == 0 ? main_s : P->side[lane - 1];
        a.comm = lane_comm[lane];
        OpList lists[kMaxGroup];
        size_t off = lane_base[lane];
        for (int g = 0; g < ng; ++g) {
            const Job& jb = jobs[j0 + g];
            if (!cx.dry && k.in_ev) HIP_TRY(hipStreamWaitEvent(cx.s, k.in_ev[jb.slot], 0));     // fused step: producer still running
            cx.ar->off = off;
            if (cx.dry) cx.ar->peak = off;
            cx.rec = (ng > 1 && !cx.dry) ? &lists[g] : nullptr;
            int rc = run_block(cx, *jb.bw, c, a, *jb.bp, jb.in, jb.out, jb.h, jb.wd);
            if (!rc) rc = fill_holes(jb.out, jb.frame_elems);
            cx.rec = nullptr;
            if (rc) { cx.s = main_s; return rc; }
            if (cx.dry) P->slot_need[jb.slot] = (cx.ar->peak - off + 255) & ~(size_t)255;
            off += P->slot_need[jb.slot];
        }
        if (cx.dry && off - lane_base[lane] > P->lane_need[lane]) P->lane_need[lane] = off - lane_base[lane];
        if (ng > 1 && !cx.dry) {
            const int rc = replay_lockstep(lists, ng);
            if (rc) { cx.s = main_s; return rc; }
        }
        cx.s = main_s;
        j0 = j1;
    }
    (void)lane_off;
    if (!cx.dry && nl > 1) {
        for (int l = 1; l < nl; ++l) {
            HIP_TRY(hipEventRecord(P->join_ev[l - 1], P->side[l - 1]));
            HIP_TRY(hipStreamWaitEvent(main_s, P->join_ev[l - 1], 0));
        }
    }
    if (cx.dry) {
        size_t tot = common_end;
        for (int l = 0; l < ctrl_adapter::kLanes; ++l) tot += P->lane_need[l];
        cx.ar->peak = tot;
    }
    cx.ar->off = common_end;
    return 0;
}

}  // namespace

extern "C" {

int ctrl_adapter_param_count(const ctrl_adapter_config* cfg) {
    if (!cfg) return -1;
    SpecCollector sc; AdapterW w;
    if (build_adapter(sc, *cfg, &w)) return -1;
    return (int)sc.entries.size();
}

int ctrl_adapter_param_spec(const ctrl_adapter_config* cfg, int i, char* name, int name_len, int64_t shape[6], int* ndim) {
    CTRL_CHECK(cfg && name && shape && ndim, "param_spec: null argument");
    SpecCollector sc; AdapterW w;
    TRY(build_adapter(sc, *cfg, &w));
    CTRL_CHECK(i >= 0 && i < (int)sc.entries.size(), "param_spec: index out of range");
    std::strncpy(name, sc.entries[i].name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
    *ndim = (int)sc.entries[i].shape.size();
    for (int k = 0; k < 6; ++k) shape[k] = k < *ndim ? sc.entries[i].shape[k] : 1;
    return 0;
}

int ctrl_adapter_create(const ctrl_adapter_config* cfg, const ctrl_tensor_ref* tensors, int n_tensors, void* stream,
                        ctrl_adapter** out) {
    CTRL_CHECK(cfg && tensors && out, "adapter_create: null argument");
    std::unique_ptr<ctrl_adapter> h(new ctrl_adapter());
    TRY(h->init_base(n_tensors > 0 ? tensors[0].data : nullptr));
    DeviceGuard dg(h->device);
    h->packer.reset(new Packer(tensors, n_tensors, (hipStream_t)stream));
    int rc = build_adapter(*h->packer, *cfg, &h->w);
    if (rc) return rc;
    TRY(h->init_lanes());
    TRY(h->packer->finish());
    *out = h.release();
    return 0;
}

void ctrl_adapter_destroy(ctrl_adapter* h) { delete h; }

int ctrl_adapter_trim(ctrl_adapter* h) {
    CTRL_CHECK(h, "adapter_trim: null plan");
    DeviceGuard dg(h->device);
    // trim synchronises the device and frees retired blocks: not while a forward of this plan is being recorded (the sync would
    // invalidate that capture), and never concurrently with a forward on the same plan (a plan has no lock: include/ctrl_hip.h)
    CTRL_CHECK(!h->capture_active(), "adapter_trim: a stream capture of this plan's forward is in progress");
    HIP_TRY(hipDeviceSynchronize());
    h->arena.trim();
    h->kvc.trim();
    return 0;
}

int ctrl_adapter_selection(ctrl_adapter* h, char* buf, int len) {
    CTRL_CHECK(h && buf && len > 0, "adapter_selection: null argument");
    int n32 = 0, n = 0;
    for (const AdapterBlockW& b : h->w.blocks) { ++n; n32 += b.tok_f32 ? 1 : 0; }
    if (h->w.has_mid) { ++n; n32 += h->w.mid.tok_f32 ? 1 : 0; }
    snprintf(buf, (size_t)len, "blocks=%d token_stream_fp32_blocks=%d (outlier norm scales, gate %.1f) token_stream_fp16=%d stream_f32=%d ff_fused=%d", n, n32,
             kNormSpreadGate, adapter_tok_f16() ? 1 : 0, stream_f32_enabled() ? 1 : 0, policy_is0(P_FF_FUSED) ? 0 : 1);
    return 0;
}

int ctrl_adapter_text_cache(ctrl_adapter* h, int mode) {
    CTRL_CHECK(h && mode >= 0 && mode <= 2, "adapter_text_cache: mode must be 0 (off), 1 (keep) or 2 (reuse)");
    h->kvc.mode = mode;
    return 0;
}

static int adapter_forward_impl(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                                const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                                int ehs_batch, int Lk, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                                void* stream, const hipEvent_t* in_ev = nullptr, ctrl_clip_comm* comm = nullptr) {
    CTRL_CHECK(h && ins && outs && timesteps, "adapter_forward: null argument");
    if (comm) {
        CTRL_CHECK(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world, "clip_sharded: bad rank / world");
        CTRL_CHECK(comm->all_gather && comm->all_reduce_sum_f32 && comm->halo_exchange, "clip_sharded: missing transport callback");
        CTRL_CHECK(num_frames * comm->world <= 32, "clip_sharded: at most 32 frames per clip");
        CTRL_CHECK(((uintptr_t)comm->ws & 255) == 0, "clip_sharded: the exchange workspace must be 256-byte aligned");
    }
    CTRL_CHECK(N >= 1 && H0 >= 1 && W0 >= 1 && num_frames >= 1 && N % num_frames == 0,
               "adapter_forward: batch must be a multiple of num_frames");
    CTRL_CHECK(t_count == 1 || t_count == N, "adapter_forward: need 1 or N timesteps");
    CTRL_CHECK(num_frames <= 32, "adapter_forward: at most 32 frames per clip");
    const bool needs_ehs = h->w.cfg.add_spatial_transformer || h->w.cfg.add_temporal_transformer;
    CTRL_CHECK(!needs_ehs || (encoder_hidden_states && Lk >= 1 && (ehs_batch == 1 || ehs_batch == N)),
               "adapter_forward: encoder_hidden_states batch must be 1 or N");
    for (int i = 0; i < 12; ++i) CTRL_CHECK(ins[i] && outs[i], "adapter_forward: null slot pointer");
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard dg(h->device);
    bool capturing = false;
    TRY(h->enter(s, &capturing));
    // a KEEP forward that still has to allocate its text K/V buffers cannot be recorded into a hipGraph: refuse up front,
    // before any lane is forked (an error in the middle of a capture leaves unjoined streams behind)
    // (single-key states, Lk == 1 -- what the video pipelines pass -- never project text K/V: their slots stay empty for ever and
    // need none)
    CTRL_CHECK(!(capturing && needs_ehs && Lk > 1 && h->kvc.mode == KvCache::KEEP && h->kvc.slots.empty()),
               "text K/V cache: the first KEEP forward allocates its buffers and cannot run under stream capture -- run it "
               "eagerly once, then capture");
    const int* map_dev = nullptr;
    if (frame_pos) {
        CTRL_CHECK(N_out >= N && N_out <= ctrl_adapter::kMaxMap, "adapter_forward_scatter: need N <= N_out <= 1024");
        std::vector<char> seen(N_out, 0);
        for (int j = 0; j < N; ++j) {
            CTRL_CHECK(frame_pos[j] >= 0 && frame_pos[j] < N_out && !seen[frame_pos[j]],
                       "adapter_forward_scatter: frame positions must be distinct and < N_out");
            seen[frame_pos[j]] = 1;
        }
        if (!h->map_dev) {
            HIP_TRY(hipHostMalloc((void**)&h->map_host, sizeof(int) * ctrl_adapter::kMaxMap, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void**)&h->map_dev, sizeof(int) * ctrl_adapter::kMaxMap));
        }
        if (h->map_cur.size() != (size_t)N || !std::equal(h->map_cur.begin(), h->map_cur.end(), frame_pos)) {
            // the staging buffer may still be the source of an in-flight copy of the previous map
            HIP_TRY(hipStreamSynchronize(s));
            h->map_cur.assign(frame_pos, frame_pos + N);
            std::copy(frame_pos, frame_pos + N, h->map_host);
            HIP_TRY(hipMemcpyAsync(h->map_dev, h->map_host, sizeof(int) * N, hipMemcpyHostToDevice, s));
        }
        map_dev = h->map_dev;
    } else {
        N_out = N;
    }
    // lanes are off while the per-launch profiler is recording (overlapping kernels make per-kernel times meaningless)
    const bool split_top = policy_int(P_ADAPTER_SPLIT_TOP, 0) != 0;
    const int env_lanes = policy_int(P_ADAPTER_LANES, split_top ? (int)ctrl_adapter::kLanes : (int)ctrl_adapter::kLevelLanes);
    // frame-sharded clips: the exchanges of one communicator must be issued and executed in the same order on every rank, so
    // there are as many lanes as the caller chained transports (ctrl_clip_comm::next_lane; one = everything on the caller's stream)
    int comm_lanes = 0;
    for (ctrl_clip_comm* cc = comm; cc; cc = cc->next_lane) {
        CTRL_CHECK(cc->rank == comm->rank && cc->world == comm->world && cc->all_gather && cc->all_reduce_sum_f32 && cc->halo_exchange &&
                   (((uintptr_t)cc->ws & 255) == 0) && (!comm->all_to_all == !cc->all_to_all),
                   "clip_sharded: the transports of a lane chain must agree in rank / world / callbacks and have 256-byte aligned workspaces");
        ++comm_lanes;
        CTRL_CHECK(comm_lanes <= (int)ctrl_adapter::kLanes, "clip_sharded: transport chain longer than the lanes there are (or cyclic)");
    }
    const int want_lanes = std::min(std::max(env_lanes, 1), (int)ctrl_adapter::kLanes);
    // Under stream capture a frame-sharded forward stays on ONE lane: RCCL refuses a second communicator on a forked stream inside
    // one capture (hipErrorStreamCaptureUnsupported), and that error would arrive mid-capture, after the lanes are forked, leaving
    // unjoined streams behind an invalidated capture.  Eager calls use the chain the caller passed.
    const int comm_use = capturing ? 1 : std::min(comm_lanes, (int)ctrl_adapter::kLevelLanes);
    const int nlanes = g_prof_on ? 1 : (comm ? std::min(want_lanes, comm_use) : want_lanes);
    size_t ws_peak = 0;
    AdapterCall k = {ins, in_dtype, N, H0, W0, num_frames, timesteps, t_count, encoder_hidden_states, ehs_dtype,
                     ehs_batch, Lk, outs, out_dtype, map_dev, frame_pos, N_out, h, nlanes, in_ev, comm, &ws_peak};
    h->arena.off = 0; h->arena.peak = 0;
    Ctx dry{&h->arena, s, true};
    dry.f32stream = stream_f32_enabled();
    dry.h1_f16 = adapter_h1_f16();
    dry.kvc = &h->kvc; h->kvc.next = 0;
    if (h->kvc.mode == KvCache::REUSE)
        CTRL_CHECK(h->kvc.key_batch == ehs_batch && h->kvc.key_Lk == Lk, "adapter_forward: text K/V cache was kept for another batch / prompt length");
    TRY(adapter_run(dry, h->w, k));
    bool ws_small = false;
    for (ctrl_clip_comm* cc = comm; cc; cc = cc->next_lane) ws_small = ws_small || (size_t)cc->ws_bytes < ws_peak;
    if (comm && ws_small) {
        comm->ws_needed = (int64_t)ws_peak;
        ctrl_set_error("clip_sharded: exchange workspace too small (need " + std::to_string(ws_peak) + " bytes); retry with ws_needed");
        return 2;
    }
    TRY(h->arena.ensure(workspace_bytes(dry)));
    h->arena.off = 0;
    Ctx cx{&h->arena, s, false};
    cx.f32stream = dry.f32stream;
    cx.h1_f16 = dry.h1_f16;
    cx.stats_total = dry.stats_total;
    cx.kvc = &h->kvc; h->kvc.next = 0;
    cx.capturing = capturing;
    TRY(adapter_run(cx, h->w, k));
    if (h->kvc.mode == KvCache::KEEP) { h->kvc.key_batch = ehs_batch; h->kvc.key_Lk = Lk; }
    return h->leave(s, capturing);
}

int ctrl_adapter_forward(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                         const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                         int ehs_batch, int Lk, void* const* outs, int out_dtype, void* stream) {
    return adapter_forward_impl(h, ins, in_dtype, N, H0, W0, num_frames, timesteps, t_count, encoder_hidden_states,
                                ehs_dtype, ehs_batch, Lk, outs, out_dtype, nullptr, N, stream);
}

int ctrl_adapter_forward_scatter(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                                 const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                                 int ehs_batch, int Lk, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                                 void* stream) {
    CTRL_CHECK(frame_pos, "adapter_forward_scatter: null frame_pos");
    return adapter_forward_impl(h, ins, in_dtype, N, H0, W0, num_frames, timesteps, t_count, encoder_hidden_states,
                                ehs_dtype, ehs_batch, Lk, outs, out_dtype, frame_pos, N_out, stream);
}

int ctrl_adapter_forward_clip_sharded(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                                      const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                                      int ehs_batch, int Lk, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                                      ctrl_clip_comm* comm, void* stream) {
    CTRL_CHECK(comm, "adapter_forward_clip_sharded: null comm");
    return adapter_forward_impl(h, ins, in_dtype, N, H0, W0, num_frames, timesteps, t_count, encoder_hidden_states,
                                ehs_dtype, ehs_batch, Lk, outs, out_dtype, frame_pos, frame_pos ? N_out : N, stream, nullptr, comm);
}

}  // extern "C"

// Fused-step half (plan_fused.cpp): like ctrl_adapter_forward[_scatter], but input slot i may still be in production on
// another stream -- the block that consumes it waits for in_ev[i] on its lane (in_ev = null: inputs are ready)
int adapter_forward_events(ctrl_adapter* h, const void* const* ins, int in_dtype, int N, int H0, int W0, int num_frames,
                           const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype,
                           int ehs_batch, int Lk, void* const* outs, int out_dtype, const int32_t* frame_pos, int N_out,
                           void* stream, const hipEvent_t* in_ev) {
    return adapter_forward_impl(h, ins, in_dtype, N, H0, W0, num_frames, timesteps, t_count, encoder_hidden_states,
                                ehs_dtype, ehs_batch, Lk, outs, out_dtype, frame_pos, frame_pos ? N_out : N, stream, in_ev);
}
