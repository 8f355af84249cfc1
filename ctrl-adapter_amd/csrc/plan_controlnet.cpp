// ControlNet forward orchestrator (C++ host code, enqueues the gfx950 kernels of libctrlhip).
//
// Restates the control flow of the reference's ControlNetModel (controlnet/controlnet.py): constructor
// :179-438 fixes the module tree / parameter names, forward :662-881 fixes the order of operations:
//   time embedding :735-758 -> conv_in / skip flags :802-811 -> + conditioning embedding :816-817 ->
//   down blocks :820-833 -> mid block :836-846 -> 13 zero-convs :850-858 -> scaling :861-868.
// Only the configuration the hot path reaches is supported (no class/addition embeddings, rgb order,
// no attention mask, global_pool_conditions=False); anything else fails loudly at create().
#include "plan_common.h"
#include <cmath>
#include <cstring>
#include <memory>

namespace {

struct DownBlockW {
    std::vector<ResnetW> resnets;
    std::vector<Norm> tnorm;            // Transformer2DModel.norm
    std::vector<ConvW> proj_in, proj_out;
    std::vector<BasicTBW> tb;
    bool has_attn = false, has_down = false;
    ConvW down;
    int Cin = 0, Cout = 0;
};

struct ControlNetW {
    ctrl_controlnet_config cfg;
    ConvD conv_in;
    Lin te1, te2;                       // time_embedding.linear_1/2
    Lin temb_cat;                       // every resnet's time_emb_proj, concatenated along N
    // conditioning embedder
    ConvD ce_direct[4]; int n_direct = 0;
    ConvW ce_gemm[5]; int n_gemm = 0;   // remaining layers incl. conv_out (last)
    std::vector<int> ce_chain_c, ce_chain_stride;   // per layer: Cout, stride
    DownBlockW down[4];
    ResnetW mid_r0, mid_r1;
    Norm mid_tnorm; ConvW mid_pin, mid_pout; BasicTBW mid_tb;
    std::vector<ConvW> zero_convs;      // 12 down + 1 mid
    int temb_total = 0;
    // split-operand convolutions (CTRL_CN_SPLIT, default on): every convolution fed by a GroupNorm output or by the fp32
    // residual stream takes its A operand as [hi | lo] fp16 halves, so the operand is exact to ~2^-22 instead of 2^-11.
    // The ControlNet's ~25 residual branches otherwise accumulate the fp16 operand rounding to ~1e-3 rel-inf on its outputs
    // (tests/experiments/fp16_error_budget.py), which the adapter chain inherits; costs 2 x the MFMA work of those convs.
    bool split = true;
    // what plan creation selected (ctrl_controlnet_selection): levels with split operands, levels whose ResNet 3x3 convolutions take them, and
    // the largest max|gamma| / median|gamma| of the checkpoint's normalisation scales that decided it
    int sel_levels = 0, sel_res_levels = 0;
    float sel_norm_spread = 0.f;
    bool sel_outliers = false;
};

bool cn_split_enabled() {
    return !policy_is0(P_CN_SPLIT) && stream_f32_enabled();
}
// How deep the split goes (CTRL_CN_SPLIT_LEVELS, default 3): down blocks 0 .. levels-1 take split operands (mid block =
// level 4); the 13 zero-convs always do.  Round 3: the 8x8 level (down block 3 + mid block) is where a split conv costs
// most -- M = 64 rows per image, split-K 16, 0.07-0.19 of the MFMA peak -- and, by tests/experiments/fp16_error_budget.py
// with the rounding points of each selection, adds least: plain operands there leave the ControlNet outputs / the chain at
// the all-exact level (+0..5e-5), while also un-splitting the 16x16 level costs ~1e-4 -- measured on the GPU at the SVD-16
// shapes: chain mid output 1.01e-3 with levels = 2 (profiles/r03_split_levels.md), over the bound.
int cn_split_levels() {
    const int v = policy_int(P_CN_SPLIT_LEVELS, 3);
    return v < 0 ? 0 : (v > 5 ? 5 : v);
}
// The same depth for the ResNets' 3x3 convolutions alone (CTRL_CN_SPLIT_RESNET_LEVELS, default 1 since round 6): the 3x3 convolutions of the
// 640- and 1280-channel levels take PLAIN operands -- half the matrix work of 12 of the most expensive split launches -- while the 1x1
// shortcuts, proj_in / proj_out, down-samplers and zero-convs of those levels, and everything of level 0, keep the split.  History: the CPU
// emulation per conv kind said so in round 4 (tests/experiments/split_per_conv.py); round 5 ran the full GPU suite with it and reverted it over
// ONE tensor of the config-5 miniature chain at 1.001e-3 -- then confined the adapter's fp16 token stream, which is what that tensor had
// really been paying for (1.00e-3 -> 7.1e-4).  Round 6 measured every selectable rounding point on its own, on three chains against the fp32
// CPU restatement (tests/error_attribution.py -> profiles/r06_error_attribution.md): with "1" the SVD-16, SDXL and config-5 miniature chains move by
// +0 / +2.4e-5 / +0 (8.52e-4 / 6.92e-4 / 8.76e-4; the ControlNet's own outputs 7.06e-4 / 6.14e-4 / 6.46e-4), for -1.6 ms per eager SVD-16
// step; "0" (level 0 plain too) is another -0.4 ms but puts the ControlNet's outputs at 8.6e-4, and un-splitting the 1x1 kinds
// (CTRL_CN_SPLIT_LEVELS=2) moves the chains by +2e-5...1e-4 -- both left alone.  "3" = the round-5 default.
int cn_split_resnet_levels() {
    const char* e = policy_raw(P_CN_SPLIT_RESNET_LEVELS);
    const int lv = cn_split_levels();
    if (!e) return lv < 1 ? lv : 1;
    const int v = atoi(e);
    return v < 0 ? 0 : (v > lv ? lv : v);
}
// CTRL_CN_SPLIT=dup: the first form of the split (weights packed twice, [hi | lo] walked as one long K) for A/B runs
bool cn_split_paired() {
    const char* e = policy_raw(P_CN_SPLIT);
    return !(e && e[0] == 'd');
}

int build_transformer2d(ParamSink& ps, const std::string& pre, int C, int heads, int cross, Norm* n, ConvW* pin,
                        ConvW* pout, BasicTBW* tb, bool dup) {
    TRY(ps.norm(pre + ".norm", C, n));
    TRY(ps.conv(pre + ".proj_in", C, C, 1, false, pin, dup));
    TRY(build_basic_tb(ps, pre + ".transformer_blocks.0", C, heads, C / heads, cross, tb));
    TRY(ps.conv(pre + ".proj_out", C, C, 1, false, pout, dup));
    return 0;
}

int build_controlnet(ParamSink& ps, const ctrl_controlnet_config& c, ControlNetW* w) {
    w->cfg = c;
    w->split = cn_split_enabled();
    const bool dup = w->split;
    const int c0 = c.block_out_channels[0], temb_dim = 4 * c0;
    CTRL_CHECK(c.in_channels == 4 || c.in_channels == 3, "controlnet: in_channels must be 3 or 4 (direct stem kernel)");
    CTRL_CHECK(c.conditioning_channels == 3 || c.conditioning_channels == 4, "controlnet: conditioning_channels must be 3 or 4");
    CTRL_CHECK(c.layers_per_block >= 1 && c.layers_per_block <= 4, "controlnet: layers_per_block out of range");
    for (int i = 0; i < 4; ++i) {
        CTRL_CHECK(c.block_out_channels[i] % 64 == 0, "controlnet: block_out_channels must be multiples of 64");
        const int d = c.block_out_channels[i] / c.num_attention_heads;
        CTRL_CHECK(!c.down_block_has_attn[i] || d == 40 || d == 64 || d == 80 || d == 160,
                   "controlnet: head_dim must be one of 40/64/80/160");
    }
    TRY(ps.conv_direct("conv_in", c0, c.in_channels, &w->conv_in));
    TRY(ps.linear("time_embedding.linear_1", temb_dim, c0, true, false, &w->te1));
    TRY(ps.linear("time_embedding.linear_2", temb_dim, temb_dim, true, false, &w->te2));

    // conditioning embedder (controlnet/controlnet.py:62-104): conv_in, (same, stride-2) x3, conv_out
    {
        const int* cc = c.cond_embed_channels;
        std::vector<std::string> names = {"controlnet_cond_embedding.conv_in"};
        std::vector<int> cin = {c.conditioning_channels}, cout = {cc[0]}, stride = {1};
        for (int i = 0; i < 3; ++i) {
            names.push_back("controlnet_cond_embedding.blocks." + std::to_string(2 * i));
            cin.push_back(cc[i]); cout.push_back(cc[i]); stride.push_back(1);
            names.push_back("controlnet_cond_embedding.blocks." + std::to_string(2 * i + 1));
            cin.push_back(cc[i]); cout.push_back(cc[i + 1]); stride.push_back(2);
        }
        names.push_back("controlnet_cond_embedding.conv_out");
        cin.push_back(cc[3]); cout.push_back(c0); stride.push_back(1);
        bool gemm_phase = false;
        for (size_t i = 0; i < names.size(); ++i) {
            const bool can_gemm = (cin[i] % 32 == 0) && (cout[i] >= 64);
            if (can_gemm) gemm_phase = true;
            if (!gemm_phase) {
                CTRL_CHECK(w->n_direct < 4, "controlnet: too many small conditioning-embedder layers");
                CTRL_CHECK(i == 0 || cin[i] == 16 || cin[i] == 32, "controlnet: unsupported conditioning embedder width");
                TRY(ps.conv_direct(names[i], cout[i], cin[i], &w->ce_direct[w->n_direct++]));
            } else {
                CTRL_CHECK(can_gemm && w->n_gemm < 5, "controlnet: unsupported conditioning embedder layout");
                TRY(ps.conv(names[i], cout[i], cin[i], 3, false, &w->ce_gemm[w->n_gemm++]));
            }
            w->ce_chain_c.push_back(cout[i]);
            w->ce_chain_stride.push_back(stride[i]);
        }
    }

    std::vector<std::string> temb_names;
    std::vector<int> temb_ns;
    // per-conv selection of the split operands; a checkpoint whose normalisation scales have outlier channels keeps them on every level
    // the 1x1 convolutions take them on (ParamSink::norm_scale_spread)
    const int levels = cn_split_levels();
    // (CTRL_CN_SPLIT_RESNET_LEVELS < levels -- the default -- is not applied to a checkpoint with outlier norm scales)
    const float spread = ps.norm_scale_spread("");
    const bool outliers = spread > kNormSpreadGate;
    const int res_levels = outliers ? levels : cn_split_resnet_levels();
    w->sel_levels = dup ? levels : 0; w->sel_res_levels = dup ? res_levels : 0; w->sel_norm_spread = spread; w->sel_outliers = outliers;
    auto add_resnet = [&](const std::string& pre, int Cin, int Cout, ResnetW* r, bool dup_r, int dup_sc = -1) -> int {
        TRY(build_resnet(ps, pre, Cin, Cout, false, r, dup_r, dup_sc));
        r->temb_off = w->temb_total;
        w->temb_total += Cout;
        temb_names.push_back(pre + ".time_emb_proj");
        temb_ns.push_back(Cout);
        return 0;
    };

    int out_c = c0;
    for (int i = 0; i < 4; ++i) {
        DownBlockW& d = w->down[i];
        d.Cin = out_c; d.Cout = c.block_out_channels[i]; out_c = d.Cout;
        d.has_attn = c.down_block_has_attn[i] != 0;
        d.has_down = (i != 3);
        const std::string pre = "down_blocks." + std::to_string(i);
        d.resnets.resize(c.layers_per_block);
        if (d.has_attn) { d.tnorm.resize(c.layers_per_block); d.proj_in.resize(c.layers_per_block);
                          d.proj_out.resize(c.layers_per_block); d.tb.resize(c.layers_per_block); }
        const bool dup_i = dup && i < levels;
        for (int j = 0; j < c.layers_per_block; ++j) {
            TRY(add_resnet(pre + ".resnets." + std::to_string(j), j == 0 ? d.Cin : d.Cout, d.Cout, &d.resnets[j], dup && i < res_levels, dup_i ? 1 : 0));
            if (d.has_attn)
                TRY(build_transformer2d(ps, pre + ".attentions." + std::to_string(j), d.Cout, c.num_attention_heads,
                                        c.cross_attention_dim, &d.tnorm[j], &d.proj_in[j], &d.proj_out[j], &d.tb[j], dup_i));
        }
        if (d.has_down) TRY(ps.conv(pre + ".downsamplers.0.conv", d.Cout, d.Cout, 3, false, &d.down, dup_i));
    }
    const bool dup_mid = dup && levels > 4;
    TRY(add_resnet("mid_block.resnets.0", out_c, out_c, &w->mid_r0, dup_mid));
    TRY(build_transformer2d(ps, "mid_block.attentions.0", out_c, c.num_attention_heads, c.cross_attention_dim,
                            &w->mid_tnorm, &w->mid_pin, &w->mid_pout, &w->mid_tb, dup_mid));
    TRY(add_resnet("mid_block.resnets.1", out_c, out_c, &w->mid_r1, dup_mid));
    TRY(ps.linear_cat(temb_names, temb_ns, temb_dim, true, &w->temb_cat));

    // zero convs: slot channels follow the residual list (controlnet/controlnet.py:360-408)
    std::vector<int> slot_c = {c0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < c.layers_per_block + (i != 3 ? 1 : 0); ++j) slot_c.push_back(c.block_out_channels[i]);
    w->zero_convs.resize(slot_c.size() + 1);
    for (size_t i = 0; i < slot_c.size(); ++i)
        TRY(ps.conv("controlnet_down_blocks." + std::to_string(i), slot_c[i], slot_c[i], 1, false, &w->zero_convs[i], dup));
    TRY(ps.conv("controlnet_mid_block", out_c, out_c, 1, false, &w->zero_convs[slot_c.size()], dup));
    return 0;
}

}  // namespace

struct ctrl_controlnet : PlanBase {
    ControlNetW w;
    std::shared_ptr<Packer> packer;      // owns the packed weights; shared with the plan's clones (ctrl_controlnet_clone)
    Arena arena;
    KvCache kvc;                         // text K/V cache (ctrl_*_text_cache)
    // fused step (ctrl_step_forward): the network runs on its own stream and signals every output with an event
    hipStream_t side = nullptr;
    hipEvent_t fork_ev = nullptr, done_ev = nullptr, out_ev[13] = {};
    // auxiliary lane: work off the critical chain of the network -- the text K/V projections of all transformer blocks
    // (they depend on encoder_hidden_states only) and the 13 zero-convs (each needs only its residual) -- runs on this
    // stream, forked / joined with events (hipGraph-capturable), in the CUs the small-M chain leaves idle
    hipStream_t aux = nullptr;
    hipEvent_t aux_fork = nullptr, aux_kv = nullptr, aux_done = nullptr, res_ev[13] = {};
    // step-invariant cache of the conditioning embedder's last hidden map (CTRL_COND_KEEP / CTRL_COND_REUSE)
    half_t* cond_cache = nullptr;
    std::vector<void*> cond_retired;    // outgrown caches stay alive (captured graphs may still address them)
    size_t cond_cache_elems = 0;        // capacity
    int cond_N = 0, cond_H = 0, cond_W = 0;   // what the cache holds (0 = nothing)
    int init_async() {
        HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&done_ev, hipEventDisableTiming));
        for (int i = 0; i < 13; ++i) HIP_TRY(hipEventCreateWithFlags(&out_ev[i], hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&aux_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&aux_kv, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&aux_done, hipEventDisableTiming));
        for (int i = 0; i < 13; ++i) HIP_TRY(hipEventCreateWithFlags(&res_ev[i], hipEventDisableTiming));
        return 0;
    }
    ~ctrl_controlnet() {
        if (packer && packer.use_count() == 1) packer->release_all();      // (the last plan over these weights frees them: ctrl_controlnet_clone)
        if (cond_cache) (void)hipFree(cond_cache);
        for (void* p : cond_retired) (void)hipFree(p);
        if (side) (void)hipStreamDestroy(side);
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        if (done_ev) (void)hipEventDestroy(done_ev);
        for (int i = 0; i < 13; ++i) if (out_ev[i]) (void)hipEventDestroy(out_ev[i]);
        if (aux) (void)hipStreamDestroy(aux);
        if (aux_fork) (void)hipEventDestroy(aux_fork);
        if (aux_kv) (void)hipEventDestroy(aux_kv);
        if (aux_done) (void)hipEventDestroy(aux_done);
        for (int i = 0; i < 13; ++i) if (res_ev[i]) (void)hipEventDestroy(res_ev[i]);
    }
};

namespace {

struct FwdArgs {
    const void* sample; int sample_dt; int N, Hs, Ws;
    const float* t; int t_count;
    const void* ehs; int ehs_dt; int Lk;
    const void* cond; int cond_dt;
    float scale; int flags;
    void* const* outs; int out_dt;
    hipEvent_t* out_ev;     // optional [13]: recorded on the launch stream right after output i has been enqueued
    half_t* cond_cache;     // plan-owned [N][H][W][c_last_hidden] or null
    bool cond_reuse;        // start the conditioning embedder from cond_cache
    ctrl_controlnet* plan;  // auxiliary lane (streams / events); aux_on false: everything on the launch stream
    bool aux_on;
};

int run_transformer2d(Ctx& cx, const Norm& tn, const ConvW& pin, const ConvW& pout, const BasicTBW& tb, const TV& x,
                      const TV& out, int N, int H, int W, const EhsCtx& e, const PreKV* kv) {
    const size_t mk = cx.mark();
    const int C = tn.C, M = N * H * W;
    half_t* n = cx.h((size_t)M * pin.Cin);
    TRY(run_groupnorm(cx, tn, x, n, N, H * W, 1e-6f, false, pin.dup));
    TV t0 = stream_alloc(cx, (size_t)M * C, false);
    ConvOpts o;
    TRY(run_conv(cx, pin, n, t0, N, H, W, o));
    // its fp16 mirror is proj_out's operand: [hi | lo] rows only when that conv takes split operands
    TV t1 = pout.dup ? stream_alloc_rc(cx, (size_t)M, C, true) : stream_alloc(cx, (size_t)M * C, true);
    TRY(run_basic_tb(cx, tb, t0, t1, N, H * W, e, nullptr, kv));
    ConvOpts oo; oo.res = x;
    TRY(run_conv(cx, pout, t1.m16, out, N, H, W, oo));
    cx.release(mk);
    return 0;
}

int controlnet_run(Ctx& cx, const ControlNetW& w, const FwdArgs& a) {
    const ctrl_controlnet_config& c = w.cfg;
    const int N = a.N, c0 = c.block_out_channels[0], temb_dim = 4 * c0;
    TRY(begin_forward(cx));
    // ---- 1. time embedding (:735-758) and every resnet's projection of it, one batched small-M linear ----
    float* tsin = cx.f((size_t)N * c0);
    RUN(cx, op_timestep_sincos(a.t, a.t_count, tsin, N, c0, cx.s));
    float* t1 = cx.f((size_t)N * temb_dim);
    RUN(cx, op_linear_small(tsin, c0, w.te1.w, w.te1.b, t1, temb_dim, N, temb_dim, c0, 0, 1, cx.s));
    float* emb = cx.f((size_t)N * temb_dim);
    if (a.flags & CTRL_SKIP_TIME_EMB) RUN(cx, op_fill_zero(emb, (size_t)N * temb_dim * sizeof(float), cx.s));   // :809-811
    // (emb leaves linear_2 as SiLU(emb): its only consumer is the concatenated time_emb_proj of the resnets, which all start with SiLU(emb)
    //  -- applied once per element here instead of once per element and OUTPUT COLUMN there; SiLU(0) = 0 keeps the skip_time_emb zeros)
    else RUN(cx, op_linear_small(t1, temb_dim, w.te2.w, w.te2.b, emb, temb_dim, N, temb_dim, temb_dim, 0, 1, cx.s));
    float* tproj = cx.f((size_t)N * w.temb_total);
    RUN(cx, op_linear_small(emb, temb_dim, w.temb_cat.w, w.temb_cat.b, tproj, w.temb_total, N, w.temb_total, temb_dim, 0, 0, cx.s));

    // ---- encoder hidden states -> fp16 [N*Lk][cross] ----
    EhsCtx e;
    e.batch = N; e.Lk = a.Lk; e.cross = c.cross_attention_dim;
    {
        half_t* e16 = cx.h((size_t)N * a.Lk * e.cross);
        RUN(cx, op_nchw_to_nhwc(a.ehs, a.ehs_dt, e16, 1, 1, N * a.Lk * e.cross, cx.s));
        e.h16 = e16;
    }

    // ---- text K / V^T projections of every transformer block, ahead of time on the auxiliary lane ----
    hipStream_t const main_s = cx.s;
    const bool aux = a.aux_on && !cx.dry;
    std::vector<const BasicTBW*> tbs;
    for (int i = 0; i < 4; ++i)
        if (w.down[i].has_attn) for (const BasicTBW& t : w.down[i].tb) tbs.push_back(&t);
    tbs.push_back(&w.mid_tb);
    std::vector<PreKV> pre_kv(tbs.size());
    {
        if (aux) {
            HIP_TRY(hipEventRecord(a.plan->aux_fork, main_s));
            HIP_TRY(hipStreamWaitEvent(a.plan->aux, a.plan->aux_fork, 0));
            cx.s = a.plan->aux;
        }
        // (the blocks of one level project the same text states with equal shapes: recorded, then replayed in lock-step so that
        //  they leave as grouped launches -- ops.h: OpCollector; 7 launches -> 3)
        const bool grp_kv = group_launches_enabled() && !cx.dry && tbs.size() <= 16;
        std::vector<OpList> kv_ops(grp_kv ? tbs.size() : 0);
        for (size_t i = 0; i < tbs.size(); ++i) {
            cx.rec = grp_kv ? &kv_ops[i] : nullptr;
            const int rc = project_text_kv(cx, tbs[i]->attn2, e, &pre_kv[i]);
            cx.rec = nullptr;
            if (rc) return rc;
        }
        for (size_t i0 = 0; grp_kv && i0 < tbs.size();) {
            size_t i1 = i0 + 1;
            while (i1 < tbs.size() && i1 - i0 < (size_t)kMaxGroup && tbs[i1]->attn2.inner == tbs[i0]->attn2.inner) ++i1;
            TRY(replay_lockstep(&kv_ops[i0], (int)(i1 - i0)));
            i0 = i1;
        }
        if (aux) {
            HIP_TRY(hipEventRecord(a.plan->aux_kv, a.plan->aux));
            cx.s = main_s;
        }
    }
    bool kv_waited = !aux;
    size_t tb_next = 0;
    auto next_kv = [&]() -> const PreKV* {           // the main chain meets the projections at its first transformer block
        if (!kv_waited) { (void)hipStreamWaitEvent(main_s, a.plan->aux_kv, 0); kv_waited = true; }
        return &pre_kv[tb_next++];
    };

    // ---- 2. stem: conv_in(sample) (:802-807) ----
    const int H = a.Hs, W = a.Ws;
    TV x = stream_alloc_rc(cx, (size_t)N * H * W, c0, true);   // block input (also residual slot 0)
    half_t* stem = nullptr;
    if (!(a.flags & CTRL_SKIP_CONV_IN)) {
        stem = cx.h((size_t)N * H * W * c0);
        RUN(cx, op_conv3x3_direct(a.sample, a.sample_dt, 1, w.conv_in.w, w.conv_in.b, stem, N, c.in_channels, c0, H, W, 1, 0, cx.s));
    }
    // ---- conditioning embedding (:94-104) + add (:816-817) ----
    {
        const size_t mk = cx.mark();
        int ch = c.conditioning_channels, hh = 8 * H, ww = 8 * W;
        const void* cur = a.cond; int cur_dt = a.cond_dt; int nchw = 1;
        const size_t nl = w.ce_chain_c.size();
        int gi = 0;
        for (size_t i = 0; i < nl; ++i) {
            const int co = w.ce_chain_c[i], st = w.ce_chain_stride[i];
            const int ho = (hh - 1) / st + 1, wo = (ww - 1) / st + 1;
            const bool last = (i + 1 == nl);
            // the layer in front of the last one writes the step-invariant map: into the plan's cache when there is one
            const bool cached_layer = (i + 2 == nl) && a.cond_cache;
            half_t* y16 = last ? nullptr : (cached_layer ? a.cond_cache : cx.h((size_t)N * ho * wo * co));
            if (a.cond_reuse && !last) {
                // unchanged condition image: skip everything up to the cached map
                if ((int)i >= w.n_direct) ++gi;
            } else if ((int)i < w.n_direct) {
                CTRL_CHECK(!last, "controlnet: the conditioning embedder must end with an implicit-GEMM layer");
                // channels-last 16 / 32-channel layers: on the matrix cores (round 5; CTRL_SMALLCONV_MFMA=0: the direct VALU kernel)
                const bool small_mfma = !policy_is0(P_SMALLCONV_MFMA);
                if (small_mfma && !nchw && cur_dt == DT_F16 && w.ce_direct[i].w16)
                    RUN(cx, op_conv3x3_small_mfma((const half_t*)cur, w.ce_direct[i].w16, w.ce_direct[i].b, y16, N, ch, co, hh, ww, st, 1, cx.s));
                else
                    RUN(cx, op_conv3x3_direct(cur, cur_dt, nchw, w.ce_direct[i].w, w.ce_direct[i].b, y16, N, ch, co, hh, ww, st, 1, cx.s));
            } else {
                ConvOpts o; o.stride = st; o.act = last ? 0 : 1;
                if (last && stem) o.res = tv16(stem);
                TRY(run_conv(cx, w.ce_gemm[gi++], (const half_t*)cur, last ? x : tv16(y16), N, hh, ww, o));
            }
            cur = y16; cur_dt = DT_F16; nchw = 0; ch = co; hh = ho; ww = wo;
        }
        CTRL_CHECK(hh == H && ww == W, "controlnet_cond must be 8x the latent resolution");
        cx.release(mk);
    }

    // ---- 5./6. zero convs (:850-858) with the conditioning scale fused (:861-868), NCHW outputs.  Each one is enqueued
    //      as soon as its residual exists (the reference applies them after the mid block; they only read the residual),
    //      so a consumer of output i -- the adapter block of slot i in the fused step -- can start while the rest of the
    //      network is still running ----
    const size_t nout = w.zero_convs.size();
    size_t n_emitted = 0;
    // Grouped launches (round 5): consecutive residuals of one shape -- r0..r2, r4..r5, r7..r8, r9..r11 + mid -- are held back and
    // their zero-convs leave as ONE launch when the shape changes (13 launches -> 6).  Not in the fused step (a.out_ev): there every
    // output is wanted as early as possible (an adapter block starts when ITS input exists).
    const bool grp_zc = group_launches_enabled() && !a.out_ev;
    IGemmArgs zc_pend[kMaxGroup];
    size_t zc_first = 0;
    int zc_n = 0;
    auto zc_flush = [&]() -> int {
        if (zc_n == 0) return 0;
        hipStream_t zs = cx.s;
        if (aux) {      // the group only reads residuals that exist since its last member was emitted (event recorded there): off the
                        // chain, onto the auxiliary lane
            HIP_TRY(hipStreamWaitEvent(a.plan->aux, a.plan->res_ev[zc_first + zc_n - 1], 0));
            zs = a.plan->aux;
        }
        const int n = zc_n;
        zc_n = 0;
        RUN(cx, op_igemm_group(zc_pend, n, zs));
        return 0;
    };
    auto emit = [&](const TV& r, int hh, int ww) -> int {
        const size_t i = n_emitted++;
        CTRL_CHECK(i < nout, "controlnet: residual/zero-conv count mismatch");
        float sc = a.scale;
        if (a.flags & CTRL_GUESS_MODE) sc *= powf(10.f, -1.f + (float)i / (float)(nout - 1));   // torch.logspace(-1, 0, n) * scale
        const ConvW& z = w.zero_convs[i];
        const int HW = hh * ww;
        IGemmArgs g = {};
        CTRL_CHECK(cx.dry || (r.lo_off > 0) == z.dup, "controlnet: split-operand zero conv and its input mirror disagree");
        g.A = r.m16; g.lda = z.Cin; g.mode = IG_ROWS; g.Cin = z.Cin; g.taps = 1; g.a_split = z.paired ? 2 : (z.dup ? 1 : 0);
        g.W = z.w; g.M = N * HW; g.Nout = z.Cout; g.Ktot = z.Cin; g.bias = z.b; g.scale = sc;
        g.nseg = 1;
        g.seg[0] = IGemmSeg{a.outs[i], HW, 0, z.Cout, SEG_TRANSPOSED, a.out_dt, HW, 0};
        if (grp_zc) {
            if (zc_n > 0 && (zc_n == kMaxGroup || zc_pend[0].M != g.M || zc_pend[0].Nout != g.Nout || zc_pend[0].Ktot != g.Ktot)) TRY(zc_flush());
            if (zc_n == 0) zc_first = i;
            zc_pend[zc_n++] = g;
            if (aux) HIP_TRY(hipEventRecord(a.plan->res_ev[i], main_s));      // residual i exists from here on
            return 0;
        }
        hipStream_t zs = cx.s;
        if (aux) {      // the zero-conv only reads the residual that exists now: off the chain, onto the auxiliary lane
            HIP_TRY(hipEventRecord(a.plan->res_ev[i], main_s));
            HIP_TRY(hipStreamWaitEvent(a.plan->aux, a.plan->res_ev[i], 0));
            zs = a.plan->aux;
        }
        RUN(cx, op_igemm(g, zs));
        if (!cx.dry && a.out_ev) HIP_TRY(hipEventRecord(a.out_ev[i], zs));
        return 0;
    };
    // ---- 3. down blocks (:820-833) ----
    TRY(emit(x, H, W));
    int h = H, wd = W;
    TV cur = x;
    for (int i = 0; i < 4; ++i) {
        const DownBlockW& d = w.down[i];
        for (size_t j = 0; j < d.resnets.size(); ++j) {
            TV r = stream_alloc_rc(cx, (size_t)N * h * wd, d.Cout, true);
            TRY(run_resnet(cx, d.resnets[j], cur, r, N, h, wd, 1, tproj + d.resnets[j].temb_off, w.temb_total, c.norm_eps));
            cur = r;
            if (d.has_attn) {
                TV t = stream_alloc_rc(cx, (size_t)N * h * wd, d.Cout, true);
                TRY(run_transformer2d(cx, d.tnorm[j], d.proj_in[j], d.proj_out[j], d.tb[j], cur, t, N, h, wd, e, next_kv()));
                cur = t;
            }
            TRY(emit(cur, h, wd));
        }
        if (d.has_down) {
            const int ho = (h - 1) / 2 + 1, wo = (wd - 1) / 2 + 1;
            TV y = stream_alloc_rc(cx, (size_t)N * ho * wo, d.Cout, true);
            ConvOpts o; o.stride = 2;
            if (cur.lo_off > 0 && !d.down.dup) o.lda = 2 * d.Cout;      // plain conv on a split mirror: the hi half of every row
            TRY(run_conv(cx, d.down, cur.m16, y, N, h, wd, o));
            cur = y; h = ho; wd = wo;
            TRY(emit(cur, h, wd));
        }
    }
    // ---- 4. mid block (:836-846) ----
    {
        const int C = c.block_out_channels[3];
        TV m0 = stream_alloc_rc(cx, (size_t)N * h * wd, C, true);
        TRY(run_resnet(cx, w.mid_r0, cur, m0, N, h, wd, 1, tproj + w.mid_r0.temb_off, w.temb_total, c.norm_eps));
        TV m1 = stream_alloc_rc(cx, (size_t)N * h * wd, C, true);
        TRY(run_transformer2d(cx, w.mid_tnorm, w.mid_pin, w.mid_pout, w.mid_tb, m0, m1, N, h, wd, e, next_kv()));
        TV m2 = stream_alloc_rc(cx, (size_t)N * h * wd, C, true);
        TRY(run_resnet(cx, w.mid_r1, m1, m2, N, h, wd, 1, tproj + w.mid_r1.temb_off, w.temb_total, c.norm_eps));
        TRY(emit(m2, h, wd));
    }
    TRY(zc_flush());
    CTRL_CHECK(n_emitted == nout, "controlnet: residual/zero-conv count mismatch");
    if (aux) {          // join: the launch stream does not run past the forward before the auxiliary lane has finished
        HIP_TRY(hipEventRecord(a.plan->aux_done, a.plan->aux));
        HIP_TRY(hipStreamWaitEvent(main_s, a.plan->aux_done, 0));
    }
    return 0;
}

}  // namespace

extern "C" {

int ctrl_controlnet_param_count(const ctrl_controlnet_config* cfg) {
    if (!cfg) return -1;
    SpecCollector sc; ControlNetW w;
    if (build_controlnet(sc, *cfg, &w)) return -1;
    return (int)sc.entries.size();
}

int ctrl_controlnet_param_spec(const ctrl_controlnet_config* cfg, int i, char* name, int name_len, int64_t shape[6], int* ndim) {
    CTRL_CHECK(cfg && name && shape && ndim, "param_spec: null argument");
    SpecCollector sc; ControlNetW w;
    TRY(build_controlnet(sc, *cfg, &w));
    CTRL_CHECK(i >= 0 && i < (int)sc.entries.size(), "param_spec: index out of range");
    std::strncpy(name, sc.entries[i].name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
    *ndim = (int)sc.entries[i].shape.size();
    for (int k = 0; k < 6; ++k) shape[k] = k < *ndim ? sc.entries[i].shape[k] : 1;
    return 0;
}

int ctrl_controlnet_create(const ctrl_controlnet_config* cfg, const ctrl_tensor_ref* tensors, int n_tensors, void* stream,
                           ctrl_controlnet** out) {
    CTRL_CHECK(cfg && tensors && out, "controlnet_create: null argument");
    std::unique_ptr<ctrl_controlnet> h(new ctrl_controlnet());
    TRY(h->init_base(n_tensors > 0 ? tensors[0].data : nullptr));
    DeviceGuard dg(h->device);           // the plan lives on the parameters' device, whatever the current device is
    h->packer.reset(new Packer(tensors, n_tensors, (hipStream_t)stream));
    h->packer->split_paired = cn_split_paired();
    int rc = build_controlnet(*h->packer, *cfg, &h->w);
    if (rc) return rc;     // ~ctrl_controlnet frees what was packed so far
    TRY(h->init_async());
    TRY(h->packer->finish());            // sync: packed copies are complete (source tensors may be freed); fp16 range check
    *out = h.release();
    return 0;
}

// A second plan over the SAME packed weights (shared, reference-counted) with a workspace, streams and events of its own: two forwards
// of the module can then be in flight at once -- the Python mirror runs the two halves of a batch on two stream lanes through it
// (ControlNetModel.forward, round 6).  Destroy like any plan; the weights go with the last plan that holds them.
int ctrl_controlnet_clone(ctrl_controlnet* h, ctrl_controlnet** out) {
    CTRL_CHECK(h && out, "controlnet_clone: null argument");
    std::unique_ptr<ctrl_controlnet> c(new ctrl_controlnet());
    c->device = h->device;
    DeviceGuard dg(h->device);
    c->w = h->w;
    c->packer = h->packer;
    TRY(c->init_async());
    *out = c.release();
    return 0;
}

void ctrl_controlnet_destroy(ctrl_controlnet* h) { delete h; }

int ctrl_controlnet_trim(ctrl_controlnet* h) {
    CTRL_CHECK(h, "controlnet_trim: null plan");
    DeviceGuard dg(h->device);
    CTRL_CHECK(!h->capture_active(), "controlnet_trim: a stream capture of this plan's forward is in progress");      // see ctrl_adapter_trim
    HIP_TRY(hipDeviceSynchronize());             // nothing queued still touches a retired block
    h->arena.trim();
    h->kvc.trim();
    for (void* p : h->cond_retired) (void)hipFree(p);
    h->cond_retired.clear();
    return 0;
}

int ctrl_controlnet_selection(ctrl_controlnet* h, char* buf, int len) {
    CTRL_CHECK(h && buf && len > 0, "controlnet_selection: null argument");
    snprintf(buf, (size_t)len, "split_operands=%d split_levels=%d split_resnet_levels=%d norm_scale_spread=%.2f gate=%.1f selection=%s", h->w.split ? 1 : 0,
             h->w.sel_levels, h->w.sel_res_levels, h->w.sel_norm_spread, kNormSpreadGate,
             h->w.sel_outliers ? "conservative(outlier norm scales)" : "default");
    return 0;
}

int ctrl_controlnet_text_cache(ctrl_controlnet* h, int mode) {
    CTRL_CHECK(h && mode >= 0 && mode <= 2, "controlnet_text_cache: mode must be 0 (off), 1 (keep) or 2 (reuse)");
    h->kvc.mode = mode;
    return 0;
}

static int controlnet_forward_impl(ctrl_controlnet* h, const void* sample, int sample_dtype, int N, int Hs, int Ws,
                                   const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype, int Lk,
                                   const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                                   void* const* outs, int out_dtype, hipStream_t s, hipEvent_t* out_ev) {
    CTRL_CHECK(h && sample && timesteps && encoder_hidden_states && controlnet_cond && outs, "controlnet_forward: null argument");
    DeviceGuard dg(h->device);
    bool capturing = false;
    TRY(h->enter(s, &capturing));
    // a KEEP forward that still has to allocate its text K/V buffers cannot be recorded into a hipGraph: refuse up front,
    // before any lane is forked (an error in the middle of a capture leaves unjoined streams behind)
    CTRL_CHECK(!(capturing && h->kvc.mode == KvCache::KEEP && h->kvc.slots.empty()),
               "text K/V cache: the first KEEP forward allocates its buffers and cannot run under stream capture -- run it "
               "eagerly once, then capture");
    CTRL_CHECK(N >= 1 && N <= 4096 && Hs >= 1 && Ws >= 1 && Lk >= 1, "controlnet_forward: bad sizes");
    CTRL_CHECK(Hs % 8 == 0 && Ws % 8 == 0, "controlnet_forward: latent height/width must be multiples of 8 (3 stride-2 stages)");
    CTRL_CHECK(t_count == 1 || t_count == N, "controlnet_forward: need 1 or N timesteps");
    for (int i = 0; i < 13; ++i) CTRL_CHECK(outs[i] != nullptr, "controlnet_forward: null output pointer");
    // step-invariant conditioning-embedder cache
    const bool keep = (flags & CTRL_COND_KEEP) != 0, reuse = (flags & CTRL_COND_REUSE) != 0;
    half_t* cache = nullptr;
    if (keep || reuse) {
        const size_t nl = h->w.ce_chain_c.size();
        CTRL_CHECK(nl >= 2, "controlnet: conditioning embedder too short to cache");
        const size_t need = (size_t)N * Hs * Ws * h->w.ce_chain_c[nl - 2];
        if (reuse) {
            CTRL_CHECK(h->cond_cache && h->cond_N == N && h->cond_H == Hs && h->cond_W == Ws,
                       "controlnet_forward: CTRL_COND_REUSE without a matching CTRL_COND_KEEP forward");
        } else if (need > h->cond_cache_elems) {
            if (h->cond_cache) h->cond_retired.push_back(h->cond_cache);
            h->cond_cache = nullptr; h->cond_cache_elems = 0; h->cond_N = 0;
            HIP_TRY(hipMalloc((void**)&h->cond_cache, need * sizeof(half_t)));
            h->cond_cache_elems = need;
        }
        cache = h->cond_cache;
    }
    const bool aux_env = policy_int(P_CN_AUX, 1) != 0;
    FwdArgs a = {sample, sample_dtype, N, Hs, Ws, timesteps, t_count, encoder_hidden_states, ehs_dtype, Lk,
                 controlnet_cond, cond_dtype, conditioning_scale, flags, outs, out_dtype, out_ev, cache, reuse,
                 h, aux_env && !g_prof_on && out_ev == nullptr && !(flags & CTRL_NO_AUX_LANE)};
    // (off while the per-launch profiler records -- one kernel at a time -- and in the fused step, whose ControlNet already
    //  overlaps with the adapter: capturing fused + auxiliary lane into a hipGraph crashed hipGraphInstantiate on ROCm 7.2)
    // sizing pass (no launches) -> workspace; then the real pass
    h->arena.off = 0; h->arena.peak = 0;
    Ctx dry{&h->arena, s, true};
    dry.f32stream = stream_f32_enabled();
    dry.split = h->w.split;
    dry.kvc = &h->kvc; h->kvc.next = 0;
    if (h->kvc.mode == KvCache::REUSE)
        CTRL_CHECK(h->kvc.key_batch == N && h->kvc.key_Lk == Lk, "controlnet_forward: text K/V cache was kept for another batch / prompt length");
    CTRL_CHECK(!dry.split || dry.f32stream, "controlnet_forward: the plan was built with split-operand convolutions "
               "(CTRL_CN_SPLIT) and needs fp32 residual streams (CTRL_STREAM_F32 was switched off after create)");
    TRY(controlnet_run(dry, h->w, a));
    TRY(h->arena.ensure(workspace_bytes(dry)));
    h->arena.off = 0;
    Ctx cx{&h->arena, s, false};
    cx.f32stream = dry.f32stream;
    cx.split = dry.split;
    cx.kvc = &h->kvc; h->kvc.next = 0;
    cx.capturing = capturing;
    cx.stats_total = dry.stats_total;
    TRY(controlnet_run(cx, h->w, a));
    if (h->kvc.mode == KvCache::KEEP) { h->kvc.key_batch = N; h->kvc.key_Lk = Lk; }
    if (keep && !reuse) { h->cond_N = N; h->cond_H = Hs; h->cond_W = Ws; }
    return h->leave(s, capturing);
}

int ctrl_controlnet_forward(ctrl_controlnet* h, const void* sample, int sample_dtype, int N, int Hs, int Ws,
                            const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype, int Lk,
                            const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                            void* const* outs, int out_dtype, void* stream) {
    return controlnet_forward_impl(h, sample, sample_dtype, N, Hs, Ws, timesteps, t_count, encoder_hidden_states, ehs_dtype, Lk,
                                   controlnet_cond, cond_dtype, conditioning_scale, flags, outs, out_dtype, (hipStream_t)stream, nullptr);
}

}  // extern "C"

// Fused-step half (plan_fused.cpp): the whole network on the plan's own stream, forked from `main_s`; *out_events gets the
// 13 per-output events and *done the event that marks the end of the network.  With `async` false (profiler recording)
// everything runs on main_s and no events are produced.
int controlnet_forward_async(ctrl_controlnet* h, const void* sample, int sample_dtype, int N, int Hs, int Ws,
                             const float* timesteps, int t_count, const void* encoder_hidden_states, int ehs_dtype, int Lk,
                             const void* controlnet_cond, int cond_dtype, float conditioning_scale, int flags,
                             void* const* outs, int out_dtype, hipStream_t main_s, bool async,
                             hipEvent_t** out_events, hipEvent_t* done) {
    CTRL_CHECK(h, "controlnet_forward: null plan");
    if (!async) {
        *out_events = nullptr; *done = nullptr;
        return controlnet_forward_impl(h, sample, sample_dtype, N, Hs, Ws, timesteps, t_count, encoder_hidden_states, ehs_dtype, Lk,
                                       controlnet_cond, cond_dtype, conditioning_scale, flags, outs, out_dtype, main_s, nullptr);
    }
    HIP_TRY(hipEventRecord(h->fork_ev, main_s));
    HIP_TRY(hipStreamWaitEvent(h->side, h->fork_ev, 0));
    TRY(controlnet_forward_impl(h, sample, sample_dtype, N, Hs, Ws, timesteps, t_count, encoder_hidden_states, ehs_dtype, Lk,
                                controlnet_cond, cond_dtype, conditioning_scale, flags, outs, out_dtype, h->side, h->out_ev));
    HIP_TRY(hipEventRecord(h->done_ev, h->side));
    *out_events = h->out_ev; *done = h->done_ev;
    return 0;
}
