// Host-side plan infrastructure shared by the ControlNet and Ctrl-Adapter orchestrators (C++, no torch):
//   * ParamSink: one description of the module tree drives both the parameter inventory
//     (ctrl_*_param_spec: names/shapes = the reference's state-dict keys) and the weight packer
//   * Arena: stack-discipline device workspace owned by the plan (sized by a dry pass, so the steady-state
//     forward performs no hipMalloc and is graph-capturable)
//   * block runners (GroupNorm+SiLU, ResnetBlock2D, BasicTransformerBlock, ...) that enqueue the HIP kernels
#pragma once
#include "ops.h"
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------------------------------ packed params
// wperm: ff.net.2 only -- the weights once more in the layout / k order of the fused feed-forward kernel (ffn.hip; dim 512 blocks)
struct Lin { half_t* w = nullptr; float* b = nullptr; int N = 0, K = 0; bool geglu = false; half_t* wperm = nullptr; };
// dup: split-operand convolution -- Cin is the K per tap the kernel walks (2 x the logical width: [hi | lo] halves of the
// operand rows against the weights repeated twice), see ctrl_igemm_desc::a_split
// paired: the split operand is walked in (hi, lo) pairs against the PLAIN weight pack (ctrl_igemm_desc::a_split == 2)
struct ConvW { half_t* w = nullptr; float* b = nullptr; int Cout = 0, Cin = 0, taps = 0; bool dup = false, paired = false; };
// w16: fp16 [Cout][9][Cin] copy for the MFMA form of the small convolutions (op_conv3x3_small_mfma), when the layer qualifies
struct ConvD { float* w = nullptr; float* b = nullptr; int Cout = 0, Cin = 0; half_t* w16 = nullptr; };
struct Norm { float* g = nullptr; float* b = nullptr; int C = 0; };

struct SpecEntry { std::string name; std::vector<int64_t> shape; };

struct ParamSink {
    virtual ~ParamSink() {}
    // nn.Linear(K, N): <name>.weight [N][K] (+ <name>.bias [N])
    virtual int linear(const std::string& name, int N, int K, bool bias, bool geglu, Lin* out) = 0;
    // several bias-free / biased linears with the same K concatenated along N (QKV, batched time projections)
    virtual int linear_cat(const std::vector<std::string>& names, const std::vector<int>& Ns, int K, bool bias, Lin* out) = 0;
    // nn.Conv2d(Cin, Cout, k) for the implicit GEMM: k = 1 | 3 ; nn.Conv3d (3,1,1) when temporal
    virtual int conv(const std::string& name, int Cout, int Cin, int k, bool temporal, ConvW* out, bool dup = false) = 0;
    // nn.Conv2d(Cin, Cout, 3) for the direct small-channel kernel
    virtual int conv_direct(const std::string& name, int Cout, int Cin, ConvD* out) = 0;
    virtual int norm(const std::string& name, int C, Norm* out) = 0;
    virtual int scalar(const std::string& name, float** out) = 0;
    // raw fp32 copy of a small tensor (router weights ...)
    virtual int raw_f32(const std::string& name, const std::vector<int64_t>& shape, float** out) = 0;
    // second pack of a feed-forward's output projection for the fused GEGLU kernel (op_ffn_pack_w2), where its shape qualifies
    virtual int ffn_perm(Lin* ff2) { (void)ff2; return 0; }
    // spread of the normalisation scales under `prefix`: max over its GroupNorm / LayerNorm weights of max|gamma| / median|gamma|
    // (0 = unknown: the parameter inventory pass).  Outlier channels in a trained checkpoint's norm scales are what the precision
    // selections of the path (fp16 adapter token stream, plain operands for the ControlNet's low-resolution 3x3 convolutions) are
    // least robust to (tests/test_gpu_e2e.py::test_weight_distribution_sweep_sdxl_chain): the plans fall back to the conservative
    // selection where the spread exceeds kNormSpreadGate.
    virtual float norm_scale_spread(const std::string& prefix) { (void)prefix; return 0.f; }
};
constexpr float kNormSpreadGate = 4.0f;

struct SpecCollector : ParamSink {
    std::vector<SpecEntry> entries;
    int linear(const std::string& name, int N, int K, bool bias, bool, Lin*) override {
        entries.push_back({name + ".weight", {N, K}});
        if (bias) entries.push_back({name + ".bias", {N}});
        return 0;
    }
    int linear_cat(const std::vector<std::string>& names, const std::vector<int>& Ns, int K, bool bias, Lin*) override {
        for (size_t i = 0; i < names.size(); ++i) linear(names[i], Ns[i], K, bias, false, nullptr);
        return 0;
    }
    int conv(const std::string& name, int Cout, int Cin, int k, bool temporal, ConvW*, bool) override {
        if (temporal) entries.push_back({name + ".weight", {Cout, Cin, 3, 1, 1}});
        else entries.push_back({name + ".weight", {Cout, Cin, k, k}});
        entries.push_back({name + ".bias", {Cout}});
        return 0;
    }
    int conv_direct(const std::string& name, int Cout, int Cin, ConvD*) override {
        entries.push_back({name + ".weight", {Cout, Cin, 3, 3}});
        entries.push_back({name + ".bias", {Cout}});
        return 0;
    }
    int norm(const std::string& name, int C, Norm*) override {
        entries.push_back({name + ".weight", {C}});
        entries.push_back({name + ".bias", {C}});
        return 0;
    }
    int scalar(const std::string& name, float**) override { entries.push_back({name, {1}}); return 0; }
    int raw_f32(const std::string& name, const std::vector<int64_t>& shape, float**) override {
        entries.push_back({name, shape});
        return 0;
    }
};

// Packs reference-layout tensors into plan-owned device memory (fp16 GEMM operands, fp32 vectors).
struct Packer : ParamSink {
    std::unordered_map<std::string, const ctrl_tensor_ref*> map;
    std::vector<void*> allocs;
    size_t bytes = 0;
    hipStream_t s;
    int* ovf = nullptr;          // device flag: set by a packer kernel that met a weight outside the fp16 range
    bool split_paired = true;    // split-operand convs: paired walk on the plain weight pack (false: weights packed twice)
    Packer(const ctrl_tensor_ref* t, int n, hipStream_t stream) : s(stream) {
        for (int i = 0; i < n; ++i) map[t[i].name] = &t[i];
    }
    void release_all() { for (void* p : allocs) hipFree(p); allocs.clear(); if (ovf) hipFree(ovf); ovf = nullptr; }
    int* ovf_flag() {
        if (!ovf && hipMalloc((void**)&ovf, sizeof(int)) == hipSuccess) (void)hipMemsetAsync(ovf, 0, sizeof(int), s);
        return ovf;
    }
    // end of packing: wait for the packed copies (the source tensors may be freed afterwards) and refuse checkpoints
    // whose GEMM weights do not fit the fp16 MFMA operand format (bf16 / fp32 values beyond +-65504, inf, nan)
    int finish() {
        HIP_TRY(hipStreamSynchronize(s));
        if (ovf) {
            int h = 0;
            HIP_TRY(hipMemcpy(&h, ovf, sizeof(int), hipMemcpyDeviceToHost));
            CTRL_CHECK(h == 0, "a weight tensor holds values outside the fp16 range (|w| > 65504, inf or nan): the MFMA "
                               "operand format of this library cannot represent this checkpoint");
        }
        return 0;
    }
    int dalloc(size_t nbytes, void** out) {
        HIP_TRY(hipMalloc(out, nbytes ? nbytes : 16));
        allocs.push_back(*out);
        bytes += nbytes;
        return 0;
    }
    int get(const std::string& name, const std::vector<int64_t>& shape, const ctrl_tensor_ref** out) {
        auto it = map.find(name);
        CTRL_CHECK(it != map.end(), "missing parameter '" + name + "'");
        const ctrl_tensor_ref* t = it->second;
        int64_t want = 1, have = 1;
        for (auto v : shape) want *= v;
        for (int i = 0; i < t->ndim; ++i) have *= t->shape[i];
        bool same = ((int)shape.size() == t->ndim);
        for (int i = 0; same && i < t->ndim; ++i) same = (shape[i] == t->shape[i]);
        CTRL_CHECK(same && want == have, "parameter '" + name + "' has the wrong shape");
        CTRL_CHECK(t->data != nullptr, "parameter '" + name + "' has no data");
        *out = t;
        return 0;
    }
    int vec(const std::string& name, int N, bool geglu, float* dst) {
        const ctrl_tensor_ref* t;
        TRY(get(name, {N}, &t));
        return op_pack_vec(t->data, t->dtype, dst, N, geglu ? 1 : 0, s);
    }
    int linear(const std::string& name, int N, int K, bool bias, bool geglu, Lin* out) override {
        const ctrl_tensor_ref* t;
        TRY(get(name + ".weight", {N, K}, &t));
        TRY(dalloc((size_t)N * K * sizeof(half_t), (void**)&out->w));
        TRY(op_pack_linear_w(t->data, t->dtype, out->w, N, K, geglu ? 1 : 0, s, ovf_flag()));
        out->b = nullptr;
        if (bias) {
            TRY(dalloc((size_t)N * sizeof(float), (void**)&out->b));
            TRY(vec(name + ".bias", N, geglu, out->b));
        }
        out->N = N; out->K = K; out->geglu = geglu;
        return 0;
    }
    int linear_cat(const std::vector<std::string>& names, const std::vector<int>& Ns, int K, bool bias, Lin* out) override {
        int Ntot = 0;
        for (int n : Ns) Ntot += n;
        TRY(dalloc((size_t)Ntot * K * sizeof(half_t), (void**)&out->w));
        out->b = nullptr;
        if (bias) TRY(dalloc((size_t)Ntot * sizeof(float), (void**)&out->b));
        int off = 0;
        for (size_t i = 0; i < names.size(); ++i) {
            const ctrl_tensor_ref* t;
            TRY(get(names[i] + ".weight", {Ns[i], K}, &t));
            TRY(op_pack_linear_w(t->data, t->dtype, out->w + (size_t)off * K, Ns[i], K, 0, s, ovf_flag()));
            if (bias) TRY(vec(names[i] + ".bias", Ns[i], false, out->b + off));
            off += Ns[i];
        }
        out->N = Ntot; out->K = K; out->geglu = false;
        return 0;
    }
    int conv(const std::string& name, int Cout, int Cin, int k, bool temporal, ConvW* out, bool dup) override {
        const ctrl_tensor_ref* t;
        const int taps = temporal ? 3 : k * k;
        if (temporal) TRY(get(name + ".weight", {Cout, Cin, 3, 1, 1}, &t));
        else TRY(get(name + ".weight", {Cout, Cin, k, k}, &t));
        const bool paired = dup && split_paired && (Cin % 64 == 0) && !temporal;
        TRY(dalloc((size_t)Cout * Cin * taps * ((dup && !paired) ? 2 : 1) * sizeof(half_t), (void**)&out->w));
        if (dup && !paired) TRY(op_pack_conv_w_dup(t->data, t->dtype, out->w, Cout, Cin, taps, s, ovf_flag()));
        else TRY(op_pack_conv_w(t->data, t->dtype, out->w, Cout, Cin, taps, s, ovf_flag()));
        out->paired = paired;
        TRY(dalloc((size_t)Cout * sizeof(float), (void**)&out->b));
        TRY(vec(name + ".bias", Cout, false, out->b));
        out->Cout = Cout; out->Cin = dup ? 2 * Cin : Cin; out->taps = taps; out->dup = dup;
        return 0;
    }
    int conv_direct(const std::string& name, int Cout, int Cin, ConvD* out) override {
        const ctrl_tensor_ref* t;
        TRY(get(name + ".weight", {Cout, Cin, 3, 3}, &t));
        TRY(dalloc((size_t)9 * Cin * Cout * sizeof(float), (void**)&out->w));
        TRY(op_pack_conv_w_direct(t->data, t->dtype, out->w, Cout, Cin, s));
        out->w16 = nullptr;
        if (op_conv3x3_small_mfma_applies(Cin, Cout)) {
            TRY(dalloc((size_t)9 * Cin * Cout * sizeof(half_t), (void**)&out->w16));
            TRY(op_pack_conv_w(t->data, t->dtype, out->w16, Cout, Cin, 9, s, ovf_flag()));
        }
        TRY(dalloc((size_t)Cout * sizeof(float), (void**)&out->b));
        TRY(vec(name + ".bias", Cout, false, out->b));
        out->Cout = Cout; out->Cin = Cin;
        return 0;
    }
    int norm(const std::string& name, int C, Norm* out) override {
        TRY(dalloc((size_t)C * sizeof(float), (void**)&out->g));
        TRY(dalloc((size_t)C * sizeof(float), (void**)&out->b));
        TRY(vec(name + ".weight", C, false, out->g));
        TRY(vec(name + ".bias", C, false, out->b));
        out->C = C;
        return 0;
    }
    int scalar(const std::string& name, float** out) override {
        TRY(dalloc(sizeof(float), (void**)out));
        return vec(name, 1, false, *out);
    }
    int ffn_perm(Lin* ff2) override {
        if (!op_ffn_fused_shape_ok(ff2->N, ff2->K) || !ff2->w) return 0;
        TRY(dalloc((size_t)ff2->N * ff2->K * sizeof(half_t), (void**)&ff2->wperm));
        return op_ffn_pack_w2(ff2->w, ff2->wperm, ff2->N, ff2->K, s);
    }
    // load-time scan of every 1-D "...norm....weight" tensor (converted on the device, read back once: plan creation only)
    std::unordered_map<std::string, float> spread_by_name;
    bool spread_scanned = false;
    void scan_norm_scales() {
        spread_scanned = true;
        size_t maxc = 0;
        std::vector<const ctrl_tensor_ref*> ts;
        for (const auto& kv : map) {
            const std::string& n = kv.first;
            const ctrl_tensor_ref* t = kv.second;
            if (t->ndim != 1 || !t->data || n.size() < 7 || n.compare(n.size() - 7, 7, ".weight") != 0 || n.find("norm") == std::string::npos) continue;
            ts.push_back(t);
            if ((size_t)t->shape[0] > maxc) maxc = (size_t)t->shape[0];
        }
        if (ts.empty()) return;
        float* tmp = nullptr;
        if (hipMalloc((void**)&tmp, maxc * sizeof(float)) != hipSuccess) return;
        std::vector<float> h(maxc);
        for (const ctrl_tensor_ref* t : ts) {
            const int C = (int)t->shape[0];
            if (op_pack_vec(t->data, t->dtype, tmp, C, 0, s) != 0) continue;
            if (hipMemcpyAsync(h.data(), tmp, (size_t)C * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) continue;
            if (hipStreamSynchronize(s) != hipSuccess) continue;
            std::vector<float> a(h.begin(), h.begin() + C);
            float mx = 0.f;
            for (float& v : a) { v = v < 0 ? -v : v; if (v > mx) mx = v; }
            std::nth_element(a.begin(), a.begin() + C / 2, a.end());
            const float med = a[C / 2];
            spread_by_name[t->name] = med > 0.f ? mx / med : (mx > 0.f ? 1e9f : 1.f);
        }
        (void)hipFree(tmp);
    }
    float norm_scale_spread(const std::string& prefix) override {
        if (!spread_scanned) scan_norm_scales();
        float worst = 0.f;
        for (const auto& kv : spread_by_name)
            if (kv.first.compare(0, prefix.size(), prefix) == 0 && kv.second > worst) worst = kv.second;
        return worst;
    }
    int raw_f32(const std::string& name, const std::vector<int64_t>& shape, float** out) override {
        const ctrl_tensor_ref* t;
        TRY(get(name, shape, &t));
        int64_t n = 1;
        for (auto v : shape) n *= v;
        TRY(dalloc((size_t)n * sizeof(float), (void**)out));
        return op_pack_vec(t->data, t->dtype, *out, (int)n, 0, s);
    }
};

// ------------------------------------------------------------------------------------------ workspace
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    std::vector<void*> retired;      // outgrown blocks: kept until the plan is destroyed (see ensure)
    ~Arena() {
        if (base) hipFree(base);
        for (void* p : retired) hipFree(p);
    }
    // Grows by allocating a NEW block; the old one is retired, not freed: launches already queued on any stream and
    // hipGraphs captured with the old addresses baked in (bench.py, per-shape graphs of a server) keep reading and
    // writing valid memory.  The cost is the retained bytes of the smaller shapes seen before the largest one.
    int ensure(size_t need) {
        if (need <= cap) return 0;
        char* nb = nullptr;
        HIP_TRY(hipMalloc((void**)&nb, need));
        if (base) retired.push_back(base);
        base = nb; cap = need;
        return 0;
    }
    // frees the retired blocks (ctrl_*_trim): the caller guarantees that no captured graph still addresses them
    size_t trim() {
        size_t n = retired.size();
        for (void* p : retired) (void)hipFree(p);
        retired.clear();
        return n;
    }
};

// State every plan carries besides its weights: the device it lives on and the stream that used its workspace last.
struct PlanBase {
    int device = 0;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    hipEvent_t last_done = nullptr;
    int init_base(const void* any_param) {
        hipPointerAttribute_t at;
        if (any_param && hipPointerGetAttributes(&at, any_param) == hipSuccess) device = at.device;
        else HIP_TRY(hipGetDevice(&device));
        return 0;
    }
    ~PlanBase() { if (last_done) (void)hipEventDestroy(last_done); }
    // Two forwards on different streams share one workspace: the later one waits for the earlier one's last launch.
    // (Inside a stream capture nothing is recorded or waited for: a graph replay is ordered by its launch stream.)
    hipStream_t cap_stream = nullptr;      // stream of the last forward that ran under capture (trim refuses while it still captures)
    bool capture_active() const {
        if (!cap_stream) return false;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        return hipStreamIsCapturing(cap_stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    }
    int enter(hipStream_t s, bool* capturing) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cs);
        *capturing = (cs == hipStreamCaptureStatusActive);
        cap_stream = *capturing ? s : nullptr;
        if (!*capturing && last_valid && last_stream != s) HIP_TRY(hipStreamWaitEvent(s, last_done, 0));
        return 0;
    }
    int leave(hipStream_t s, bool capturing) {
        if (capturing) return 0;
        if (!last_done) HIP_TRY(hipEventCreateWithFlags(&last_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(last_done, s));
        last_stream = s; last_valid = true;
        return 0;
    }
};
// RAII: run a plan call on the plan's device (model.to('cuda:1') without torch.cuda.set_device), restore afterwards
struct DeviceGuard {
    int prev = -1, want;
    explicit DeviceGuard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

// Step-invariant cache of the cross-attention K / V^T projections of the text states: the
// encoder_hidden_states of a request do not change over its ~50 denoise steps, but the reference recomputes to_k / to_v of
// every cross-attention each step.  mode KEEP: the projections are written into plan-owned buffers (allocated on that
// forward: run it eagerly, not under stream capture); mode REUSE: the GEMMs are skipped and the buffers are read.
struct KvCache {
    enum { OFF = 0, KEEP = 1, REUSE = 2 };
    int mode = OFF;
    size_t next = 0;                     // index of the next cross-attention of the forward
    struct Slot { half_t* k = nullptr; half_t* vt = nullptr; size_t k_elems = 0, vt_elems = 0; };
    std::vector<Slot> slots;
    std::vector<void*> retired;          // outgrown buffers stay alive (captured graphs may address them)
    int key_batch = 0, key_Lk = 0;       // what the cache was filled for
    ~KvCache() {
        for (auto& s : slots) { if (s.k) hipFree(s.k); if (s.vt) hipFree(s.vt); }
        for (void* p : retired) hipFree(p);
    }
    size_t trim() {
        size_t n = retired.size();
        for (void* p : retired) (void)hipFree(p);
        retired.clear();
        return n;
    }
    // `capturing`: the forward is being recorded into a hipGraph -- a KEEP forward that has to allocate cannot be captured
    int get(size_t k_elems, size_t vt_elems, bool dry, Slot** out, bool capturing = false) {
        if (next >= slots.size()) slots.resize(next + 1);
        Slot& s = slots[next++];
        if (!dry && mode == KEEP && (s.k_elems < k_elems || s.vt_elems < vt_elems)) {
            CTRL_CHECK(!capturing, "text K/V cache: the first KEEP forward of a shape allocates its buffers and cannot run under "
                                   "stream capture -- run it eagerly once, then capture (mode REUSE or KEEP with the buffers in place)");
            if (s.k) retired.push_back(s.k);
            if (s.vt) retired.push_back(s.vt);
            s.k = s.vt = nullptr;
            HIP_TRY(hipMalloc((void**)&s.k, k_elems * sizeof(half_t)));
            HIP_TRY(hipMalloc((void**)&s.vt, vt_elems * sizeof(half_t)));
            s.k_elems = k_elems; s.vt_elems = vt_elems;
        }
        if (!dry && mode == REUSE) CTRL_CHECK(s.k && s.k_elems >= k_elems && s.vt_elems >= vt_elems, "text K/V cache: REUSE without a matching KEEP forward");
        *out = &s;
        return 0;
    }
};

// Recorded op sequence of one sibling block (grouped launches, ops.h: OpCollector): while Ctx::rec is set, RUN() appends the op as a
// closure instead of enqueueing it; replay_lockstep() then runs the sequences of the siblings position by position.
struct OpList { std::vector<std::function<int()>> ops; };

// Execution context: `dry` = sizing pass (allocations only advance the offset, nothing is launched).
struct Ctx {
    Arena* ar;
    hipStream_t s;
    bool dry;
    OpList* rec = nullptr;     // record instead of launching (see OpList)
    bool f32stream = true;     // residual streams kept in fp32 (set from CTRL_STREAM_F32, default on)
    bool split = false;        // GEMM-operand mirrors of the streams are split [hi | lo] rows (ControlNet, CTRL_CN_SPLIT)
    bool h1_f16 = false;       // a ResNet's conv1 output (read by GroupNorm only) in fp16 instead of the stream dtype (adapter, adapter_h1_f16())
    KvCache* kvc = nullptr;    // text K/V cache of the plan (mode OFF: not used)
    bool capturing = false;    // the forward is being recorded into a hipGraph (no allocation may happen)
    // pooled GroupNorm statistics (zeroed once per forward with a single memset)
    float* stats_base = nullptr;
    size_t stats_off = 0, stats_total = 0;

    void* alloc(size_t bytes) {
        const size_t a = (ar->off + 255) & ~(size_t)255;
        ar->off = a + bytes;
        if (ar->off > ar->peak) ar->peak = ar->off;
        return dry ? (void*)(uintptr_t)(0x1000 + a) : (void*)(ar->base + a);   // dry: fake, never dereferenced
    }
    half_t* h(size_t n) { return (half_t*)alloc(n * sizeof(half_t)); }
    float* f(size_t n) { return (float*)alloc(n * sizeof(float)); }
    size_t mark() const { return ar->off; }
    void release(size_t m) { ar->off = m; }
    float* stats(size_t n) {
        float* p = dry ? nullptr : stats_base + stats_off;
        stats_off += n;
        if (dry) stats_total = stats_off;
        return p;
    }
};
// (the closure copies what the expression names -- the context with its stream, descriptors, pointers: everything an op call takes
// is a value or a pointer into plan-owned memory that outlives the forward)
#define RUN(cx, expr)                                                                      \
    do {                                                                                   \
        if (!(cx).dry) {                                                                   \
            if ((cx).rec) (cx).rec->ops.push_back([=]() -> int { return (expr); });        \
            else TRY(expr);                                                                \
        }                                                                                  \
    } while (0)
// Replays n recorded sequences in lock-step: position k of every sibling is issued with a collector installed, so that the ops of the
// grouped kinds (implicit GEMM, GroupNorm, LayerNorm, attention) deposit their arguments and leave as ONE launch per position when
// the siblings agree; every other op is enqueued as it comes.  Sequences of different lengths are replayed one after the other.
inline int replay_lockstep(OpList* lists, int n) {
    bool same = true;
    for (int i = 1; i < n; ++i) same = same && lists[i].ops.size() == lists[0].ops.size();
    if (!same || n == 1) {
        for (int i = 0; i < n; ++i)
            for (auto& f : lists[i].ops) TRY(f());
        return 0;
    }
    OpCollector col;
    for (size_t k = 0; k < lists[0].ops.size(); ++k) {
        t_collect = &col;
        int rc = 0;
        for (int i = 0; i < n && !rc; ++i) rc = lists[i].ops[k]();
        if (!rc) rc = col.flush();
        t_collect = nullptr;
        if (rc) return rc;
    }
    return 0;
}
// first call of every forward body: carve (real pass) the pooled GroupNorm statistics and zero them once
inline int begin_forward(Ctx& cx) {
    if (!cx.dry) {
        cx.stats_base = (float*)cx.alloc(cx.stats_total * sizeof(float));
        if (cx.stats_total) TRY(op_fill_zero(cx.stats_base, cx.stats_total * sizeof(float), cx.s));
    }
    return 0;
}
// workspace needed by the real pass, given the finished dry pass
inline size_t workspace_bytes(const Ctx& dry) { return dry.ar->peak + ((dry.stats_total * sizeof(float) + 255) & ~(size_t)255) + 1024; }

// ------------------------------------------------------------------------------------------ block params
struct ResnetW {            // diffusers ResnetBlock2D / the adapter's copy
    Norm norm1, norm2;
    ConvW conv1, conv2, shortcut;
    bool has_shortcut = false;
    int Cin = 0, Cout = 0;
    int temb_off = 0;       // column offset of this block's time projection inside the batched projection
};
struct AttnW { Lin qkv, q, kv, v, out; int heads = 0, D = 0, inner = 0; };   // qkv (self) or q + kv (cross); v for Lk==1
struct BasicTBW {           // BasicTransformerBlock
    Norm norm1, norm2, norm3;
    AttnW attn1, attn2;
    Lin ff1, ff2;           // GEGLU proj (interleaved), out
    int dim = 0, cross = 0;
};
struct TemporalTBW {        // TemporalBasicTransformerBlock
    Norm norm_in, norm1, norm2, norm3;
    Lin ffin1, ffin2, ff1, ff2;
    AttnW attn1, attn2;
    int dim = 0, cross = 0;
};

// dup: split [hi | lo] operands for conv1 / conv2; dup_shortcut: the same for the 1x1 shortcut (-1 = as dup)
int build_resnet(ParamSink& ps, const std::string& pre, int Cin, int Cout, bool force_shortcut, ResnetW* w, bool dup = false, int dup_shortcut = -1);
int build_attn_self(ParamSink& ps, const std::string& pre, int dim, int heads, int D, AttnW* w);
int build_attn_cross(ParamSink& ps, const std::string& pre, int dim, int cross, int heads, int D, AttnW* w);
int build_basic_tb(ParamSink& ps, const std::string& pre, int dim, int heads, int D, int cross, BasicTBW* w);
int build_temporal_tb(ParamSink& ps, const std::string& pre, int dim, int heads, int D, int cross, TemporalTBW* w);

// fp32 residual streams are the default; CTRL_STREAM_F32=0 selects fp16 streams (faster by a few %, ~2x the error)
inline bool stream_f32_enabled() {
    return !policy_is0(P_STREAM_F32);
}

// the adapter's spatial-transformer token stream in fp16 (default since round 4; CTRL_ADAPTER_TOK_F16=0: fp32 like the ControlNet's
// and the temporal streams, which stay fp32)
inline bool adapter_tok_f16() {
    return !policy_is0(P_ADAPTER_TOK_F16);
}

inline bool adapter_tok_f16_forced() {
    const char* e = policy_raw(P_ADAPTER_TOK_F16);
    return e && e[0] == 'f';
}

// the adapter ResNets' conv1 -> GroupNorm intermediate in fp16 (CTRL_ADAPTER_H1_F16=1)
inline bool adapter_h1_f16() {
    return policy_is1(P_ADAPTER_H1_F16);
}

// ------------------------------------------------------------------------------------------ tensor views
// A residual-stream tensor: master copy `p` in dtype `dt` (fp32 when Ctx::f32stream, else fp16) and, where a GEMM / conv
// consumes it as an operand, an fp16 copy `m16` (== p for an fp16 master; written by the producing GEMM's epilogue as
// a mirror for an fp32 master).  Keeping the stream in fp32 removes the accumulated fp16 rounding of ~60 sequential
// residual updates (the dominant error term of the chained ControlNet -> adapter path).
struct TV {
    void* p = nullptr;
    int dt = DT_F16;
    half_t* m16 = nullptr;
    int lo_off = 0;      // > 0: m16 rows are [hi | lo] (2 x the width), lo at this column offset (split operand)
    bool ok() const { return p != nullptr; }
};
inline TV tv16(const half_t* x) { TV t; t.p = (void*)x; t.dt = DT_F16; t.m16 = (half_t*)x; return t; }
inline TV stream_alloc(Ctx& cx, size_t n, bool need16) {
    TV t;
    if (cx.f32stream) {
        t.p = cx.f(n); t.dt = DT_F32;
        t.m16 = need16 ? cx.h(n) : nullptr;
    } else {
        t.m16 = cx.h(n); t.p = t.m16; t.dt = DT_F16;
    }
    return t;
}
// stream tensor [rows][C] whose GEMM-operand mirror is a split [hi | lo] row pair when the context asks for it
inline TV stream_alloc_rc(Ctx& cx, size_t rows, int C, bool need16) {
    if (!(cx.split && cx.f32stream && need16)) return stream_alloc(cx, rows * C, need16);
    TV t;
    t.p = cx.f(rows * C); t.dt = DT_F32;
    t.m16 = cx.h(rows * C * 2); t.lo_off = C;
    return t;
}
inline void set_out(IGemmArgs& g, const TV& out, long ld, int ncols) {
    g.nseg = 1;
    g.seg[0] = IGemmSeg{out.p, ld, 0, ncols, SEG_ROW, out.dt, 1, 0};
    g.out16 = nullptr; g.ld16 = 0; g.out16_lo_off = 0;
    if (out.dt == DT_F32 && out.m16) { g.out16 = out.m16; g.ld16 = out.lo_off ? 2 * ld : ld; g.out16_lo_off = out.lo_off; }
}
inline void set_res(IGemmArgs& g, const TV& res, long ld) {
    g.res = res.p; g.ldres = ld; g.res_f32 = (res.ok() && res.dt == DT_F32) ? 1 : 0;
}
// AlphaBlender fold: result = (1-a) * (this GEMM's epilogue value) + a * other, a = sigmoid(*mix)  (mix = null: off)
inline void set_blend(IGemmArgs& g, const float* mix, const TV& other, long ld) {
    g.blend_mix = mix; g.blend_x = mix ? other.p : nullptr; g.ld_blend = ld;
    g.blend_f32 = (mix && other.dt == DT_F32) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------ block runners
// y = GroupNorm(x) (optional SiLU);  x [imgs*rows][C] fp16 or fp32 stream, y fp16
// split: y rows are [hi | lo] (2C wide), the operand of a dup-packed convolution
int run_groupnorm(Ctx& cx, const Norm& n, const TV& x, half_t* y, int imgs, int rows, float eps, bool silu, bool split = false);
// 3x3 / 1x1 conv through the implicit GEMM, NHWC fp16 operand -> NHWC stream tensor
struct ConvOpts {
    int stride = 1, up = 1;
    long lda = 0;              // row stride of the operand (0 = the conv's Cin); 2*C: a plain conv reading the hi half of split rows
    const float* rowvec = nullptr; int rowvec_ld = 0;
    TV res;
    int res_up = 0;            // 2: res is the half-resolution map, read through a nearest x2 up-sampling (ctrl_igemm_desc::res_up)
    int act = 0;
};
int run_conv(Ctx& cx, const ConvW& c, const half_t* x, const TV& y, int N, int Hin, int Win, const ConvOpts& o);
// rowvec (optional): fp32 [M / rows_per_vec][rowvec_ld] added to every row of its group by the GEMM epilogue
// ln / ln_out (optional): the LayerNorm that the NEXT op applies to y, written as fp16 rows [M][l.N] into ln_out right after
// the GEMM (layernorm_kernel).  Computing it in this GEMM's epilogue was built and measured in round 3 (a full-row 128 x 512
// tile, cross-wave row statistics through LDS): 0.286 ms against 0.145 + 0.084 ms for the 128^2 proj_in shape -- one
// workgroup per CU and 78 spilled registers lose more than the saved pass is worth -- and removed again
int run_linear(Ctx& cx, const Lin& l, const half_t* x, long ldx, const TV& y, long ldy, int M, const TV& res, long ldres,
               const float* rowvec = nullptr, int rowvec_ld = 0, int rows_per_vec = 0,
               const float* blend_mix = nullptr, const TV& blend_other = TV(), const Norm* ln = nullptr, half_t* ln_out = nullptr);
// out = ResnetBlock2D(x, temb); temb_proj = rowvec [N][ld] (already time_emb_proj(SiLU(emb)) incl. bias).
// x.m16 must exist when the block has a shortcut conv (it is that conv's operand).
int run_resnet(Ctx& cx, const ResnetW& w, const TV& x, const TV& out, int N, int H, int W, int up,
               const float* temb_proj, int temb_ld, float eps);
// Encoder-hidden-state context prepared once per forward: fp16 copy [B*Lk][cross] (+ fp32 copy when Lk == 1)
struct EhsCtx { const half_t* h16 = nullptr; const float* f32 = nullptr; int batch = 1, Lk = 0, cross = 0;
                long row_ld = 0;   /* floats between the batch rows of f32 (0 = dense, Lk * cross) */ };
// K / V^T projections of the text states of one cross-attention, computed ahead of the block that uses them (they depend
// on encoder_hidden_states only): project_text_kv enqueues the two GEMMs on cx.s (or takes the plan's text cache)
struct PreKV { half_t* k = nullptr; half_t* vt = nullptr; };
int project_text_kv(Ctx& cx, const AttnW& w, const EhsCtx& e, PreKV* out);
// X [B*L][dim] stream -> out stream (out.m16 is filled when the caller needs a GEMM-operand copy)
// ov_pre (optional, Lk == 1 only): the single-key cross-attention vector computed ahead of time (single_key_vector)
// X_ln (optional): LayerNorm(norm1)(X) already computed by X's producer (run_linear(..., &w.norm1, X_ln))
int run_basic_tb(Ctx& cx, const BasicTBW& w, const TV& X, const TV& out, int B, int L, const EhsCtx& e, const float* ov_pre = nullptr,
                 const PreKV* kv_pre = nullptr, const half_t* X_ln = nullptr);
// [e.batch][dim] fp32 output of a one-key cross-attention (query independent, note N5)
int single_key_vector(Ctx& cx, const AttnW& w, int dim, const EhsCtx& e, float** out);
int run_layernorm(Ctx& cx, const Norm& n, const TV& x, half_t* y, int M, int dim);
// out = res + ff2(GEGLU(ff1(xn)))  [+ AlphaBlender fold]  -- diffusers FeedForward(GEGLU) on LayerNorm'd tokens xn [M][dim]; ONE launch
// (ffn.hip) for the dim-512 blocks at M >= kFfnFusedMinM unless CTRL_FF_FUSED=0, else the GEGLU GEMM + the output GEMM.
// ln / ln_out (optional): the LayerNorm the next op applies to `out`, launched right behind (as run_linear does)
constexpr int kFfnFusedMinM = 16384;
int run_ffn(Ctx& cx, const Lin& ff1, const Lin& ff2, const half_t* xn, int dim, const TV& out, int M, const TV& res,
            const float* blend_mix = nullptr, const TV& blend_other = TV(), const Norm* ln = nullptr, half_t* ln_out = nullptr);
