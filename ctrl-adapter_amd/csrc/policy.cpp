// The policy table (policy.h): one snapshot of the CTRL_* environment, overridable through the test / experiment ABI.
#include "policy.h"
#include "common.h"
#include "../../include/ctrl_hip.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace {
const char* const kNames[P_COUNT] = {
    "CTRL_GROUP", "CTRL_CLIP_A2A", "CTRL_ADAPTER_SPLIT_TOP", "CTRL_ADAPTER_LANES", "CTRL_QKV_ONE", "CTRL_CN_SPLIT", "CTRL_CN_SPLIT_LEVELS",
    "CTRL_CN_SPLIT_RESNET_LEVELS", "CTRL_SMALLCONV_MFMA", "CTRL_CN_AUX", "CTRL_STEP_OVERLAP", "CTRL_CHECK_FINITE", "CTRL_PROF_DUMP",
    "CTRL_STREAM_F32", "CTRL_ADAPTER_TOK_F16", "CTRL_ADAPTER_H1_F16", "CTRL_ATTN_NW4", "CTRL_ATTN_VARIANT", "CTRL_IGEMM_ORDER", "CTRL_IGEMM8",
    "CTRL_SPLITK_INLAUNCH", "CTRL_IGEMM_FORCE", "CTRL_SHORTK_PAIR", "CTRL_SMALL_TILES", "CTRL_GN_FUSED", "CTRL_FF_FUSED",
    "CTRL_MULTI_CN_LANES", "CTRL_CN_BATCH_LANES",
};
struct Entry { bool set = false; std::string v; };
Entry g_tab[P_COUNT];
// a value handed out by policy_raw stays valid: an override never frees the string it replaces (the handful of ctrl_policy_set calls
// of a test process leak a few bytes each)
const char* g_ptr[P_COUNT] = {};
std::once_flag g_once;
std::mutex g_mu;
void snapshot() {
    for (int k = 0; k < P_COUNT; ++k) {
        const char* e = getenv(kNames[k]);
        g_tab[k].set = e != nullptr;
        if (e) { g_tab[k].v = e; g_ptr[k] = g_tab[k].v.c_str(); }
    }
}
int key_of(const char* name) {
    if (!name) return -1;
    for (int k = 0; k < P_COUNT; ++k)
        if (!strcmp(name, kNames[k])) return k;
    return -1;
}
}  // namespace

const char* policy_raw(PolicyKey k) {
    std::call_once(g_once, snapshot);
    return __atomic_load_n(&g_ptr[k], __ATOMIC_ACQUIRE);
}
int policy_int(PolicyKey k, int dflt) {
    const char* e = policy_raw(k);
    return e ? atoi(e) : dflt;
}

extern "C" {
int ctrl_policy_set(const char* name, const char* value) {
    const int k = key_of(name);
    CTRL_CHECK(k >= 0, std::string("policy_set: unknown variable '") + (name ? name : "(null)") + "'");
    std::call_once(g_once, snapshot);
    std::lock_guard<std::mutex> lk(g_mu);
    const char* p = nullptr;
    if (value) {
        char* c = (char*)malloc(strlen(value) + 1);      // never freed: see g_ptr
        CTRL_CHECK(c != nullptr, "policy_set: out of memory");
        strcpy(c, value);
        p = c;
    }
    __atomic_store_n(&g_ptr[k], p, __ATOMIC_RELEASE);
    return 0;
}
const char* ctrl_policy_get(const char* name) {
    const int k = key_of(name);
    return k < 0 ? nullptr : policy_raw((PolicyKey)k);
}
int ctrl_policy_count(void) { return P_COUNT; }
const char* ctrl_policy_name(int i) { return (i >= 0 && i < P_COUNT) ? kNames[i] : nullptr; }
}
