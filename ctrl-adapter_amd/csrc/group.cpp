// Collector of the grouped launches (ops.h: OpCollector): deposits of one lock-step position, launched together by flush().
#include "ops.h"
#include "../../include/ctrl_hip.h"
#include <cstdlib>

thread_local OpCollector* t_collect = nullptr;

static int group_mode() {
    const char* e = policy_raw(P_GROUP);
    return !e ? 1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1));
}
bool group_launches_enabled() { return group_mode() != 0; }
bool group_tiles_as_alone() { return group_mode() == 2; }
extern "C" int ctrl_group_launches(int on) {
    if (on >= 0 && on <= 2) { const char* v[3] = {"0", "1", "2"}; (void)ctrl_policy_set("CTRL_GROUP", v[on]); }
    return group_mode();
}

int OpCollector::slot(int type_, hipStream_t s_, int* rc) {
    *rc = 0;
    if (n > 0 && (type != type_ || s != s_ || n == kMaxGroup)) {
        *rc = flush();
        if (*rc) return -1;
    }
    type = type_; s = s_;
    return n++;
}

int OpCollector::flush() {
    if (n == 0) return 0;
    OpCollector* const me = t_collect;
    t_collect = nullptr;                 // the group ops below launch for real
    int rc = 0;
    switch (type) {
        case IGEMM: rc = op_igemm_group(ig, n, s); break;
        case GN_STATS: rc = op_gn_stats_group(gs, n, s); break;
        case GN_APPLY: rc = op_gn_apply_group(ga, n, s); break;
        case LAYERNORM: rc = op_layernorm_group(ln, n, s); break;
        case ATTN: rc = op_flash_attn_group(at, n, s); break;
        case GN_FUSED: rc = op_gn_fused_group(ga, n, s); break;
        case FFN: {
            // (problems of different M cannot share the launch: one by one then)
            bool same = true;
            for (int i = 1; i < n; ++i) same = same && ff[i].out.M == ff[0].out.M;
            if (same) rc = op_ffn_fused_group(ff, n, s);
            else for (int i = 0; i < n && !rc; ++i) rc = op_ffn_fused_group(&ff[i], 1, s);
            break;
        }
        case FILL: {
            void* ps[kMaxGroup]; size_t bs[kMaxGroup];
            for (int i = 0; i < n; ++i) { ps[i] = fz[i].p; bs[i] = fz[i].bytes; }
            rc = op_fill_zero_group(ps, bs, n, s);
            break;
        }
        default: break;
    }
    t_collect = me;
    n = 0; type = NONE;
    return rc;
}
