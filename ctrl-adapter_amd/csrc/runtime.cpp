// Runtime glue of libctrlhip: thread-local error string and the per-kernel-class HIP-event profiler
// used by bench.py's roofline leg (events are recorded on the stream the kernels are launched on).
#include "common.h"
#include "policy.h"
#include "../../include/ctrl_hip.h"
#include <map>
#include <mutex>
#include <vector>
#include <cstring>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

static thread_local std::string g_err;
void ctrl_set_error(const std::string& s) { g_err = s; }

const void* device_zero_page() {
    static void* z[kMaxDevices] = {};
    static std::mutex mu;
    const int d = cur_device();
    std::lock_guard<std::mutex> lk(mu);
    if (!z[d]) {
        if (hipMalloc(&z[d], 4096) != hipSuccess) return nullptr;
        if (hipMemset(z[d], 0, 4096) != hipSuccess) return nullptr;
    }
    return z[d];
}

// ---- range check of the fp16 activations (ctrl_range_check / CTRL_CHECK_FINITE) ----
static int g_range_on = -1;
bool range_check_on() {
    if (g_range_on < 0) g_range_on = policy_is1(P_CHECK_FINITE) ? 1 : 0;
    return g_range_on == 1;
}
int* range_flag() {
    static int* f[kMaxDevices] = {};
    static std::mutex mu;
    const int d = cur_device();
    std::lock_guard<std::mutex> lk(mu);
    if (!f[d]) {
        if (hipMalloc((void**)&f[d], 256) != hipSuccess) return nullptr;
        if (hipMemset(f[d], 0, 256) != hipSuccess) return nullptr;
    }
    return f[d];
}
extern "C" int ctrl_range_check(int on) {
    if (on == 0 || on == 1) g_range_on = on;
    // the flag word is allocated HERE, eagerly (its first use may otherwise fall inside a stream capture, where hipMalloc / hipMemset
    // are illegal); -1: switched on but the flag could not be allocated -- op_igemm then fails loudly instead of checking nothing
    if (range_check_on() && !range_flag()) return -1;
    return range_check_on() ? 1 : 0;
}
extern "C" int ctrl_range_status(int reset) {
    if (!range_check_on()) return 2;
    int* f = range_flag();
    if (!f) return 2;
    int v = 0;
    // (synchronises the device: not to be called while a stream capture is in progress -- include/ctrl_hip.h says so)
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    if (hipMemcpy(&v, f, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (v && reset) (void)hipMemset(f, 0, sizeof(int));
    return v ? 1 : 0;
}

bool g_prof_on = false;
double g_prof_flops = 0, g_prof_bytes = 0;
namespace {
struct Rec { const char* tag; std::string sym; long grid_threads; hipEvent_t e0, e1; double flops, bytes; std::string detail; float ms; };
std::string g_detail, g_sym;
std::vector<Rec> g_launches;      // per-launch records of the last finished profile (ctrl_prof_launch_get)
struct Sum { double ms = 0, flops = 0, bytes = 0; int n = 0; };
std::vector<Rec> g_recs;
std::vector<std::pair<std::string, Sum>> g_summary;
}
void prof_before(const char* tag, const char* kern_expr, long grid_threads, hipStream_t s) {
    Rec r; r.tag = tag; r.flops = g_prof_flops; r.bytes = g_prof_bytes; r.detail = g_detail; g_detail.clear();
    if (g_sym.empty()) {
        // launch expression "(kernel<ARGS>)" -> bare kernel name (tools/pmc_traffic.py:symbol_of spells it the same way)
        std::string e(kern_expr);
        size_t b = e.find_first_not_of("( ");
        size_t en = e.find_first_of("<) ", b == std::string::npos ? 0 : b);
        r.sym = e.substr(b == std::string::npos ? 0 : b, en == std::string::npos ? std::string::npos : en - b);
    } else {
        r.sym = g_sym;
    }
    g_sym.clear();
    r.grid_threads = grid_threads; r.ms = 0.f;
    g_prof_flops = g_prof_bytes = 0;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_detail(const char* fmt, ...) {
    if (!g_prof_on) return;
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_detail = buf;
}
void prof_symbol(const char* fmt, ...) {
    if (!g_prof_on) return;
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_sym = buf;
}
void prof_after(hipStream_t s) { hipEventRecord(g_recs.back().e1, s); }

extern "C" {
int ctrl_abi_version(void) { return CTRL_ABI_VERSION; }
const char* ctrl_last_error(void) { return g_err.c_str(); }
int ctrl_prof_begin(void) { g_recs.clear(); g_summary.clear(); g_launches.clear(); g_prof_on = true; return 0; }
int ctrl_prof_end(void) {
    g_prof_on = false;
    HIP_TRY(hipDeviceSynchronize());
    std::map<std::string, Sum> acc;
    FILE* dump = nullptr;
    if (const char* path = policy_raw(P_PROF_DUMP)) dump = fopen(path, "w");
    for (auto& r : g_recs) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        r.ms = ms;
        if (dump) fprintf(dump, "%s\t%.4f\t%.1f\t%s\t%s\t%ld\n", r.tag, ms, r.flops > 0 ? r.flops / (ms * 1e-3) / 1e12 : 0.0, r.detail.c_str(),
                          r.sym.c_str(), r.grid_threads);
        auto& a = acc[r.tag];
        a.ms += ms; a.n += 1; a.flops += r.flops; a.bytes += r.bytes;
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    if (dump) fclose(dump);
    g_launches.swap(g_recs);
    g_recs.clear();
    g_summary.assign(acc.begin(), acc.end());
    return 0;
}
int ctrl_prof_launch_count(void) { return (int)g_launches.size(); }
int ctrl_prof_launch_get(int i, char* tag, int tag_len, char* symbol, int symbol_len, char* detail, int detail_len,
                         double* ms, double* flops, double* bytes, int64_t* grid_threads) {
    CTRL_CHECK(i >= 0 && i < (int)g_launches.size() && tag && symbol && detail, "prof_launch_get: bad argument");
    const Rec& r = g_launches[i];
    std::strncpy(tag, r.tag, tag_len - 1); tag[tag_len - 1] = 0;
    std::strncpy(symbol, r.sym.c_str(), symbol_len - 1); symbol[symbol_len - 1] = 0;
    std::strncpy(detail, r.detail.c_str(), detail_len - 1); detail[detail_len - 1] = 0;
    if (ms) *ms = r.ms;
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    if (grid_threads) *grid_threads = r.grid_threads;
    return 0;
}
int ctrl_prof_count(void) { return (int)g_summary.size(); }
int ctrl_prof_get(int i, char* name, int name_len, double* total_ms, int* launches, double* flops, double* bytes) {
    CTRL_CHECK(i >= 0 && i < (int)g_summary.size(), "prof_get: index out of range");
    std::strncpy(name, g_summary[i].first.c_str(), name_len - 1);
    name[name_len - 1] = 0;
    *total_ms = g_summary[i].second.ms;
    *launches = g_summary[i].second.n;
    if (flops) *flops = g_summary[i].second.flops;
    if (bytes) *bytes = g_summary[i].second.bytes;
    return 0;
}
}
