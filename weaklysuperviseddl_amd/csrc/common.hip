// common.hip - error channel, version, and the per-kernel-class event timing used by bench.py.
#include "common.h"

#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace wsdl {

static thread_local char g_err[512] = "";
Opt g_range_sentinel{0};
Opt g_bn_coop{0}, g_bn_coop_wide{0};      // off by default: four workgroups per channel pay at 64 channels only (0.3 % of the step), and
                                            // workgroups that wait for each other are not something to have on by default (r05_notes.md)

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

std::atomic<int> g_trace_launches{0};
static thread_local char g_trace[2048] = "";
static thread_local size_t g_trace_len = 0;
void trace_launch(const char* fmt, ...) {
    if (g_trace_len + 4 >= sizeof(g_trace)) return;
    if (g_trace_len) g_trace_len += (size_t)snprintf(g_trace + g_trace_len, sizeof(g_trace) - g_trace_len, "; ");
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(g_trace + g_trace_len, sizeof(g_trace) - g_trace_len, fmt, ap);
    va_end(ap);
    if (n > 0) g_trace_len = std::min(sizeof(g_trace) - 1, g_trace_len + (size_t)n);
}

struct ProfRec {
    hipEvent_t start, stop;
    double work, executed, bytes;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof[WSDL_PROF_NCLASSES];
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free_events;

bool prof_enabled() { return g_prof_on; }

// roctx, loaded on demand (the library carries no link-time dependency on the profiler SDK)
static bool g_ranges_on = false;
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static bool load_roctx() {
    if (g_roctx_push && g_roctx_pop) return true;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
            g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (g_roctx_push && g_roctx_pop) return true;
        }
    }
    return false;
}

ProfScope::ProfScope(int c, hipStream_t st, double work, double work_executed, double bytes)
    : cls(c), s(st), slot(nullptr) {
    if (g_ranges_on) g_roctx_push(wsdl_prof_class_name(c));
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.work = work;
    r.executed = work_executed >= 0.0 ? work_executed : work;
    r.bytes = bytes;
    if (!g_free_events.empty()) {
        r.start = g_free_events.back().first;
        r.stop = g_free_events.back().second;
        g_free_events.pop_back();
    } else {
        if (hipEventCreate(&r.start) != hipSuccess) return;
        if (hipEventCreate(&r.stop) != hipSuccess) return;
    }
    (void)hipEventRecord(r.start, s);
    g_prof[cls].push_back(r);
    slot = reinterpret_cast<void*>(g_prof[cls].size());  // index + 1
}

ProfScope::~ProfScope() {
    if (g_ranges_on) g_roctx_pop();
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    size_t idx = reinterpret_cast<size_t>(slot) - 1;
    if (idx < g_prof[cls].size()) (void)hipEventRecord(g_prof[cls][idx].stop, s);
}

}  // namespace wsdl

extern "C" {

int wsdl_launch_trace(int on) {
    wsdl::g_trace_launches.store(on ? 1 : 0);
    wsdl::g_trace[0] = 0;
    wsdl::g_trace_len = 0;
    return WSDL_OK;
}

const char* wsdl_last_launches(void) {      // this thread's launches since the previous call (the string stays valid until the next one)
    static thread_local char out[sizeof(wsdl::g_trace)];
    memcpy(out, wsdl::g_trace, sizeof(out));
    wsdl::g_trace[0] = 0;
    wsdl::g_trace_len = 0;
    return out;
}

const char* wsdl_last_error(void) { return wsdl::g_err; }
int wsdl_version(void) { return 100; }
const char* wsdl_target_arch(void) { return "gfx950"; }

const char* wsdl_prof_class_name(int cls) {
    static const char* names[WSDL_PROF_NCLASSES] = {
        "conv_igemm_fast_kernel<128, 128, 2, 16>", "conv_igemm_kernel<128, 128, 2, false>",
        "conv_igemm_fast_kernel<128, 64, 2, 32>", "conv_igemm_kernel<128, 64, 2, false>",
        "conv_igemm_fast_kernel<64, 256, 1, 16>", "conv_igemm_kernel<64, 256, 1, false>",
        "conv_igemm_fast_kernel<64, 128, 1, 32>", "conv_igemm_kernel<64, 128, 1, false>",
        "conv_wgrad_kernel<128, 128, 2>", "conv_wgrad_kernel<64, 128, 1>", "conv_wgrad_fast_kernel<128, 128, 2, 16>",
        "pairwise_kernel", "layercam_partial_kernel",
        "conv_igemm_split_kernel<128, 128, 2, 16, 256, AR>", "conv_igemm_split_kernel<128, 64, 2, 32, 256, AR>",
        "conv_igemm_split_kernel<64, 256, 1, 16, 256, AR>", "conv_igemm_split_kernel<64, 128, 1, 32, 256, AR>",
        "conv_wgrad_split16_kernel<128, 128, false>", "conv_igemm_split_kernel<256, 128, 4, BK, 512, AR, false, false>",
        "stem_conv7x7s2_kernel", "conv_wgrad_split16d_kernel<MODE, DYRAW>",
        "conv_igemm_split_group_kernel<256, 128, 4, 16, 512, AR, false>",
        "conv_igemm_split_kernel<256, 128, 4, 16, 512, AR, false, true>"};
    // (BK: 16 in the 256x128 form - the profiler's name has the number; the split weight-gradient class: the default fp16x2 kernel's name; bf16x3 launches conv_wgrad_split32_kernel)
    return cls >= 0 && cls < WSDL_PROF_NCLASSES ? names[cls] : "?";
}

int wsdl_range_enable(int on) {
    if (on && !wsdl::load_roctx()) {
        wsdl::set_error("range_enable: librocprofiler-sdk-roctx.so / libroctx64.so could not be loaded");
        return WSDL_EINVAL;
    }
    wsdl::g_ranges_on = on != 0;
    return WSDL_OK;
}
int wsdl_range_push(const char* name) {
    if (wsdl::g_ranges_on && name) wsdl::g_roctx_push(name);
    return WSDL_OK;
}
int wsdl_range_pop(void) {
    if (wsdl::g_ranges_on) wsdl::g_roctx_pop();
    return WSDL_OK;
}

int wsdl_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(wsdl::g_prof_mu);
    wsdl::g_prof_on = on != 0;
    return WSDL_OK;
}

int wsdl_prof_reset(void) {
    std::lock_guard<std::mutex> lk(wsdl::g_prof_mu);
    for (auto& v : wsdl::g_prof) {
        for (auto& r : v) wsdl::g_free_events.emplace_back(r.start, r.stop);
        v.clear();
    }
    return WSDL_OK;
}

int wsdl_prof_collect(int cls, long long* launches, double* total_ms, double* total_work,
                      double* total_work_executed, double* total_bytes) {
    WSDL_REQUIRE(cls >= 0 && cls < WSDL_PROF_NCLASSES, "prof class %d out of range", cls);
    std::lock_guard<std::mutex> lk(wsdl::g_prof_mu);
    double ms = 0.0, work = 0.0, exe = 0.0, byt = 0.0;
    for (auto& r : wsdl::g_prof[cls]) {
        WSDL_HIP_CHECK(hipEventSynchronize(r.stop));
        float t = 0.f;
        WSDL_HIP_CHECK(hipEventElapsedTime(&t, r.start, r.stop));
        ms += t;
        work += r.work;
        exe += r.executed;
        byt += r.bytes;
    }
    if (launches) *launches = (long long)wsdl::g_prof[cls].size();
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    if (total_work_executed) *total_work_executed = exe;
    if (total_bytes) *total_bytes = byt;
    return WSDL_OK;
}

}  // extern "C"
