// plan.hip - launch plans: record the kernel launches of a sequence of library calls once, issue them again from one C loop.
//
// A training step of the segmentation model is ~520 kernel launches on three streams.  Issued call by call from the host
// language (Python -> ctypes -> entry point -> tile choice -> launch) that is 10-14 ms of host time per 18.7 ms step; the
// runtime's own hipGraph replay costs 9.5 ms on this ROCm.  A plan is the same sequence as a flat table of
// (function, grid, block, LDS, stream, argument values) + the cross-stream dependencies between them, recorded while the
// sequence runs once in the ordinary way (common.h: wsdl::launch) and replayed by wsdl_plan_replay: one host call per step,
// no per-launch host logic.  The arithmetic is the recorded path's, kernel for kernel and argument for argument.
//
// What a plan fixes: every pointer argument (the caller keeps those buffers alive and at their addresses - the host side
// records into a private memory pool), every scalar argument (learning rate, geometry), the streams.  What changes between
// replays must live on the device (Adam's step number, the dropout call counters - as for a hipGraph).
#include "common.h"

#include <time.h>

#include <algorithm>
#include <deque>
#include <mutex>
#include <vector>

namespace wsdl {

enum PlanKind : int { kKernel = 0, kMemset = 1, kStreamWait = 2, kEventRecord = 3, kEventWait = 4, kMark = 5 };

struct PlanOp {
    int kind;
    hipStream_t s;        // launch stream / the waiting stream
    hipStream_t s2;       // kStreamWait: the stream waited for
    const void* fn;
    dim3 grid, block;
    unsigned shmem;
    void** argv;
    void* dst;
    int value;
    size_t bytes;
    hipEvent_t ev;
    long long tag;
};

struct Plan {
    std::vector<PlanOp> ops;
    std::vector<std::shared_ptr<void>> storage;      // argument values, one block per launch
    std::deque<std::vector<void*>> argvs;            // pointers into them (stable addresses)
    std::vector<hipEvent_t> own_events;              // one per kStreamWait
    std::vector<size_t> marks;                       // op index of every kMark, in order
    long long n_kernel = 0, n_memset = 0, n_wait = 0, n_event = 0;
    bool poisoned = false;
    char why[256] = "";
    ~Plan() {
        for (hipEvent_t e : own_events) (void)hipEventDestroy(e);
    }
};

thread_local Plan* g_plan_rec = nullptr;
std::atomic<int> g_plans_recording{0};
static thread_local Plan* g_plan_paused = nullptr;      // wsdl_plan_pause: the recording a host section interrupted

void plan_add_kernel(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, std::shared_ptr<void> storage,
                     void* const* argv, int argc) {
    Plan* p = g_plan_rec;
    if (!p) return;
    p->storage.push_back(std::move(storage));
    p->argvs.emplace_back(argv, argv + argc);
    PlanOp op{};
    op.kind = kKernel;
    op.s = s;
    op.fn = fn;
    op.grid = grid;
    op.block = block;
    op.shmem = (unsigned)shmem;
    op.argv = p->argvs.back().data();
    p->ops.push_back(op);
    ++p->n_kernel;
}

void plan_poison(const char* why) {
    Plan* p = g_plan_rec;
    if (!p || p->poisoned) return;
    p->poisoned = true;
    snprintf(p->why, sizeof(p->why), "%s", why);
}

hipError_t memset_async(void* dst, int value, size_t bytes, hipStream_t s) {
    if (Plan* p = g_plan_rec) {
        PlanOp op{};
        op.kind = kMemset;
        op.s = s;
        op.dst = dst;
        op.value = value;
        op.bytes = bytes;
        p->ops.push_back(op);
        ++p->n_memset;
    }
    return (hipMemsetAsync)(dst, value, bytes, s);
}

// stream-waits-for-stream outside a plan: events from a ring (a wait holds on to the record it saw; re-recording the event
// later does not disturb it)
static hipEvent_t ring_event() {
    static std::mutex mu;                   // the ring is shared by every host thread
    static std::vector<hipEvent_t> ring;
    static size_t next = 0;
    constexpr size_t kRing = 512;
    std::lock_guard<std::mutex> lock(mu);
    if (ring.size() < kRing) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        ring.push_back(e);
        return e;
    }
    hipEvent_t e = ring[next];
    next = (next + 1) % kRing;
    return e;
}

static int replay_range(Plan* p, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
        const PlanOp& op = p->ops[i];
        switch (op.kind) {
            case kKernel:
                WSDL_HIP_CHECK(hipLaunchKernel(op.fn, op.grid, op.block, op.argv, op.shmem, op.s));
                break;
            case kMemset:
                WSDL_HIP_CHECK((hipMemsetAsync)(op.dst, op.value, op.bytes, op.s));
                break;
            case kStreamWait:
                WSDL_HIP_CHECK(hipEventRecord(op.ev, op.s2));
                WSDL_HIP_CHECK(hipStreamWaitEvent(op.s, op.ev, 0));
                break;
            case kEventRecord:
                WSDL_HIP_CHECK(hipEventRecord(op.ev, op.s));
                break;
            case kEventWait:
                WSDL_HIP_CHECK(hipStreamWaitEvent(op.s, op.ev, 0));
                break;
            default:
                break;
        }
    }
    return WSDL_OK;
}

__global__ void add_int_kernel(void* p, int is64, long long delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (is64) *reinterpret_cast<long long*>(p) += delta;
        else *reinterpret_cast<int*>(p) += (int)delta;
    }
}

__global__ void mul_scalars_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

__global__ void clamp_max_i64_kernel(const long long* __restrict__ x, long long* __restrict__ y, long long n, long long hi) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long v = x[i];
        y[i] = v > hi ? hi : v;
    }
}

// out = w * mean(x[0..n)) - the weight of a loss term and the mean over per-image values (reference
// AlternatingDirectionBoundaryLoss.py:199 "0.1 * boundary.mean()"); one workgroup, fixed summation order
__global__ void scale_mean_kernel(const float* __restrict__ x, int n, float w, float* __restrict__ out) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)x[i];
    acc = block_sum_d(acc, sm);
    if (threadIdx.x == 0) out[0] = w * (float)(acc / (double)n);
}

__global__ void scale_fill_kernel(const float* __restrict__ g, float c, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g[0] * c;
}

// (max, ~min channel maximum) pairs of one step's tensors -> out[0] = the largest log2(max / min) over the pairs that hold both,
// out[1] = how many pairs exceed `limit_log2`, out[2] = how many were looked at.  One workgroup, plain stores: `out` may be
// host memory the device can write (the host reads it a step later, no synchronisation).
__global__ void range_check_kernel(const unsigned* __restrict__ pairs, int npairs, int limit_log2, float* __restrict__ out) {
    __shared__ float sm[16];
    float worst = 0.f, over = 0.f, seen = 0.f;
    for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
        const unsigned mx = pairs[2 * i], inv = pairs[2 * i + 1];
        if (mx == 0u || inv == 0u) continue;
        const unsigned mn = ~inv;
        const int spread = (int)((mx >> 23) & 0xffu) - (int)((mn >> 23) & 0xffu);      // floats >= 0: exponent difference
        seen += 1.f;
        worst = fmaxf(worst, (float)spread);
        if (spread > limit_log2) over += 1.f;
    }
    worst = wave_max(worst);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = worst;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) worst = fmaxf(worst, sm[i]);
    over = block_sum(over, sm);
    seen = block_sum(seen, sm);
    if (threadIdx.x == 0) {
        out[0] = worst;
        out[1] = over;
        out[2] = seen;
    }
}

}  // namespace wsdl

using wsdl::Plan;
using wsdl::PlanOp;

extern "C" {

int wsdl_plan_begin(void) {
    WSDL_REQUIRE(wsdl::g_plan_rec == nullptr, "plan_begin: this thread is already recording a plan (one at a time per host thread)");
    wsdl::g_plan_rec = new Plan();
    wsdl::g_plans_recording.fetch_add(1);
    return WSDL_OK;
}

int wsdl_plan_recording(void) { return wsdl::g_plan_rec != nullptr; }

int wsdl_plan_end(void** plan_out) {
    Plan* p = wsdl::g_plan_rec;
    WSDL_REQUIRE(p != nullptr, "plan_end: no plan is being recorded");
    wsdl::g_plan_rec = nullptr;
    wsdl::g_plans_recording.fetch_sub(1);
    if (p->poisoned || !plan_out) {
        wsdl::set_error("plan_end: the recorded sequence cannot be replayed: %s", p->poisoned ? p->why : "no output argument");
        delete p;
        if (plan_out) *plan_out = nullptr;
        return WSDL_EINVAL;
    }
    *plan_out = p;
    return WSDL_OK;
}

int wsdl_plan_abort(void) {
    if (wsdl::g_plan_rec || wsdl::g_plan_paused) wsdl::g_plans_recording.fetch_sub(1);
    delete wsdl::g_plan_rec;
    delete wsdl::g_plan_paused;
    wsdl::g_plan_paused = nullptr;
    wsdl::g_plan_rec = nullptr;
    return WSDL_OK;
}

int wsdl_plan_mark(long long tag) {
    if (Plan* p = wsdl::g_plan_rec) {
        PlanOp op{};
        op.kind = wsdl::kMark;
        op.tag = tag;
        p->marks.push_back(p->ops.size());
        p->ops.push_back(op);
    }
    return WSDL_OK;
}

int wsdl_plan_pause(void) {
    WSDL_REQUIRE(wsdl::g_plan_rec != nullptr && wsdl::g_plan_paused == nullptr, "plan_pause: no recording to pause");
    wsdl::g_plan_paused = wsdl::g_plan_rec;
    wsdl::g_plan_rec = nullptr;
    return WSDL_OK;
}

int wsdl_plan_resume(void) {
    WSDL_REQUIRE(wsdl::g_plan_paused != nullptr && wsdl::g_plan_rec == nullptr, "plan_resume: no paused recording");
    wsdl::g_plan_rec = wsdl::g_plan_paused;
    wsdl::g_plan_paused = nullptr;
    return WSDL_OK;
}

int wsdl_plan_poison(const char* why) {
    wsdl::plan_poison(why ? why : "poisoned by the caller");
    return WSDL_OK;
}

int wsdl_plan_replay(void* plan) {
    Plan* p = static_cast<Plan*>(plan);
    WSDL_REQUIRE(p != nullptr, "plan_replay: null plan");
    WSDL_REQUIRE(wsdl::g_plan_rec == nullptr, "plan_replay: a plan is being recorded");
    return wsdl::replay_range(p, 0, p->ops.size());
}

int wsdl_plan_replay_segment(void* plan, int segment) {
    Plan* p = static_cast<Plan*>(plan);
    WSDL_REQUIRE(p != nullptr, "plan_replay_segment: null plan");
    WSDL_REQUIRE(wsdl::g_plan_rec == nullptr, "plan_replay_segment: a plan is being recorded");
    const int nseg = (int)p->marks.size() + 1;
    WSDL_REQUIRE(segment >= 0 && segment < nseg, "plan_replay_segment: segment %d of %d", segment, nseg);
    const size_t lo = segment == 0 ? 0 : p->marks[segment - 1] + 1;
    const size_t hi = segment == nseg - 1 ? p->ops.size() : p->marks[segment];
    return wsdl::replay_range(p, lo, hi);
}

int wsdl_plan_replay_timed(void* plan, double* us_by_kind, long long* n_by_kind) {
    // diagnostic twin of wsdl_plan_replay: host microseconds spent in the runtime per kind of operation
    // (0 kernel launch, 1 memset, 2 stream-waits-for-stream, 3 event record, 4 event wait, 5 mark)
    Plan* p = static_cast<Plan*>(plan);
    WSDL_REQUIRE(p != nullptr && us_by_kind && n_by_kind, "plan_replay_timed: null argument");
    WSDL_REQUIRE(wsdl::g_plan_rec == nullptr, "plan_replay_timed: a plan is being recorded");
    for (int k = 0; k < 6; ++k) us_by_kind[k] = 0.0, n_by_kind[k] = 0;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        const int rc = wsdl::replay_range(p, i, i + 1);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != WSDL_OK) return rc;
        const int k = p->ops[i].kind;
        us_by_kind[k] += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
        ++n_by_kind[k];
    }
    return WSDL_OK;
}

int wsdl_plan_stats(void* plan, long long* kernels, long long* memsets, long long* stream_waits, long long* events,
                    long long* marks) {
    Plan* p = static_cast<Plan*>(plan);
    WSDL_REQUIRE(p != nullptr, "plan_stats: null plan");
    if (kernels) *kernels = p->n_kernel;
    if (memsets) *memsets = p->n_memset;
    if (stream_waits) *stream_waits = p->n_wait;
    if (events) *events = p->n_event;
    if (marks) *marks = (long long)p->marks.size();
    return WSDL_OK;
}

long long wsdl_plan_mark_tag(void* plan, int i) {
    Plan* p = static_cast<Plan*>(plan);
    if (!p || i < 0 || i >= (int)p->marks.size()) return -1;
    return p->ops[p->marks[i]].tag;
}

int wsdl_plan_destroy(void* plan) {
    delete static_cast<Plan*>(plan);
    return WSDL_OK;
}

/* ---- stream ordering through the library (recorded into a plan) ---- */
int wsdl_event_create(void** ev) {
    WSDL_REQUIRE(ev != nullptr, "event_create: null output");
    hipEvent_t e = nullptr;
    WSDL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return WSDL_OK;
}

int wsdl_event_destroy(void* ev) {
    if (ev) WSDL_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return WSDL_OK;
}

int wsdl_event_record(void* ev, wsdl_stream_t stream) {
    WSDL_REQUIRE(ev != nullptr, "event_record: null event");
    if (Plan* p = wsdl::g_plan_rec) {
        PlanOp op{};
        op.kind = wsdl::kEventRecord;
        op.s = wsdl::as_stream(stream);
        op.ev = static_cast<hipEvent_t>(ev);
        p->ops.push_back(op);
        ++p->n_event;
    }
    WSDL_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev), wsdl::as_stream(stream)));
    return WSDL_OK;
}

int wsdl_stream_wait_event(wsdl_stream_t stream, void* ev) {
    WSDL_REQUIRE(ev != nullptr, "stream_wait_event: null event");
    if (Plan* p = wsdl::g_plan_rec) {
        PlanOp op{};
        op.kind = wsdl::kEventWait;
        op.s = wsdl::as_stream(stream);
        op.ev = static_cast<hipEvent_t>(ev);
        p->ops.push_back(op);
        ++p->n_event;
    }
    WSDL_HIP_CHECK(hipStreamWaitEvent(wsdl::as_stream(stream), static_cast<hipEvent_t>(ev), 0));
    return WSDL_OK;
}

int wsdl_stream_wait_stream(wsdl_stream_t waiter, wsdl_stream_t waited) {
    if (waiter == waited) return WSDL_OK;
    hipEvent_t e = nullptr;
    if (Plan* p = wsdl::g_plan_rec) {
        WSDL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->own_events.push_back(e);
        PlanOp op{};
        op.kind = wsdl::kStreamWait;
        op.s = wsdl::as_stream(waiter);
        op.s2 = wsdl::as_stream(waited);
        op.ev = e;
        p->ops.push_back(op);
        ++p->n_wait;
    } else {
        e = wsdl::ring_event();
        WSDL_REQUIRE(e != nullptr, "stream_wait_stream: could not create an event");
    }
    WSDL_HIP_CHECK(hipEventRecord(e, wsdl::as_stream(waited)));
    WSDL_HIP_CHECK(hipStreamWaitEvent(wsdl::as_stream(waiter), e, 0));
    return WSDL_OK;
}

int wsdl_memset_async(void* dst, int value, size_t bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(dst != nullptr || bytes == 0, "memset_async: null pointer");
    if (bytes) WSDL_HIP_CHECK(hipMemsetAsync(dst, value, bytes, wsdl::as_stream(stream)));
    return WSDL_OK;
}

int wsdl_add_int(void* p, int is64, long long delta, wsdl_stream_t stream) {
    WSDL_REQUIRE(p != nullptr, "add_int: null pointer");
    hipLaunchKernelGGL(wsdl::add_int_kernel, dim3(1), dim3(64), 0, wsdl::as_stream(stream), p, is64, delta);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_mul(const float* a, const float* b, float* out, int n, wsdl_stream_t stream) {
    WSDL_REQUIRE(a && b && out && n > 0, "mul: null pointer / empty");
    hipLaunchKernelGGL(wsdl::mul_scalars_kernel, dim3(wsdl::cdiv(n, 256)), dim3(256), 0, wsdl::as_stream(stream), a, b, out, n);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_scale_mean(const float* x, int n, float w, float* out, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && out && n > 0, "scale_mean: null pointer / empty");
    hipLaunchKernelGGL(wsdl::scale_mean_kernel, dim3(1), dim3(256), 0, wsdl::as_stream(stream), x, n, w, out);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_scale_fill(const float* g, float c, float* out, int n, wsdl_stream_t stream) {
    WSDL_REQUIRE(g && out && n > 0, "scale_fill: null pointer / empty");
    hipLaunchKernelGGL(wsdl::scale_fill_kernel, dim3(wsdl::cdiv(n, 256)), dim3(256), 0, wsdl::as_stream(stream), g, c, out, n);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_range_check(const float* pairs, int npairs, int limit_log2, float* out, wsdl_stream_t stream) {
    WSDL_REQUIRE(pairs && out && npairs > 0, "range_check: null pointer / empty");
    hipLaunchKernelGGL(wsdl::range_check_kernel, dim3(1), dim3(256), 0, wsdl::as_stream(stream),
                       reinterpret_cast<const unsigned*>(pairs), npairs, limit_log2, out);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_clamp_max_i64(const long long* x, long long* y, long long n, long long hi, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && n > 0, "clamp_max_i64: null pointer / empty");
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(wsdl::clamp_max_i64_kernel, dim3(blocks), dim3(256), 0, wsdl::as_stream(stream), x, y, n, hi);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
