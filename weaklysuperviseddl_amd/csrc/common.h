// common.h - shared host-side helpers for libwsdl_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <memory>
#include <tuple>
#include <utility>

#include "../../include/wsdl_hip.h"

namespace wsdl {

void set_error(const char* fmt, ...);

#define WSDL_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::wsdl::set_error(__VA_ARGS__);     \
            return WSDL_EINVAL;                 \
        }                                       \
    } while (0)

#define WSDL_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            ::wsdl::set_error("%s failed: %s", #expr, hipGetErrorString(e_));         \
            return WSDL_EHIP;                                                         \
        }                                                                             \
    } while (0)

#define WSDL_LAUNCH_CHECK()                                                           \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ::wsdl::set_error("kernel launch failed: %s", hipGetErrorString(e_));     \
            return WSDL_EHIP;                                                         \
        }                                                                             \
    } while (0)

static inline hipStream_t as_stream(wsdl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Launch trace (wsdl_launch_trace / wsdl_last_launches): when on, the convolution entry points describe every launch
// they choose - kernel form, tile, K slices, XCD order, column bands, grid - into a per-thread string the caller reads back.
// Off (the default) it is one relaxed load per launch site.  What tests/test_hip_conv_fullsize.py uses to show WHICH kernel
// configuration a full-size comparison against the float64 oracle went through.
extern std::atomic<int> g_trace_launches;
void trace_launch(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#define WSDL_TRACE(...)                                                                \
    do {                                                                               \
        if (__builtin_expect(::wsdl::g_trace_launches.load(std::memory_order_relaxed), 0)) ::wsdl::trace_launch(__VA_ARGS__); \
    } while (0)

// Profiling scope: records a start/stop event pair on the stream when wsdl_prof_enable(1) is set.
bool prof_enabled();
struct ProfScope {
    int cls;
    hipStream_t s;
    void* slot;
    ProfScope(int cls, hipStream_t s, double work, double work_executed = -1.0, double bytes = 0.0);
    ~ProfScope();
};

// A process-wide option (wsdl_set_option): an int that any host thread may read while another one sets it - relaxed atomic
// accesses (plain moves on x86), so a call sees the old or the new value, never a torn one.  wsdl_set_option itself is
// serialised and refuses to run while ANY thread records a launch plan (a plan freezes the tile choices made under the
// option set it was recorded with).  What an option means for cached weight LAYOUTS is the caller's to track: changing
// "conv_split" / "conv_arith" requires wsdl_conv2d_prep_weights again (the Python host keys its caches on a layout epoch).
struct Opt {
    std::atomic<int> v;
    explicit constexpr Opt(int x) : v(x) {}
    operator int() const { return v.load(std::memory_order_relaxed); }
    Opt& operator=(int x) { v.store(x, std::memory_order_relaxed); return *this; }
    Opt(const Opt&) = delete;
};
extern std::atomic<int> g_plans_recording;     // host threads between wsdl_plan_begin and wsdl_plan_end / _abort

// A/B switch (wsdl_set_option "bn_resident"): channel-resident fused BatchNorm kernels (norm_pool.hip)
extern Opt g_bn_resident;
extern Opt g_bn_wide_c;     // "bn_wide_c": resident BatchNorm kernels with 1024 threads up to this channel count
extern Opt g_layercam_tail_mod;   // "layercam_tail_mod": see layercam_optim.hip
extern Opt g_bn_coop;             // "bn_coop": several workgroups per channel in the resident BatchNorm kernels up to this channel count (0 off)
extern Opt g_bn_coop_wide;        // "bn_coop_wide": ... and two per channel at 256 channels
extern Opt g_range_sentinel;      // "range_sentinel": the amax pointers of the BatchNorm entry points are (max, ~min piece max) PAIRS

// deterministic two-stage sum: stage 1 kernels write `n` float partials, stage 2 adds them in order.
constexpr int kReduceSlots = 4096;

// ---- launch plans (plan.hip; include/wsdl_hip.h "launch plans") ---------------------------------------------------
// Every kernel launch and stream-ordered memset of the library goes through the two functions below.  Outside a
// recording they are the plain runtime calls.  Between wsdl_plan_begin and wsdl_plan_end each launch is ALSO appended
// to the plan being recorded - function address, grid, block, dynamic LDS, stream and a private copy of the argument
// values - so that wsdl_plan_replay can issue the same sequence again from one C loop, without the host logic of the
// entry points (tile choice, workspace carving, option look-ups) and without the caller's per-call overhead.
struct Plan;
// the plan THIS host thread is recording (one per thread), else nullptr.  Thread-local: a second host thread that calls into
// the library meanwhile (a loader worker, a second model) launches normally and is not written into the first thread's plan;
// it may record a plan of its own.
extern thread_local Plan* g_plan_rec;
void plan_add_kernel(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, std::shared_ptr<void> storage,
                     void* const* argv, int argc);
void plan_poison(const char* why);  // an entry point that cannot be replayed (library code launching kernels of its own)
hipError_t memset_async(void* dst, int value, size_t bytes, hipStream_t s);

template <typename... K, typename... A, size_t... I>
inline void launch_recorded(void (*kernel)(K...), dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                            std::index_sequence<I...>, A&&... a) {
    using Tup = std::tuple<std::remove_cv_t<std::remove_reference_t<K>>...>;
    auto heap = std::make_shared<Tup>(static_cast<K>(std::forward<A>(a))...);
    void* argv[] = {static_cast<void*>(&std::get<I>(*heap))...};
    plan_add_kernel(reinterpret_cast<const void*>(kernel), grid, block, shmem, s, heap, argv, (int)sizeof...(K));
    (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, shmem, s);
}

template <typename... K, typename... A>
inline void launch(void (*kernel)(K...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, A&&... a) {
    static_assert(sizeof...(K) == sizeof...(A), "kernel launch: argument count does not match the kernel's parameters");
    if (__builtin_expect(g_plan_rec != nullptr, 0))
        launch_recorded(kernel, grid, block, shmem, s, std::index_sequence_for<K...>{}, std::forward<A>(a)...);
    else
        kernel<<<grid, block, shmem, s>>>(std::forward<A>(a)...);
}

}  // namespace wsdl

// every launch site of the library is written as hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
    ::wsdl::launch(kernelName, dim3(numBlocks), dim3(numThreads), (size_t)(memPerBlock), (streamId), __VA_ARGS__)
#define hipMemsetAsync(dst, value, bytes, stream) ::wsdl::memset_async((dst), (value), (bytes), (stream))

// ------------------------------------------------------------------ device helpers
// Barrier for LDS hand-offs inside K loops.  __syncthreads() carries a workgroup-scope fence that is address-space
// agnostic, so the compiler drains vmcnt(0) in front of every s_barrier: global loads issued for LATER chunks are
// waited for at each barrier and a register prefetch ring degenerates to one exposed memory round trip per chunk.
// An LDS hand-off only needs the DS queue drained.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// max|v| of what a kernel stored, into a zero-initialised device scalar (the fp16x2 convolutions scale their operands by
// a power of two derived from it).  One atomic per WORKGROUP at most, and only when the workgroup's maximum beats the
// value already there (a relaxed device-scope read; a stale - smaller - value only costs an atomic): thousands of
// atomics on one address serialise (a per-wave version made the two-pass BatchNorm kernels 2x slower).  The bit patterns
// of non-negative floats order like the floats.  Every thread of the workgroup must call it.
__device__ __forceinline__ void publish_amax(float m, float* __restrict__ amax) {
    __shared__ float s_amax[16];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x * blockDim.y + 63) >> 6;
        for (int i = 1; i < nw; ++i) m = fmaxf(m, s_amax[i]);
        if (m > 0.f) {
            unsigned* a = reinterpret_cast<unsigned*>(amax);
            const unsigned bits = __builtin_bit_cast(unsigned, m);
            if (bits > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, bits);
        }
    }
}
// Range sentinel (wsdl_set_option "range_sentinel"): besides max|v| of the tensor, the SMALLEST non-zero maximum any piece of
// it has - a piece being what one workgroup stores: a CHANNEL in the channel-resident BatchNorm kernels.  (A finer piece - the
// 256 values a wave stores per pass - cost 2.3 % of the training step in cross-lane maxima inside the store loops; measured and
// dropped.)  The fp16x2 convolutions scale a tensor by ONE power of two: a region 2^E below the tensor's maximum is
// computed to 2^-(38-E) of its own maximum, past 1e-3 from E ~ 29 (conv_split.h).  wsdl_range_check turns the pairs
// (max, min piece maximum) of a step into "spread exceeded 2^25 somewhere" without a host synchronisation.  The minimum is
// kept as the bitwise complement of its float bits, so that the zero a slot starts from means "none yet" and atomicMax
// orders it.  Every thread of the workgroup must call it.
// `chan` (optional): this workgroup's own maximum as a plain store - the per-CHANNEL maximum of the channel-resident BatchNorm
// kernels, which the weight-gradient kernels use as per-channel scales (exact: a channel is a row / column of that GEMM's output).
__device__ __forceinline__ void publish_amax_min(float m, float* __restrict__ amax, float* __restrict__ cmin,
                                                 float* __restrict__ chan = nullptr) {
    __shared__ float s_amax2[16];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_amax2[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x * blockDim.y + 63) >> 6;
        for (int i = 1; i < nw; ++i) m = fmaxf(m, s_amax2[i]);
        if (chan) *chan = m;
        if (m > 0.f) {
            unsigned* a = reinterpret_cast<unsigned*>(amax);
            const unsigned bits = __builtin_bit_cast(unsigned, m);
            if (cmin && (blockIdx.x & 7) == 0) {
                // Every EIGHTH workgroup (channel) is a candidate for the smallest maximum: the workgroups of a launch's first
                // wave all find an empty slot and all fire their atomics at one cache line - with every channel taking part
                // that cost 0.9 % of the training step, sampled it is not measurable; a tensor whose channels are graded is
                // still seen (>= 8 candidates from 64 channels on).  The pair is one aligned 8-byte word: ONE relaxed load
                // filters both atomics (a second round trip to memory at the tail of every workgroup cost another 0.6 %).
                const unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long*>(amax), __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                if (bits > (unsigned)cur) atomicMax(a, bits);
                if (~bits > (unsigned)(cur >> 32)) atomicMax(a + 1, ~bits);      // this channel's maximum: a candidate for the smallest
            } else if (bits > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicMax(a, bits);
            }
        }
    }
}
// ---- the fp16x2 split, shared by the convolution kernels (conv_split.h) and by the producers that write pre-split operands
// (bn_bwd_resident_kernel: the weight gradient's dY rows)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// (x0, x1), already scaled -> packed fp16 pairs of the two pieces (round to nearest even; x - h is exact in fp32)
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& h, unsigned& l) {
    f32x2 v = {x0, x1};
    const half2v hv = __builtin_convertvector(v, half2v);
    h = __builtin_bit_cast(unsigned, hv);
    f32x2 r = {x0 - (float)hv[0], x1 - (float)hv[1]};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
}

// power-of-two scale that puts amax in [2^14, 2^15); 1 for amax = 0 / inf / NaN.  `e` returns its exponent.
__device__ __forceinline__ float pow2_scale(float amax, int& e) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax) & 0x7fffffffu;
    const int be = (int)(bits >> 23);                   // biased exponent
    e = (bits == 0u || be == 255) ? 0 : 14 - (be - 127);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
__device__ __forceinline__ float pow2(int e) {          // |e| <= 200: two exact factors
    const int e1 = e / 2, e2 = e - e1;
    return __builtin_bit_cast(float, (unsigned)(e1 + 127) << 23) * __builtin_bit_cast(float, (unsigned)(e2 + 127) << 23);
}

// max over the workgroup, returned to EVERY thread (blockDim.x <= 1024, a multiple of 64)
__device__ __forceinline__ float block_max_all(float m, float* smem /* >= 17 floats */) {
    m = wave_max(m);
    __syncthreads();                                   // (smem may still be read from a previous use)
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) m = fmaxf(m, smem[i]);
        smem[16] = m;
    }
    __syncthreads();
    return smem[16];
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
// two block-wide sums behind ONE pair of barriers (the channel-resident BatchNorm kernels reduce two statistics per launch);
// each sum in block_sum_d's order: same bits
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* smem /* >= 32 doubles */) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) {
        smem[wid] = a;
        smem[16 + wid] = b;
    }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) {
            ra += smem[i];
            rb += smem[16 + i];
        }
    a = ra;
    b = rb;
}
__device__ __forceinline__ double block_sum_d(double v, double* smem /* >= 16 doubles */) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
