// norm_pool.hip - the HBM-bound companions of the conv stack: BatchNorm2d (train statistics, backward,
// eval fold), ReLU / residual epilogues, max / average pooling, dropout, plane copies.
// All are streaming kernels: 16-byte accesses where the plane size allows, one pass per tensor.
#include "common.h"

#include <algorithm>
#include <cstdint>

wsdl::Opt wsdl::g_bn_resident{1};
wsdl::Opt wsdl::g_bn_wide_c{512};      // channel counts up to which the resident kernels run 1024 threads x 4 float4 ("bn_wide_c" option, 0 = never)

namespace {
// i / d for the (b, hw) decompositions of the BatchNorm kernels: the plane sizes of the networks are powers of two, and
// a division by a runtime divisor costs ~30 VALU instructions per element
__device__ __forceinline__ int fast_div(int i, int d, int shift) { return shift >= 0 ? i >> shift : i / d; }
__device__ __forceinline__ int pow2_shift(int d) { return (d & (d - 1)) == 0 ? __ffs(d) - 1 : -1; }
}  // namespace

namespace {

constexpr int kStatSplit = 32;  // max partial sums per channel (workspace stride); the count used is a fixed
                                // function of (C, HW), so results stay bitwise reproducible

// enough blocks to fill the chip, but not 65536 blocks of 6 KB each for C = 2048: ~2048 blocks in total,
// slices a multiple of 4 pixels so the float4 path applies
static int stat_splits(int C, int HW) {
    int s = 2048 / C;
    if (s < 1) s = 1;
    if (s > kStatSplit) s = kStatSplit;
    while (s > 1 && (HW % (4 * s)) != 0) --s;
    return s;
}

// ReLU mask as bits (relu = 3 in the backward): bit e % 8 of byte e / 8 for the dense element index e of y - written by the
// forward kernels when asked to (a residual was added, so the mask cannot be recomputed from x): the backward reads
// 1/32 of the bytes it would read from y.  A float4 at element e (a multiple of 4) owns one nibble.
__device__ __forceinline__ unsigned mask_nibble(const uint8_t* __restrict__ m, long long e) {
    return ((unsigned)m[e >> 3] >> (unsigned)(e & 4)) & 0xFu;
}
__device__ __forceinline__ void apply_nibble(float4& g, unsigned nb) {
    if (!(nb & 1u)) g.x = 0.f;
    if (!(nb & 2u)) g.y = 0.f;
    if (!(nb & 4u)) g.z = 0.f;
    if (!(nb & 8u)) g.w = 0.f;
}
// forward side: lanes 2j and 2j+1 hold the float4s at elements e and e + 4 (e a multiple of 8): the even lane stores the byte.
// Both lanes of a pair must be active (the callers' index ranges start and end on even float4 indices).
__device__ __forceinline__ void store_mask_pair(uint8_t* __restrict__ m, long long e, const float4& o) {
    const unsigned nb = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
    const unsigned hi = (unsigned)__shfl_down((int)nb, 1, 64);
    if (!(threadIdx.x & 1)) m[e >> 3] = (uint8_t)(nb | (hi << 4));
}

// ws layout for BN: double part[C][kStatSplit][2]
__global__ void bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                const float* __restrict__ y, const float* __restrict__ mean,
                                const float* __restrict__ invstd, double* __restrict__ part, int B, int C,
                                int HW, long long dy_bs, long long y_bs, int relu, int backward, int nsplit,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const uint8_t* __restrict__ rmask) {
    // forward : part = (sum x, sum x^2) ; backward: part = (sum dy', sum dy'*xhat), dy' = dy*[y>0]
    // relu == 2: the mask [y > 0] is recomputed from x (y = fma(x - mean, invstd*gamma, beta), the forward's own pinned
    // expression - no residual was added): y is not read
    __shared__ double sm[32];
    const int c = blockIdx.x, s = blockIdx.y;
    const int slice = (HW + nsplit - 1) / nsplit;
    const int r0 = s * slice;
    const int len = min(slice, HW - r0);
    double a0 = 0.0, a1 = 0.0;
    float mu = 0.f, is = 0.f;
    float mg = 0.f, mb = 0.f;
    if (backward) {
        mu = mean[c];
        is = invstd[c];
        if (relu == 2) {
            mg = is * gamma[c];
            mb = beta[c];
        }
    }
    const int total = len > 0 ? B * len : 0;
    const bool vec = ((len | r0 | HW) & 3) == 0 && ((dy_bs | y_bs) & 3) == 0;
    if (vec) {
        const int len4 = len >> 2, total4 = B * len4;
        for (int i = threadIdx.x; i < total4; i += blockDim.x) {
            const int b = i / len4, r = r0 + ((i - b * len4) << 2);
            const float4 xv = *reinterpret_cast<const float4*>(x + ((long long)b * C + c) * HW + r);
            if (!backward) {
                a0 += (double)((xv.x + xv.y) + (xv.z + xv.w));
                a1 += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
            } else {
                float4 g = *reinterpret_cast<const float4*>(dy + (long long)b * dy_bs + (long long)c * HW + r);
                if (relu == 2) {
                    if (!(fmaf(xv.x - mu, mg, mb) > 0.f)) g.x = 0.f;
                    if (!(fmaf(xv.y - mu, mg, mb) > 0.f)) g.y = 0.f;
                    if (!(fmaf(xv.z - mu, mg, mb) > 0.f)) g.z = 0.f;
                    if (!(fmaf(xv.w - mu, mg, mb) > 0.f)) g.w = 0.f;
                } else if (relu == 3) {
                    apply_nibble(g, mask_nibble(rmask, ((long long)b * C + c) * HW + r));
                } else if (relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(y + (long long)b * y_bs + (long long)c * HW + r);
                    if (!(yv.x > 0.f)) g.x = 0.f;
                    if (!(yv.y > 0.f)) g.y = 0.f;
                    if (!(yv.z > 0.f)) g.z = 0.f;
                    if (!(yv.w > 0.f)) g.w = 0.f;
                }
                a0 += (double)((g.x + g.y) + (g.z + g.w));
                a1 += (double)g.x * ((xv.x - mu) * is) + (double)g.y * ((xv.y - mu) * is) +
                      (double)g.z * ((xv.z - mu) * is) + (double)g.w * ((xv.w - mu) * is);
            }
        }
    } else
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int b = i / len, r = r0 + (i - b * len);
        const float xv = x[((long long)b * C + c) * HW + r];
        if (!backward) {
            a0 += xv;
            a1 += (double)xv * xv;
        } else {
            float g = dy[(long long)b * dy_bs + (long long)c * HW + r];
            if (relu == 2) {
                if (!(fmaf(xv - mu, mg, mb) > 0.f)) g = 0.f;
            } else if (relu == 3) {
                const long long e = ((long long)b * C + c) * HW + r;
                if (!((rmask[e >> 3] >> (e & 7)) & 1)) g = 0.f;
            } else if (relu && !(y[(long long)b * y_bs + (long long)c * HW + r] > 0.f)) g = 0.f;
            a0 += g;
            a1 += (double)g * ((xv - mu) * is);
        }
    }
    block_sum2_d(a0, a1, sm);
    if (threadIdx.x == 0) {
        part[((long long)c * kStatSplit + s) * 2 + 0] = a0;
        part[((long long)c * kStatSplit + s) * 2 + 1] = a1;
    }
}

// y = act((x-mean)*invstd*gamma + beta + res).  grid = (W, C): workgroup (w, c) owns the w-th of W equal runs of channel
// c's B*HW values (in (b, hw) order) - a few thousand values per workgroup, so that the prologue (the channel statistics
// finished from the partial sums: a dependent chain of loads and double arithmetic) is paid once per 8-16 float4 of a
// thread and not once per float4 (the former one-plane-slice-per-workgroup grid: 1.9 TB/s on the layer1 tensors).
// Workgroup (0, c) publishes save_mean / save_invstd and updates the running statistics.
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const double* __restrict__ part,
                                float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                long long n, int nsplit, const float* __restrict__ res,
                                float* __restrict__ y, int C, int HW, long long y_bs, int relu, int B,
                                float* __restrict__ amax, uint8_t* __restrict__ rmask) {
    float vmax = 0.f;
    const int c = blockIdx.y;
    double s0 = 0.0, s1 = 0.0;
    for (int s = 0; s < nsplit; ++s) {
        s0 += part[((long long)c * kStatSplit + s) * 2 + 0];
        s1 += part[((long long)c * kStatSplit + s) * 2 + 1];
    }
    const double mean_d = s0 / (double)n;
    double var = s1 / (double)n - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float mu = (float)mean_d, istd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        save_mean[c] = mu;
        save_invstd[c] = istd;
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
        if (rvar) {
            const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
    const float g = istd * gamma[c], be = beta[c];
    if ((HW & 3) == 0 && (y_bs & 3) == 0) {
        const int HW4 = HW >> 2, n4 = B * HW4, hw_sh = pow2_shift(HW4);
        int per = (n4 + (int)gridDim.x - 1) / (int)gridDim.x;
        if (rmask) per = (per + 1) & ~1;            // mask bytes are written by lane pairs: runs start on even float4s
        const int lo = min(blockIdx.x * per, n4), hi = min(lo + per, n4);
        for (int i4 = lo + threadIdx.x; i4 < hi; i4 += blockDim.x) {
            const int b = fast_div(i4, HW4, hw_sh), r = (i4 - b * HW4) << 2;
            const long long src = ((long long)b * C + c) * HW + r;
            float4 v = *reinterpret_cast<const float4*>(x + src);
            v.x = fmaf(v.x - mu, g, be); v.y = fmaf(v.y - mu, g, be);      // pinned: the backward recomputes the ReLU mask
            v.z = fmaf(v.z - mu, g, be); v.w = fmaf(v.w - mu, g, be);
            if (res) {
                const float4 q = *reinterpret_cast<const float4*>(res + src);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(y + (long long)b * y_bs + (long long)c * HW + r) = v;
            vmax = amax4(vmax, v);
            if (rmask) store_mask_pair(rmask, src, v);
        }
    } else {
        const int nn = B * HW;
        const int per = (nn + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lo = blockIdx.x * per, hi = min(lo + per, nn);
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const int b = i / HW, r = i - b * HW;
            const long long src = ((long long)b * C + c) * HW + r;
            float v = fmaf(x[src] - mu, g, be);
            if (res) v += res[src];
            if (relu) v = fmaxf(v, 0.f);
            y[(long long)b * y_bs + (long long)c * HW + r] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (amax) publish_amax(vmax, amax);
}

// dx = gamma*invstd*(dy' - k0 - xhat*k1) ; dres = dy'
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ y, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const double* __restrict__ part, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate, long long n, int nsplit,
                                    float* __restrict__ dx,
                                    float* __restrict__ dres, int C, int HW, long long dy_bs, long long y_bs,
                                    int relu, int B, float* __restrict__ amax, const float* __restrict__ beta,
                                    const uint8_t* __restrict__ rmask) {
    // grid = (W, C), as bn_apply_kernel: one prologue per run of a few thousand values of channel c
    float vmax = 0.f;
    const int c = blockIdx.y;
    // finish the channel's two sums (sum dy', sum dy'*xhat) from the partials; one block publishes the parameter
    // gradients (same arithmetic as the former finalize kernel: fixed order, double)
    double s0 = 0.0, s1 = 0.0;
    for (int s = 0; s < nsplit; ++s) {
        s0 += part[((long long)c * kStatSplit + s) * 2 + 0];
        s1 += part[((long long)c * kStatSplit + s) * 2 + 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s1 : (float)s1;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s0 : (float)s0;
    }
    const float k0 = (float)(s0 / (double)n), k1 = (float)(s1 / (double)n);
    const float mu = mean[c], is = invstd[c], gi = gamma[c] * is;
    const float mg = is * gamma[c], mb = relu == 2 ? beta[c] : 0.f;          // relu == 2: mask recomputed from x
    if ((HW & 3) == 0 && ((dy_bs | y_bs) & 3) == 0) {
        const int HW4 = HW >> 2, n4 = B * HW4, hw_sh = pow2_shift(HW4);
        const int per = (n4 + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lo = blockIdx.x * per, hi = min(lo + per, n4);
        for (int i4 = lo + threadIdx.x; i4 < hi; i4 += blockDim.x) {
            const int b = fast_div(i4, HW4, hw_sh), r = (i4 - b * HW4) << 2;
            const long long src = ((long long)b * C + c) * HW + r;
            float4 g = *reinterpret_cast<const float4*>(dy + (long long)b * dy_bs + (long long)c * HW + r);
            const float4 xv = *reinterpret_cast<const float4*>(x + src);
            if (relu == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(y + (long long)b * y_bs + (long long)c * HW + r);
                if (!(yv.x > 0.f)) g.x = 0.f;
                if (!(yv.y > 0.f)) g.y = 0.f;
                if (!(yv.z > 0.f)) g.z = 0.f;
                if (!(yv.w > 0.f)) g.w = 0.f;
            } else if (relu == 3) {
                apply_nibble(g, mask_nibble(rmask, src));
            } else if (relu == 2) {
                if (!(fmaf(xv.x - mu, mg, mb) > 0.f)) g.x = 0.f;
                if (!(fmaf(xv.y - mu, mg, mb) > 0.f)) g.y = 0.f;
                if (!(fmaf(xv.z - mu, mg, mb) > 0.f)) g.z = 0.f;
                if (!(fmaf(xv.w - mu, mg, mb) > 0.f)) g.w = 0.f;
            }
            float4 o;
            o.x = gi * (g.x - k0 - (xv.x - mu) * is * k1);
            o.y = gi * (g.y - k0 - (xv.y - mu) * is * k1);
            o.z = gi * (g.z - k0 - (xv.z - mu) * is * k1);
            o.w = gi * (g.w - k0 - (xv.w - mu) * is * k1);
            *reinterpret_cast<float4*>(dx + src) = o;
            vmax = amax4(vmax, o);
            if (dres) *reinterpret_cast<float4*>(dres + src) = g;
        }
    } else {
        const int nn = B * HW;
        const int per = (nn + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lo = blockIdx.x * per, hi = min(lo + per, nn);
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const int b = i / HW, r = i - b * HW;
            const long long src = ((long long)b * C + c) * HW + r;
            float g = dy[(long long)b * dy_bs + (long long)c * HW + r];
            if (relu == 1 && !(y[(long long)b * y_bs + (long long)c * HW + r] > 0.f)) g = 0.f;
            if (relu == 2 && !(fmaf(x[src] - mu, mg, mb) > 0.f)) g = 0.f;
            if (relu == 3 && !((rmask[src >> 3] >> (src & 7)) & 1)) g = 0.f;
            const float xh = (x[src] - mu) * is;
            const float o = gi * (g - k0 - xh * k1);
            dx[src] = o;
            vmax = fmaxf(vmax, fabsf(o));
            if (dres) dres[src] = g;
        }
    }
    if (amax) publish_amax(vmax, amax);
}

// ---------------------------------------------------------------------------------------------
// Channel-resident BatchNorm: when one channel's B*HW values fit in the registers of one workgroup (64 per thread,
// 256 or 512 threads: layer2-4 and the heads at B=16, 256^2), the workgroup of channel c reads them ONCE, reduces
// the statistics, and applies the normalisation (forward) / the input gradient (backward) from registers - the
// second pass over x (forward) and over x, dy and the ReLU mask (backward) of the two-kernel form never happens.
// Same arithmetic: double sums, fixed reduction tree (bitwise reproducible).  grid = C.
// Several workgroups per channel (round 5; "bn_coop").  A 64-channel layer is 64 workgroups on 256 CUs (16.8 us at 2.0 TB/s for a
// tensor the chip moves in 7); with S workgroups per channel each holds 1/S of the channel in registers, publishes its partial
// sums, and waits for its siblings': partials and the arrival counter are single coherent (agent-scope) accesses - no fence,
// which would write a whole L2 back - and every workgroup adds the S partials in the SAME order (bitwise reproducible).
// Siblings are neighbours in the grid, so they are dispatched together; the wait is bounded (a sibling that never arrives
// poisons the channel with NaN after ~1 s instead of hanging the queue).  The last workgroup to leave re-arms the channel's
// two counters for the stream's next launch (`cnt`: zero-initialised by the caller, 2 ints per channel, one region per stream).
__device__ __forceinline__ void coop_exchange(double& a0, double& a1, unsigned long long* __restrict__ part, int* __restrict__ cnt,
                                              int c, int s, int S) {
    unsigned long long* mine = part + ((size_t)c * S + s) * 2;
    // Exchanges, not stores: a returning read-modify-write is performed where every XCD sees it, and its RETURN is the proof
    // that it has been (the acknowledgement of a plain store only says the local L2 took it) - the arrival is counted after both
    const unsigned long long r0 = __hip_atomic_exchange(mine, __builtin_bit_cast(unsigned long long, a0), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long r1 = __hip_atomic_exchange(mine + 1, __builtin_bit_cast(unsigned long long, a1), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(r0), "v"(r1) : "memory");
    __hip_atomic_fetch_add(cnt + 2 * c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(cnt + 2 * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S && spins < (1 << 21)) {
        __builtin_amdgcn_s_sleep(16);
        ++spins;
    }
    double t0 = 0.0, t1 = 0.0;
    for (int j = 0; j < S; ++j) {
        const unsigned long long* q = part + ((size_t)c * S + j) * 2;
        t0 += __builtin_bit_cast(double, __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        t1 += __builtin_bit_cast(double, __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (spins >= (1 << 21)) t0 = t1 = __builtin_nan("");
    a0 = t0;
    a1 = t1;
}
__device__ __forceinline__ void coop_depart(int* __restrict__ cnt, int c, int S) {
    if (__hip_atomic_fetch_add(cnt + 2 * c + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1) {
        const int z0 = __hip_atomic_exchange(cnt + 2 * c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int z1 = __hip_atomic_exchange(cnt + 2 * c + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(z0), "v"(z1) : "memory");
    }
}

template <int NT, int V = 16>
__global__ __launch_bounds__(NT) void bn_fwd_resident_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ rmean, float* __restrict__ rvar,
    float momentum, float eps, const float* __restrict__ res, float* __restrict__ y, int B, int C, int HW,
    long long y_bs, int relu, float* __restrict__ amax, uint8_t* __restrict__ rmask, float* __restrict__ cmin,
    int S, unsigned long long* __restrict__ part, int* __restrict__ cnt, float* __restrict__ chan_amax) {
    // V float4 per thread: 16 with 256 / 512 threads; 4 with 1024 threads for the 256- and 512-channel layers (one
    // workgroup per CU at most: sixteen waves keep four times the requests of four waves moving - see resident_threads)
    // S > 1: workgroup blockIdx.x = c * S + s holds slice s of channel c (coop_exchange above); S = 1: the whole channel
    __shared__ double sm[32];
    __shared__ float bc[2];
    const int c = S > 1 ? blockIdx.x / S : blockIdx.x, sl = S > 1 ? blockIdx.x - c * S : 0, tid = threadIdx.x;
    const int HW4 = HW >> 2, hw_sh = pow2_shift(HW4);
    const int n4 = S > 1 ? B * HW4 / S : B * HW4, lo4 = sl * n4;      // this workgroup's float4s: [lo4, lo4 + n4)
    float4 v[V];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i4 = tid + k * NT;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < n4) {
            const int b = fast_div(lo4 + i4, HW4, hw_sh), r = (lo4 + i4 - b * HW4) << 2;
            v[k] = *reinterpret_cast<const float4*>(x + ((long long)b * C + c) * HW + r);
            a0 += (double)((v[k].x + v[k].y) + (v[k].z + v[k].w));
            a1 += (double)v[k].x * v[k].x + (double)v[k].y * v[k].y + (double)v[k].z * v[k].z + (double)v[k].w * v[k].w;
        }
    }
    block_sum2_d(a0, a1, sm);
    if (tid == 0) {
        if (S > 1) coop_exchange(a0, a1, part, cnt, c, sl, S);
        const long long n = (long long)B * HW;
        const double mean = a0 / (double)n;
        double var = a1 / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mu = (float)mean, istd = (float)(1.0 / sqrt(var + (double)eps));
        if (sl == 0) {
            save_mean[c] = mu;
            save_invstd[c] = istd;
            if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
            if (rvar) {
                const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
            }
        }
        bc[0] = mu;
        bc[1] = istd;
        if (S > 1) coop_depart(cnt, c, S);
    }
    __syncthreads();
    const float mu = bc[0], g = bc[1] * gamma[c], be = beta[c];
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i4 = tid + k * NT;
        if (i4 < n4) {
            const int b = fast_div(lo4 + i4, HW4, hw_sh), r = (lo4 + i4 - b * HW4) << 2;
            float4 o = v[k];
            o.x = fmaf(o.x - mu, g, be); o.y = fmaf(o.y - mu, g, be);
            o.z = fmaf(o.z - mu, g, be); o.w = fmaf(o.w - mu, g, be);
            if (res) {
                const float4 q = *reinterpret_cast<const float4*>(res + ((long long)b * C + c) * HW + r);
                o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
            }
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(y + (long long)b * y_bs + (long long)c * HW + r) = o;
            vmax = amax4(vmax, o);
            if (rmask) store_mask_pair(rmask, ((long long)b * C + c) * HW + r, o);
        }
    }
    if (amax) publish_amax_min(vmax, amax, cmin, (chan_amax && S == 1) ? chan_amax + c : nullptr);
}

template <int NT, int V = 16>
__global__ __launch_bounds__(NT) void bn_bwd_resident_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, float* __restrict__ dx,
    float* __restrict__ dres, int B, int C, int HW, long long dy_bs, long long y_bs, int relu,
    float* __restrict__ amax, const float* __restrict__ beta, const uint8_t* __restrict__ rmask, float* __restrict__ cmin,
    int S, unsigned long long* __restrict__ part, int* __restrict__ cnt, float* __restrict__ chan_amax,
    unsigned char* __restrict__ presplit) {
    __shared__ double sm[32];
    __shared__ float bc[2];
    const int c = S > 1 ? blockIdx.x / S : blockIdx.x, sl = S > 1 ? blockIdx.x - c * S : 0, tid = threadIdx.x;
    const int HW4 = HW >> 2, hw_sh = pow2_shift(HW4);
    const int n4 = S > 1 ? B * HW4 / S : B * HW4, lo4 = sl * n4;      // (S > 1: slice sl of the channel - see bn_fwd_resident_kernel)
    const float mu = mean[c], is = invstd[c];
    const float mg = is * gamma[c], mb = relu == 2 ? beta[c] : 0.f;          // relu == 2: mask recomputed from x
    float4 g[V], xh[V];                            // dy' = dy*[y>0] and xhat, kept for the second phase
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i4 = tid + k * NT;
        g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        xh[k] = g[k];
        if (i4 < n4) {
            const int b = fast_div(lo4 + i4, HW4, hw_sh), r = (lo4 + i4 - b * HW4) << 2;
            const float4 xv = *reinterpret_cast<const float4*>(x + ((long long)b * C + c) * HW + r);
            float4 gv = *reinterpret_cast<const float4*>(dy + (long long)b * dy_bs + (long long)c * HW + r);
            if (relu == 2) {
                if (!(fmaf(xv.x - mu, mg, mb) > 0.f)) gv.x = 0.f;
                if (!(fmaf(xv.y - mu, mg, mb) > 0.f)) gv.y = 0.f;
                if (!(fmaf(xv.z - mu, mg, mb) > 0.f)) gv.z = 0.f;
                if (!(fmaf(xv.w - mu, mg, mb) > 0.f)) gv.w = 0.f;
            } else if (relu == 3) {
                apply_nibble(gv, mask_nibble(rmask, ((long long)b * C + c) * HW + r));
            } else if (relu) {
                const float4 yv = *reinterpret_cast<const float4*>(y + (long long)b * y_bs + (long long)c * HW + r);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            g[k] = gv;
            xh[k] = make_float4((xv.x - mu) * is, (xv.y - mu) * is, (xv.z - mu) * is, (xv.w - mu) * is);
            a0 += (double)((gv.x + gv.y) + (gv.z + gv.w));
            a1 += (double)gv.x * xh[k].x + (double)gv.y * xh[k].y + (double)gv.z * xh[k].z + (double)gv.w * xh[k].w;
        }
    }
    block_sum2_d(a0, a1, sm);
    if (tid == 0) {
        if (S > 1) coop_exchange(a0, a1, part, cnt, c, sl, S);
        const long long n = (long long)B * HW;
        if (sl == 0) {
            if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)a1 : (float)a1;
            if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)a0 : (float)a0;
        }
        bc[0] = (float)(a0 / (double)n);
        bc[1] = (float)(a1 / (double)n);
        if (S > 1) coop_depart(cnt, c, S);
    }
    __syncthreads();
    const float k0 = bc[0], k1 = bc[1], gi = gamma[c] * is;
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i4 = tid + k * NT;
        if (i4 < n4) {
            const int b = fast_div(lo4 + i4, HW4, hw_sh), r = (lo4 + i4 - b * HW4) << 2;
            const long long o = ((long long)b * C + c) * HW + r;
            float4 d;
            d.x = gi * (g[k].x - k0 - xh[k].x * k1);
            d.y = gi * (g[k].y - k0 - xh[k].y * k1);
            d.z = gi * (g[k].z - k0 - xh[k].z * k1);
            d.w = gi * (g[k].w - k0 - xh[k].w * k1);
            *reinterpret_cast<float4*>(dx + o) = d;
            vmax = amax4(vmax, d);
            if (dres) *reinterpret_cast<float4*>(dres + o) = g[k];
            if (presplit) g[k] = d;                      // kept for the pre-split rows below (g is dead otherwise)
        }
    }
    if (presplit) {
        // dx is the dY operand of the producing convolution's WEIGHT GRADIENT (conv_wgrad_split16*_kernel), which wants it as
        // fp16 (high, low) pieces in 128-byte rows [pixel / 32][channel] - a pass of its own until round 5 (dy_split16_kernel:
        // one more read and write of dY per layer, 24 launches per training step).  This workgroup holds the whole channel in
        // registers: it takes the channel's maximum (one more workgroup-wide reduction), derives the channel's power-of-two scale
        // - per CHANNEL, which a separate pass could only afford per tensor without reading dY twice - and writes the rows.
        // A thread's float4 is half of a 16-byte unit (8 pixels): lane pairs fill units, a wave 8 whole rows.
        __shared__ float smx[17];
        const float cm = block_max_all(vmax, smx);
        int e_;
        const float sc = pow2_scale(cm, e_);
        const unsigned sw = (unsigned)((c >> 1) & 7);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int i4 = tid + k * NT;
            if (i4 < n4) {
                const int pix = (lo4 + i4) << 2;             // linear pixel (b, h, w) of the float4's first element
                const unsigned kg = (unsigned)((pix & 31) >> 3), half = (unsigned)((pix >> 2) & 1);
                unsigned h0, l0, h1, l1;
                split2h(g[k].x * sc, g[k].y * sc, h0, l0);
                split2h(g[k].z * sc, g[k].w * sc, h1, l1);
                unsigned char* row = presplit + ((long long)(pix >> 5) * C + c) * 128 + half * 8;
                *reinterpret_cast<u32x2*>(row + ((kg ^ sw) << 4)) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(row + (((4u + kg) ^ sw) << 4)) = u32x2{l0, l1};
            }
        }
    }
    if (amax) publish_amax_min(vmax, amax, cmin, (chan_amax && S == 1) ? chan_amax + c : nullptr);
}

// threads of the channel-resident form for (C, n = B*HW values per channel), 0 = use the two-kernel form
static int resident_threads(int C, long long n, int HW, bool backward = false) {
    // g_bn_resident: 0 off, 1 = from 64 channels, n > 1 = from n channels.  (Round 2 started at 192: fewer workgroups leave
    // most CUs idle - but the two-pass form's second read costs more: 128 x 1024 forward 13.9 -> 9.9 us, backward 15.3 -> 10.2;
    // 64 x 4096 forward 22.0 -> 16.8 in the 1024 x 16 form; profiles/r03_notes.md)
    const int min_c = wsdl::g_bn_resident > 1 ? wsdl::g_bn_resident : 64;
    if (!wsdl::g_bn_resident || C < min_c || (HW & 3) != 0) return 0;
    // 256 / 512 channels = one or two workgroups per CU.  Sixteen waves of a quarter of the work each instead of four:
    // forward 14.5 -> 11.6 us (256 channels) and 21.5 -> 16.6 us (512) on the 16 x 32 x 32 maps; backward 16.2 -> 15.0 us at 256
    // channels but 22.0 -> 23.3 us at 512 (three tensors in flight per thread already): wide up to 512 / 256 channels
    // (tools/bn_bench.py, profiles/r03_notes.md).  Round 6 tried the backward of the > 256-channel layers as 512 x 8 and 1024 x 4
    // float4 once more: 2048 x 1024 114.6 -> 126.9 / 121.2 us, the step 864.1 / 865.6 -> 856.7 / 856.6 and 861.6 / 860.2 img/s - removed again.
    if (n <= 1024 * 16 && C <= (backward ? wsdl::g_bn_wide_c / 2 : wsdl::g_bn_wide_c)) return 1024;
    if (n <= 256 * 64) return 256;
    if (n <= 512 * 64) return 512;
    if (!backward && n <= 1024 * 64) return 1025;      // forward only: 1024 threads x 16 float4 (the 64 x 64 maps at B = 16)
    return 0;
}

// slices per channel of the cooperative form (1: not used): few-channel layers only, whole float4 slices that fit 1024 x 4
static int coop_slices(int C, int B, int HW) {
    if (!wsdl::g_bn_coop || C > wsdl::g_bn_coop || (HW & 3) != 0) return 1;
    int S = 1;
    while (C * S < 256 && S < 4) S *= 2;
    if (S == 1 && wsdl::g_bn_coop_wide) S = 2;               // 256 channels: two workgroups each (512 of half the work)
    const long long n4 = (long long)B * HW / 4;
    while (S > 1 && (n4 % S != 0 || n4 / S > 1024 * 4)) S /= 2;
    return S;
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                               float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

__global__ void affine_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                      const float* __restrict__ scale, float* __restrict__ dconv,
                                      float* __restrict__ dres, int C, int HW, int relu, int planes,
                                      float* __restrict__ amax) {
  float vmax = 0.f;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int c = plane % C;
    const float sc = scale ? scale[c] : 1.f;
    const long long base = (long long)plane * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float g = dy[base + i];
        if (relu && !(y[base + i] > 0.f)) g = 0.f;
        if (dconv) dconv[base + i] = g * sc;
        if (dres) dres[base + i] = g;
        vmax = fmaxf(vmax, fabsf(g * sc));
    }
  }
  if (amax) publish_amax(vmax, amax);
}

// The same for small planes (CAM path: 14 x 14 maps, 16384 planes per batch of 8 - one 196-element plane per workgroup made
// the launch 30 us for 38 MB): a flat sweep in float4, the channel recovered from the element index.  HW % 4 == 0.
__global__ void affine_act_bwd_flat_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                           const float* __restrict__ scale, float4* __restrict__ dconv,
                                           float4* __restrict__ dres, int C, int HW4, long long total4, int relu,
                                           float* __restrict__ amax) {
    float vmax = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW4) % C);
        const float sc = scale ? scale[c] : 1.f;
        float4 g = dy[i];
        if (relu) {
            const float4 yv = y[i];
            if (!(yv.x > 0.f)) g.x = 0.f;
            if (!(yv.y > 0.f)) g.y = 0.f;
            if (!(yv.z > 0.f)) g.z = 0.f;
            if (!(yv.w > 0.f)) g.w = 0.f;
        }
        if (dres) dres[i] = g;
        g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
        if (dconv) dconv[i] = g;
        vmax = amax4(vmax, g);
    }
    if (amax) publish_amax(vmax, amax);
}

// y = act(scale[c]*x + shift[c]): a stand-alone eval-mode BatchNorm2d (folded running statistics) or a stand-alone ReLU
// (scale = shift = null).  The models never run these - their BatchNorm / ReLU are fused behind the convolution -
// but a drop-in user may call model.backbone.bn1(x) or hook a ReLU module.
__global__ void affine_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float* __restrict__ y, int C, int HW, int relu,
                                      int planes) {
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int c = plane % C;
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    const long long base = (long long)plane * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = x[base + i] * sc + sh;
        if (relu) v = fmaxf(v, 0.f);
        y[base + i] = v;
    }
  }
}

// ---- MaxPool2d(3, 2, 1)
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   uint8_t* __restrict__ am, int H, int W, int OH, int OW, int planes) {
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const float* xp = x + (long long)plane * H * W;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < OH * OW; o += gridDim.x * blockDim.x) {
        const int oh = o / OW, ow = o - oh * OW;
        float best = -INFINITY;
        int bi = 0;
        bool any = false;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ih = oh * 2 - 1 + i, iw = ow * 2 - 1 + j;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                const float v = xp[ih * W + iw];
                if (!any || v > best || v != v) {   // first max wins; NaN propagates (ATen semantics)
                    best = v;
                    bi = i * 3 + j;
                    any = true;
                }
            }
        y[(long long)plane * OH * OW + o] = best;
        if (am) am[(long long)plane * OH * OW + o] = (uint8_t)bi;
    }
  }
}

// gather form: every input pixel checks the (at most 4) windows that contain it.  One thread: four consecutive pixels
// of a row (W % 4 == 0): the windows of columns ow = iw0/2 .. iw0/2 + 2 and rows oh = ih/2, (ih+1)/2 cover them -
// six (argmax, dy) pairs for four outputs and one 16-byte store (94 -> 40 us on the stem's 16 x 64 x 128 x 128).
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                   float* __restrict__ dx, int H, int W, int OH, int OW, int planes) {
  const int W4 = W >> 2, per = H * W4;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const float* gp = dy + (long long)plane * OH * OW;
    const uint8_t* ap = am + (long long)plane * OH * OW;
    if ((W & 3) == 0) {
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += gridDim.x * blockDim.x) {
            const int ih = idx / W4, iw0 = (idx - ih * W4) << 2;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
                const int i = ih - (oh * 2 - 1);           // row inside the window: 0..2 by construction
                if (oh >= OH) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int ow = iw0 / 2 + k;
                    if (ow >= OW) continue;
                    const int a = ap[oh * OW + ow];
                    const float g = gp[oh * OW + ow];
                    if (a / 3 != i) continue;
                    const int e = (ow * 2 - 1) + (a - 3 * i) - iw0;     // the window's argmax column, relative to iw0
                    if (e >= 0 && e < 4) o[e] += g;
                }
            }
            *reinterpret_cast<float4*>(dx + (long long)plane * H * W + ih * W + iw0) = make_float4(o[0], o[1], o[2], o[3]);
        }
        continue;
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * W; idx += gridDim.x * blockDim.x) {
        const int ih = idx / W, iw = idx - ih * W;
        float s = 0.f;
        // windows oh with oh*2-1 <= ih <= oh*2+1  ->  oh in [(ih-1+1)/2 .. (ih+1)/2]
        for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
            if (oh >= OH) continue;
            const int i = ih - (oh * 2 - 1);
            if (i < 0 || i > 2) continue;
            for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow >= OW) continue;
                const int j = iw - (ow * 2 - 1);
                if (j < 0 || j > 2) continue;
                if (ap[oh * OW + ow] == i * 3 + j) s += gp[oh * OW + ow];
            }
        }
        dx[(long long)plane * H * W + idx] = s;
    }
  }
}

__global__ void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW) {
    __shared__ float sm[16];
    const float* xp = x + (long long)blockIdx.x * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += xp[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) y[blockIdx.x] = s / (float)HW;
}

// one WAVE per plane, four planes per workgroup, 16-byte loads (round 6: ASPP's pooling branch reduces 32768 planes of 1024
// values - one 256-thread workgroup with two barriers per plane took 58 us for 134 MB)
__global__ void gap_fwd_wave_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int planes) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float4* xp = reinterpret_cast<const float4*>(x + (long long)plane * HW);
    float s = 0.f;
    for (int i = lane; i < (HW >> 2); i += 64) {
        const float4 v = xp[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = wave_sum(s);
    if (lane == 0) y[plane] = s / (float)HW;
}

__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int HW, int accumulate,
                               int planes) {
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const float g = dy[plane] / (float)HW;
    float* dp = dx + (long long)plane * HW;
    if ((HW & 3) == 0) {
        float4* dp4 = reinterpret_cast<float4*>(dp);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += gridDim.x * blockDim.x) {
            float4 v = make_float4(g, g, g, g);
            if (accumulate) {
                const float4 q = dp4[i];
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            dp4[i] = v;
        }
        continue;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x)
        dp[i] = accumulate ? dp[i] + g : g;
  }
}

// ---- dropout: counter-based hash (splitmix64 of seed + element index) -> uniform [0,1)
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   uint8_t* __restrict__ mask, size_t n, float p, float inv_keep,
                                   unsigned long long seed, int gen, const unsigned long long* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;      // per-call counter kept on the device (hipGraph replay)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint8_t m;
        if (gen) {
            m = hash_uniform(seed, i) >= p ? 1 : 0;
            mask[i] = m;
        } else {
            m = mask[i];
        }
        y[i] = m ? x[i] * inv_keep : 0.f;
    }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                   float* __restrict__ dx, size_t n, float inv_keep) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = mask[i] ? dy[i] * inv_keep : 0.f;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                           size_t n, int relu) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = a[i] + b[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                 float* __restrict__ y, size_t n) {
    const float k = s[0];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = x[i] * k;
}

__global__ void copy_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                   long long src_bs, long long dst_bs, int planes) {
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int b = plane / C, c = plane - b * C;
    const float* sp = src + (long long)b * src_bs + (long long)c * HW;
    float* dp = dst + (long long)b * dst_bs + (long long)c * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) dp[i] = sp[i];
  }
}

inline dim3 plane_grid(int planes, int HW, int per_thread = 1) {
    int gx = wsdl::cdiv(HW, 256 * per_thread);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    return dim3(gx, planes > 65535 ? 65535 : planes);
}
// grid (W, C) of the two-pass BatchNorm apply kernels: about 2048 workgroups, at least 1024 float4 per workgroup
inline dim3 channel_grid(int B, int C, int HW) {
    const long long n4 = (long long)B * HW / 4;
    long long w = (2048 + C - 1) / C;
    if (w > n4 / 1024) w = n4 / 1024;
    if (w < 1) w = 1;
    return dim3((unsigned)w, (unsigned)C);
}
inline int flat_blocks(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 8192); }

}  // namespace

extern "C" {

int wsdl_bn_channel_resident(int B, int C, int HW, int backward) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return resident_threads(C, (long long)B * HW, HW, backward != 0) != 0 && coop_slices(C, B, HW) == 1;
}

size_t wsdl_bn_workspace(int C) {
    return C > 0 ? (size_t)C * kStatSplit * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float) : 0;
}

int wsdl_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                      float* save_invstd, float* running_mean, float* running_var, float momentum,
                      float eps, int B, int C, int HW, const float* residual, int relu, long long y_bs,
                      float* y_amax, uint8_t* relu_mask, void* ws, size_t ws_bytes, int* coop, float* chan_amax,
                      wsdl_stream_t stream) {
    WSDL_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && ws, "bn_train_fwd: null pointer");
    WSDL_REQUIRE(!relu_mask || (relu && (HW & 7) == 0), "bn_train_fwd: the bit mask needs relu and HW %% 8 == 0");
    WSDL_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0 && (long long)B * HW < (1ll << 31), "bn_train_fwd: bad shape");
    WSDL_REQUIRE((long long)B * HW > 1, "bn_train_fwd: needs more than one value per channel (as torch)");
    if (ws_bytes < wsdl_bn_workspace(C)) {
        wsdl::set_error("bn_train_fwd: workspace too small");
        return WSDL_EWORKSPACE;
    }
    if (!y_bs) y_bs = (long long)C * HW;
    WSDL_REQUIRE(!relu_mask || (y_bs & 3) == 0, "bn_train_fwd: the bit mask needs a 16-byte aligned batch stride");
    hipStream_t s = wsdl::as_stream(stream);
    float* cmin = (wsdl::g_range_sentinel && y_amax) ? y_amax + 1 : nullptr;      // "range_sentinel": y_amax is a (max, ~min) pair
    WSDL_REQUIRE(!chan_amax || (y_amax && wsdl_bn_channel_resident(B, C, HW, 0) && (y_bs & 3) == 0 && (!coop || coop_slices(C, B, HW) == 1)),
                 "bn_train_fwd: chan_amax needs the channel-resident kernel (wsdl_bn_channel_resident) and y_amax");
    if (const int S = (coop && (y_bs & 3) == 0) ? coop_slices(C, B, HW) : 1; S > 1) {
        // several workgroups per channel (coop_exchange): partial sums in the workspace, the channel's counters in `coop`
        // (re-armed here as well as by the last workgroup out: a launch whose wait ran into its bound - two processes sharing the
        // GPU - must not leave counters behind that poison every later launch on the stream)
        WSDL_HIP_CHECK(hipMemsetAsync(coop, 0, (size_t)2 * C * sizeof(int), s));
        hipLaunchKernelGGL((bn_fwd_resident_kernel<1024, 4>), dim3(C * S), dim3(1024), 0, s, x, gamma, beta, save_mean,
                           save_invstd, running_mean, running_var, momentum, eps, residual, y, B, C, HW, y_bs, relu,
                           y_amax, relu_mask, cmin, S, static_cast<unsigned long long*>(ws), coop, (float*)nullptr);
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    if (const int nt = ((y_bs & 3) == 0) ? resident_threads(C, (long long)B * HW, HW) : 0) {
        if (nt == 1025)
            hipLaunchKernelGGL((bn_fwd_resident_kernel<1024, 16>), dim3(C), dim3(1024), 0, s, x, gamma, beta, save_mean,
                               save_invstd, running_mean, running_var, momentum, eps, residual, y, B, C, HW, y_bs, relu,
                               y_amax, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax);
        else if (nt == 1024)
            hipLaunchKernelGGL((bn_fwd_resident_kernel<1024, 4>), dim3(C), dim3(1024), 0, s, x, gamma, beta, save_mean,
                               save_invstd, running_mean, running_var, momentum, eps, residual, y, B, C, HW, y_bs, relu,
                               y_amax, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax);
        else if (nt == 256)
            hipLaunchKernelGGL((bn_fwd_resident_kernel<256>), dim3(C), dim3(256), 0, s, x, gamma, beta, save_mean,
                               save_invstd, running_mean, running_var, momentum, eps, residual, y, B, C, HW, y_bs, relu,
                               y_amax, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax);
        else
            hipLaunchKernelGGL((bn_fwd_resident_kernel<512>), dim3(C), dim3(512), 0, s, x, gamma, beta, save_mean,
                               save_invstd, running_mean, running_var, momentum, eps, residual, y, B, C, HW, y_bs, relu,
                               y_amax, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax);
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    double* part = static_cast<double*>(ws);
    const int ns = stat_splits(C, HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, s, x, nullptr, nullptr, nullptr,
                       nullptr, part, B, C, HW, 0ll, 0ll, 0, 0, ns, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(bn_apply_kernel, channel_grid(B, C, HW), dim3(256), 0, s, x, gamma, beta, part, save_mean,
                       save_invstd, running_mean, running_var, momentum, eps, (long long)B * HW, ns, residual, y, C,
                       HW, y_bs, relu, B, y_amax, (y_bs & 3) == 0 ? relu_mask : nullptr);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_bn_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                      float* dbeta, float* dres, int B, int C, int HW, int relu,
                      int accumulate_param_grads, long long dy_bs, long long y_bs, float* dx_amax,
                      const uint8_t* relu_mask, void* ws, size_t ws_bytes, int* coop, float* chan_amax, void* dy_presplit,
                      wsdl_stream_t stream) {
    WSDL_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && ws, "bn_train_bwd: null pointer");
    WSDL_REQUIRE(relu != 1 || y, "bn_train_bwd: relu = 1 takes the mask from the forward output y");
    WSDL_REQUIRE(relu != 2 || beta, "bn_train_bwd: relu = 2 recomputes the mask from x and needs beta");
    WSDL_REQUIRE(relu != 3 || (relu_mask && (HW & 7) == 0 && (dy_bs & 3) == 0),
                 "bn_train_bwd: relu = 3 takes the mask bits written by wsdl_bn_train_fwd (HW %% 8 == 0)");
    WSDL_REQUIRE(relu >= 0 && relu <= 3, "bn_train_bwd: relu must be 0, 1, 2 or 3");
    WSDL_REQUIRE(B > 0 && C > 0 && C <= 65535 && HW > 0 && (long long)B * HW < (1ll << 31), "bn_train_bwd: bad shape");
    if (ws_bytes < wsdl_bn_workspace(C)) {
        wsdl::set_error("bn_train_bwd: workspace too small");
        return WSDL_EWORKSPACE;
    }
    if (!dy_bs) dy_bs = (long long)C * HW;
    if (!y_bs) y_bs = (long long)C * HW;
    hipStream_t s = wsdl::as_stream(stream);
    float* cmin = (wsdl::g_range_sentinel && dx_amax) ? dx_amax + 1 : nullptr;
    WSDL_REQUIRE(!(chan_amax || dy_presplit) || (dx_amax && wsdl_bn_channel_resident(B, C, HW, 1) && ((dy_bs | y_bs) & 3) == 0 &&
                                                 (!coop || coop_slices(C, B, HW) == 1)),
                 "bn_train_bwd: chan_amax / dy_presplit need the channel-resident kernel (wsdl_bn_channel_resident) and dx_amax");
    WSDL_REQUIRE(!dy_presplit || (chan_amax && ((long long)B * HW) % 32 == 0),
                 "bn_train_bwd: dy_presplit needs chan_amax (its scales) and B * HW a multiple of 32");
    if (const int S = (coop && ((dy_bs | y_bs) & 3) == 0) ? coop_slices(C, B, HW) : 1; S > 1) {
        WSDL_HIP_CHECK(hipMemsetAsync(coop, 0, (size_t)2 * C * sizeof(int), s));      // (see the forward)
        hipLaunchKernelGGL((bn_bwd_resident_kernel<1024, 4>), dim3(C * S), dim3(1024), 0, s, x, dy, y, gamma, save_mean,
                           save_invstd, dgamma, dbeta, accumulate_param_grads, dx, dres, B, C, HW, dy_bs, y_bs, relu,
                           dx_amax, beta, relu_mask, cmin, S, static_cast<unsigned long long*>(ws), coop, (float*)nullptr, (unsigned char*)nullptr);
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    if (const int nt = (((dy_bs | y_bs) & 3) == 0) ? resident_threads(C, (long long)B * HW, HW, true) : 0) {
        if (nt == 1024)
            hipLaunchKernelGGL((bn_bwd_resident_kernel<1024, 4>), dim3(C), dim3(1024), 0, s, x, dy, y, gamma, save_mean,
                               save_invstd, dgamma, dbeta, accumulate_param_grads, dx, dres, B, C, HW, dy_bs, y_bs, relu,
                               dx_amax, beta, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax, static_cast<unsigned char*>(dy_presplit));
        else if (nt == 256)
            hipLaunchKernelGGL((bn_bwd_resident_kernel<256>), dim3(C), dim3(256), 0, s, x, dy, y, gamma, save_mean,
                               save_invstd, dgamma, dbeta, accumulate_param_grads, dx, dres, B, C, HW, dy_bs, y_bs, relu,
                               dx_amax, beta, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax, static_cast<unsigned char*>(dy_presplit));
        else
            hipLaunchKernelGGL((bn_bwd_resident_kernel<512>), dim3(C), dim3(512), 0, s, x, dy, y, gamma, save_mean,
                               save_invstd, dgamma, dbeta, accumulate_param_grads, dx, dres, B, C, HW, dy_bs, y_bs, relu,
                               dx_amax, beta, relu_mask, cmin, 1, (unsigned long long*)nullptr, (int*)nullptr, chan_amax, static_cast<unsigned char*>(dy_presplit));
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    double* part = static_cast<double*>(ws);
    const int ns = stat_splits(C, HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns), dim3(256), 0, s, x, dy, y, save_mean, save_invstd,
                       part, B, C, HW, dy_bs, y_bs, relu, 1, ns, gamma, beta, relu_mask);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, channel_grid(B, C, HW), dim3(256), 0, s, x, dy, y, gamma, save_mean,
                       save_invstd, part, dgamma, dbeta, accumulate_param_grads, (long long)B * HW, ns, dx, dres, C, HW,
                       dy_bs, y_bs, relu, B, dx_amax, beta, relu_mask);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* scale, float* shift, int C,
                 wsdl_stream_t stream) {
    WSDL_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(wsdl::cdiv(C, 256)), dim3(256), 0, wsdl::as_stream(stream), gamma,
                       beta, running_mean, running_var, eps, scale, shift, C);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_affine_act_bwd(const float* dy, const float* y, const float* scale, float* dconv, float* dres,
                        int B, int C, int HW, int relu, float* dconv_amax, wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && (dconv || dres) && B > 0 && C > 0 && HW > 0, "affine_act_bwd: bad arguments");
    WSDL_REQUIRE(!relu || y, "affine_act_bwd: relu mask needs y");
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (HW % 4 == 0 && HW < 2048 && al16(dy) && al16(y) && al16(dconv) && al16(dres)) {
        const long long total4 = (long long)B * C * HW / 4;
        hipLaunchKernelGGL(affine_act_bwd_flat_kernel, dim3((unsigned)std::min<long long>((total4 + 255) / 256, 2048)), dim3(256), 0,
                           wsdl::as_stream(stream), reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), scale,
                           reinterpret_cast<float4*>(dconv), reinterpret_cast<float4*>(dres), C, HW / 4, total4, relu, dconv_amax);
    } else
        hipLaunchKernelGGL(affine_act_bwd_kernel, plane_grid(B * C, HW), dim3(256), 0, wsdl::as_stream(stream), dy, y,
                           scale, dconv, dres, C, HW, relu, B * C, dconv_amax);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                        int relu, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "affine_act_fwd: bad arguments");
    hipLaunchKernelGGL(affine_act_fwd_kernel, plane_grid(B * C, HW), dim3(256), 0, wsdl::as_stream(stream), x, scale,
                       shift, y, C, HW, relu, B * C);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int BC, int H, int W,
                          wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && BC > 0 && H > 0 && W > 0 , "maxpool_fwd: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, plane_grid(BC, OH * OW), dim3(256), 0, wsdl::as_stream(stream), x, y,
                       argmax, H, W, OH, OW, BC);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int BC, int H, int W,
                          wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && argmax && dx && BC > 0 && H > 0 && W > 0, "maxpool_bwd: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, plane_grid(BC, H * W, 4), dim3(256), 0, wsdl::as_stream(stream), dy,
                       argmax, dx, H, W, OH, OW, BC);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_global_avgpool_fwd(const float* x, float* y, int BC, int HW, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && BC > 0 && HW > 0, "global_avgpool_fwd: bad arguments");
    if (HW % 256 == 0 && BC >= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(gap_fwd_wave_kernel, dim3((BC + 3) / 4), dim3(256), 0, wsdl::as_stream(stream), x, y, HW, BC);
    else
        hipLaunchKernelGGL(gap_fwd_kernel, dim3(BC), dim3(256), 0, wsdl::as_stream(stream), x, y, HW);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_global_avgpool_bwd(const float* dy, float* dx, int BC, int HW, int accumulate,
                            wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && dx && BC > 0 && HW > 0, "global_avgpool_bwd: bad arguments");
    hipLaunchKernelGGL(gap_bwd_kernel, plane_grid(BC, HW, 4), dim3(256), 0, wsdl::as_stream(stream), dy, dx, HW,
                       accumulate, BC);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_dropout_fwd(const float* x, float* y, uint8_t* mask, size_t n, float p, unsigned long long seed,
                     int gen_mask, const unsigned long long* seed_dev, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && mask && n > 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad arguments");
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(flat_blocks(n)), dim3(256), 0, wsdl::as_stream(stream), x, y,
                       mask, n, p, 1.f / (1.f - p), seed, gen_mask, seed_dev);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, size_t n, float p,
                     wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && dx && mask && n > 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad arguments");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(flat_blocks(n)), dim3(256), 0, wsdl::as_stream(stream), dy, mask,
                       dx, n, 1.f / (1.f - p));
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_add(const float* a, const float* b, float* y, size_t n, int relu, wsdl_stream_t stream) {
    WSDL_REQUIRE(a && b && y && n > 0, "add: bad arguments");
    hipLaunchKernelGGL(add_kernel, dim3(flat_blocks(n)), dim3(256), 0, wsdl::as_stream(stream), a, b, y, n, relu);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_scale_by_device_scalar(const float* x, const float* s, float* y, size_t n, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && s && y && n > 0, "scale_by_device_scalar: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(flat_blocks(n)), dim3(256), 0, wsdl::as_stream(stream), x, s, y, n);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_copy_planes(const float* src, float* dst, int B, int C, int HW, long long src_bs,
                     long long dst_bs, wsdl_stream_t stream) {
    WSDL_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0, "copy_planes: bad arguments");
    if (!src_bs) src_bs = (long long)C * HW;
    if (!dst_bs) dst_bs = (long long)C * HW;
    hipLaunchKernelGGL(copy_planes_kernel, plane_grid(B * C, HW), dim3(256), 0, wsdl::as_stream(stream), src, dst,
                       C, HW, src_bs, dst_bs, B * C);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
