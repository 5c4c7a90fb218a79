// layercam_optim.hip - LayerCAM epilogue (channel-weighted sum / ReLU / min-max / bilinear / layer
// mean / clamp-pow / threshold) and the fused flat Adam step.
//
// The pseudo-mask is an INTEGER output (cam >= thresh), so the epilogue reproduces the arithmetic of the reference's
// PyTorch-CPU path operation by operation AND in its order of summation: on identical activations / gradients the CAM
// is bit-identical to `LayerCAMGenerator.generate` (TraditionalModel/LayerCAM.py:52-76) run on torch CPU, and so is
// every mask pixel.  What that order is (ATen's SumKernel.cpp `cascade_sum`, outer reduction over the channel
// dimension of a contiguous NCHW tensor; checked against torch 2.10 bit for bit by tests/test_hip_ops.py):
//   * pixels p < hw - hw % kTailMod ("main"; ATen's vectorised columns): four-level cascade with level step 16 -
//     acc0 sums 16 consecutive channels starting from 0, every 16 channels acc1 += acc0, every 256 acc2 += acc1, every
//     4096 acc3 += acc2; channels beyond the last full 16 stay in acc0; result ((acc0 + acc1) + acc2) + acc3;
//   * the last hw % kTailMod pixels (ATen's scalar columns, `row_sum`): four interleaved streams (channel i -> stream
//     i % 4), each the same cascade over its C / 4 entries, then stream0 += the C % 4 left-over channels, += stream 1, 2, 3.
//   kTailMod = 32 is two AVX-512 vectors of floats ("layercam_tail_mod" option: 16 for an AVX2 build of torch, 0 = the
//   cascade for every pixel).  The level step is 16 for C <= 2^19 channels (max(4, ceil_log2(C) / 4) in ATen).
//   relu(g * a) is a rounded product, min / max are order-free, `c -= min; c /= (max + 1e-8)` are IEEE operations,
//   bilinear interpolation (align_corners=False) is ATen's  s = fma(scale, o + 0.5, -0.5),  t0 * w0 + t1 * w1  evaluated
//   as fma(t0, w0, round(t1 * w1)) along x and then along y (UpSampleKernel.cpp's generic kernel as compiled for x86 with
//   FMA - the one ATen takes for outH + outW > 128, e.g. the reference's fixed 224 x 224; smaller outputs go through its
//   four-weight kernel, which this one matches to an ulp or two only), the layer
//   mean is a sum in layer order and an IEEE division, `** alpha` is the identity / x*x / x*x*x for alpha = 1 / 2 / 3 as
//   torch special-cases them: bit-identical.  alpha = 0.5 is torch.sqrt, which an MKL build of torch evaluates with VML's
//   vsSqrt - not correctly rounded (0.6 % of values differ from IEEE sqrt by one ulp): here it is the IEEE sqrt, within one
//   ulp of the reference; other exponents: powf, within a few ulp of torch's Sleef pow.  alpha = 1 is the default and what
//   every call site of the reference passes (PsuedoMasks.py:27, Abalations.py:88).
//   The whole file is compiled with floating-point contraction OFF (the __f*_rn intrinsics of this hipcc are plain
//   operators, and __fsqrt_rn is the approximate native square root): a fused multiply-add appears only where written.
//
// Three launches for all images and layers together:
//   1. layercam_partial : grid (64-channel slices, layers, B).  Level-0 sums of relu(g*a): per main pixel one value per
//      16 channels, per (tail pixel, stream) one value per 64 channels; consecutive lanes = consecutive pixels of one
//      channel -> coalesced reads of the NCHW activations/gradients (the 4.8 MB/img that dominate).
//   2. layercam_normalise: grid (layers, B).  The upper cascade levels in order, ReLU, per-image min-max
//      (second min-max after **alpha for the notebook variant) with wave shuffles.
//   3. layercam_upsample : bilinear to (outH,outW), mean over layers, clamp/pow, threshold -> uint8.
#include "common.h"

#include <algorithm>
#include <cmath>

#pragma clang fp contract(off)

wsdl::Opt wsdl::g_layercam_tail_mod{32};   // "layercam_tail_mod" option (wsdl_set_option): see above

namespace {

constexpr int kMaxLayers = 8;
constexpr int kL0 = 16;          // level step of the cascade (channels per level-0 sum)
constexpr int kSliceC = 64;      // channels per workgroup of the partial kernel = 4 level-0 blocks = one block of each stream
constexpr int kMaxTail = 32;     // tail pixels per map are < kTailMod <= 32

struct CamLayers {
    const float* act[kMaxLayers];
    const float* grad[kMaxLayers];
    int C[kMaxLayers], h[kMaxLayers], w[kMaxLayers];
    int tail0[kMaxLayers];            // first tail pixel (hw when there is none)
    long long part_off[kMaxLayers];   // float offset of this layer's [B][ceil(C/16)][hw] level-0 sums (main pixels)
    long long tpart_off[kMaxLayers];  // float offset of this layer's [B][ceil(C/64)][tail pixels][4 streams] level-0 sums
    long long map_off[kMaxLayers];    // float offset of this layer's [B][hw] normalised map
    int n;
};

// single IEEE operations (contraction is off in this file: the compiler may not fuse them)
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_div(float a, float b) { return a / b; }

__device__ __forceinline__ float relu_prod(float g, float a) { return fmaxf(rn_mul(g, a), 0.f); }

__global__ void layercam_partial_kernel(CamLayers L, float* __restrict__ ws) {
    const int s = blockIdx.x, l = blockIdx.y, b = blockIdx.z;
    const int C = L.C[l], hw = L.h[l] * L.w[l];
    const int c0 = s * kSliceC;
    if (c0 >= C) return;
    const int tail0 = L.tail0[l], tn = hw - tail0;
    const int nblk = (C + kL0 - 1) / kL0, ngrp = (C + kSliceC - 1) / kSliceC;
    const float* a = L.act[l] + (long long)b * C * hw;
    const float* g = L.grad[l] + (long long)b * C * hw;
    float* part = ws + L.part_off[l] + (long long)b * nblk * hw;
    float* tpart = ws + L.tpart_off[l] + ((long long)b * ngrp + s) * tn * 4;
    const int nstream = C / 4;          // entries per stream of a tail pixel (channels >= 4 * nstream are added later)
    for (int it = threadIdx.x; it < tail0 + 4 * tn; it += blockDim.x) {
        if (it < tail0) {
            const int p = it;
#pragma unroll
            for (int q = 0; q < kSliceC / kL0; ++q) {
                const int cb = c0 + q * kL0;
                if (cb >= C) break;
                const int cnt = min(kL0, C - cb);
                float acc = 0.f;
                if (cnt == kL0) {
#pragma unroll
                    for (int c = 0; c < kL0; ++c)
                        acc = rn_add(acc, relu_prod(g[(long long)(cb + c) * hw + p], a[(long long)(cb + c) * hw + p]));
                } else {
                    for (int c = 0; c < cnt; ++c)
                        acc = rn_add(acc, relu_prod(g[(long long)(cb + c) * hw + p], a[(long long)(cb + c) * hw + p]));
                }
                part[(long long)(cb / kL0) * hw + p] = acc;
            }
        } else {
            const int t = (it - tail0) >> 2, k = (it - tail0) & 3, p = tail0 + t;
            float acc = 0.f;
            for (int m = 0; m < kL0; ++m) {
                const int e = s * kL0 + m;      // entry of stream k
                if (e >= nstream) break;
                const long long c = 4LL * e + k;
                acc = rn_add(acc, relu_prod(g[c * hw + p], a[c * hw + p]));
            }
            tpart[t * 4 + k] = acc;
        }
    }
}

// upper levels of ATen's cascade over `nfull` level-0 sums read at `stride`, plus the left-over sum (`has_left`).  Level 1 takes 16
// level-0 sums and is flushed into level 2 exactly every 16 of them (i = 256, 512, ...), so the loop runs over whole level-1
// blocks: 16 independent loads in flight, then the 16 ordered additions (a load per addition made the kernel latency-bound:
// 35 us for sixteen workgroups).
__device__ __forceinline__ float cascade_upper(const float* __restrict__ v, long long stride, int nfull, bool has_left) {
    float acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int q = 0;
    for (; q + kL0 <= nfull; q += kL0) {
        float t[kL0];
#pragma unroll
        for (int j = 0; j < kL0; ++j) t[j] = v[(long long)(q + j) * stride];
#pragma unroll
        for (int j = 0; j < kL0; ++j) acc1 = rn_add(acc1, t[j]);
        acc2 = rn_add(acc2, acc1);                 // i = 16 (q + 16) is a multiple of 256
        acc1 = 0.f;
        if ((((q + kL0) * kL0) & 0xF00) == 0) {    // ... and of 4096
            acc3 = rn_add(acc3, acc2);
            acc2 = 0.f;
        }
    }
    for (; q < nfull; ++q) acc1 = rn_add(acc1, v[(long long)q * stride]);      // < 16 sums: no flush falls in here
    const float acc0 = has_left ? v[(long long)nfull * stride] : 0.f;
    return rn_add(rn_add(rn_add(acc0, acc1), acc2), acc3);
}

// torch's `** alpha` on a non-negative fp32 value (Pow.cpp / PowKernel.cpp special cases)
__device__ __forceinline__ float pow_like_torch(float v, float alpha) {
    if (alpha == 1.f) return v;
    if (alpha == 0.5f) return sqrtf(v);     // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
    if (alpha == 2.f) return rn_mul(v, v);
    if (alpha == 3.f) return rn_mul(rn_mul(v, v), v);
    if (alpha == 0.f) return 1.f;
    return powf(v, alpha);
}

__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* sm) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sm[wid] = mn;
        sm[16 + wid] = mx;
    }
    __syncthreads();
    mn = sm[0];
    mx = sm[16];
    for (int i = 1; i < nw; ++i) {
        mn = fminf(mn, sm[i]);
        mx = fmaxf(mx, sm[16 + i]);
    }
}

__global__ void layercam_normalise_kernel(CamLayers L, float* __restrict__ ws, float alpha, int variant) {
    __shared__ float sm[32];
    __shared__ float s_tail[kMaxTail * 4];
    const int l = blockIdx.x, b = blockIdx.y;
    const int C = L.C[l], hw = L.h[l] * L.w[l];
    const int tail0 = L.tail0[l], tn = hw - tail0;
    const int nblk = (C + kL0 - 1) / kL0, ngrp = (C + kSliceC - 1) / kSliceC;
    const float* part = ws + L.part_off[l] + (long long)b * nblk * hw;
    const float* tpart = ws + L.tpart_off[l] + (long long)b * ngrp * tn * 4;
    float* map = ws + L.map_off[l] + (long long)b * hw;
    // pass 1: the cascade's upper levels + relu (kept in the map buffer)
    const int nstream = C / 4;
    for (int it = threadIdx.x; it < tail0 + 4 * tn; it += blockDim.x) {
        if (it < tail0) {
            map[it] = fmaxf(cascade_upper(part + it, hw, C / kL0, (C % kL0) != 0), 0.f);
        } else {
            const int j = it - tail0;   // = 4 * t + k
            s_tail[j] = cascade_upper(tpart + j, (long long)tn * 4, nstream / kL0, (nstream % kL0) != 0);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < tn) {
        const int t = threadIdx.x, p = tail0 + t;
        const float* a = L.act[l] + (long long)b * C * hw;
        const float* g = L.grad[l] + (long long)b * C * hw;
        float v = s_tail[4 * t];
        for (int c = 4 * nstream; c < C; ++c) v = rn_add(v, relu_prod(g[(long long)c * hw + p], a[(long long)c * hw + p]));
        v = rn_add(rn_add(rn_add(v, s_tail[4 * t + 1]), s_tail[4 * t + 2]), s_tail[4 * t + 3]);
        map[p] = fmaxf(v, 0.f);
    }
    __syncthreads();
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        const float v = map[p];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, sm);
    // c -= min ; c /= (max_after_shift + 1e-8)      (rounding is monotonic: max(c - min) = fl(max - min))
    const float den = rn_add(rn_sub(mx, mn), 1e-8f);
    float mn2 = INFINITY, mx2 = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        float v = rn_div(rn_sub(map[p], mn), den);
        if (variant == 1) v = pow_like_torch(v, alpha);
        map[p] = v;
        mn2 = fminf(mn2, v);
        mx2 = fmaxf(mx2, v);
    }
    if (variant == 1) {
        block_minmax(mn2, mx2, sm);
        const float den2 = rn_add(rn_sub(mx2, mn2), 1e-8f);
        for (int p = threadIdx.x; p < hw; p += blockDim.x) map[p] = rn_div(rn_sub(map[p], mn2), den2);
    }
}

// y = minmax(relu(x)) per plane: c -= min; c /= (max + 1e-8)   (classic CAM, one block per (image, class) plane)
__global__ void plane_relu_minmax_kernel(const float* __restrict__ x, float* __restrict__ y, int hw) {
    __shared__ float sm[32];
    const float* xp = x + (long long)blockIdx.x * hw;
    float* yp = y + (long long)blockIdx.x * hw;
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        const float v = fmaxf(xp[p], 0.f);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, sm);
    const float den = (mx - mn) + 1e-8f;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) yp[p] = (fmaxf(xp[p], 0.f) - mn) / den;
}

// ATen's area_pixel_compute_source_index + guard_index_and_lambda (align_corners=False), as compiled with FMA contraction
__device__ __forceinline__ void src_index(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = __builtin_fmaf(scale, rn_add((float)o, 0.5f), -0.5f);
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = fminf(fmaxf(rn_sub(s, (float)i0), 0.f), 1.f);
    l0 = rn_sub(1.f, l1);
}
// t0 * w0 + t1 * w1 as the reference's CPU build evaluates it
__device__ __forceinline__ float lerp_like_torch(float t0, float w0, float t1, float w1) {
    return __builtin_fmaf(t0, w0, rn_mul(t1, w1));
}

__global__ void layercam_upsample_kernel(CamLayers L, const float* __restrict__ ws, float* __restrict__ cam,
                                         uint8_t* __restrict__ mask, int outH, int outW, float alpha,
                                         int variant, float thresh) {
    const int b = blockIdx.y;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < outH * outW; o += gridDim.x * blockDim.x) {
        const int oh = o / outW, ow = o - oh * outW;
        float sum = 0.f;
        for (int l = 0; l < L.n; ++l) {
            const int h = L.h[l], w = L.w[l];
            const float* m = ws + L.map_off[l] + (long long)b * h * w;
            float v;
            if (h == outH && w == outW) {
                v = m[oh * w + ow];     // F.interpolate to the same size copies
            } else {
                int y0, y1, x0, x1;
                float ly0, ly1, lx0, lx1;
                src_index(oh, rn_div((float)h, (float)outH), h, y0, y1, ly0, ly1);
                src_index(ow, rn_div((float)w, (float)outW), w, x0, x1, lx0, lx1);
                const float top = lerp_like_torch(m[y0 * w + x0], lx0, m[y0 * w + x1], lx1);
                const float bot = lerp_like_torch(m[y1 * w + x0], lx0, m[y1 * w + x1], lx1);
                v = lerp_like_torch(top, ly0, bot, ly1);
            }
            sum = rn_add(sum, v);
        }
        float v = rn_div(sum, (float)L.n);
        if (variant == 0) v = pow_like_torch(fmaxf(v, 0.f), alpha);
        cam[(long long)b * outH * outW + o] = v;
        if (mask) mask[(long long)b * outH * outW + o] = (v >= thresh && v > 0.f) ? 1 : 0;
    }
}

// torch.optim.Adam (no amsgrad, no weight decay): exp_avg, exp_avg_sq, bias corrections as torch
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float b1, float b2, float eps, float step_size,
                            float sqrt_bc2, float gscale, const int* __restrict__ step_dev, float lr,
                            const float* __restrict__ hyper) {
    if (hyper) {            // lr, beta1, beta2, eps, grad_scale on the device: a schedule changes them without a new launch plan
        lr = hyper[0];
        b1 = hyper[1];
        b2 = hyper[2];
        eps = hyper[3];
        gscale = hyper[4];
    }
    if (step_dev) {
        // the step number lives on the device (hipGraph replay: the host value is frozen at capture time); same
        // double-precision bias corrections as the host path
        const int st = *step_dev;
        const double bc1 = 1.0 - pow((double)b1, (double)st), bc2 = 1.0 - pow((double)b2, (double)st);
        step_size = (float)((double)lr / bc1);
        sqrt_bc2 = (float)sqrt(bc2);
    }
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gscale;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv) / sqrt_bc2 + eps;
        pp -= step_size * (mm / denom);
    };
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = p4[i], mv = m4[i], vv = v4[i];
        const float4 gv = g4[i];
        upd(pv.x, gv.x, mv.x, vv.x);
        upd(pv.y, gv.y, mv.y, vv.y);
        upd(pv.z, gv.z, mv.z, vv.z);
        upd(pv.w, gv.w, mv.w, vv.w);
        p4[i] = pv;
        m4[i] = mv;
        v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        upd(p[i], g[i], m[i], v[i]);
    }
}

int fill_layers(CamLayers& L, const float* const* act, const float* const* grad, const int* C, const int* h,
                const int* w, int n_layers, int B, size_t* total_floats) {
    WSDL_REQUIRE(n_layers >= 1 && n_layers <= kMaxLayers, "layercam: 1..%d layers supported", kMaxLayers);
    WSDL_REQUIRE(B >= 1 && B <= 65535, "layercam: bad batch");
    const int tail_mod = wsdl::g_layercam_tail_mod;
    WSDL_REQUIRE(tail_mod >= 0 && tail_mod <= kMaxTail, "layercam: layercam_tail_mod must be 0..%d", kMaxTail);
    size_t off = 0;
    L.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        // C <= 2^19: the reference's cascade has level step 16 up to there (and so has each of the four streams)
        WSDL_REQUIRE(C[l] > 0 && C[l] <= (1 << 19) && h[l] > 0 && w[l] > 0 && (long long)h[l] * w[l] < (1 << 24),
                     "layercam: bad layer %d shape", l);
        L.act[l] = act ? act[l] : nullptr;
        L.grad[l] = grad ? grad[l] : nullptr;
        L.C[l] = C[l]; L.h[l] = h[l]; L.w[l] = w[l];
        const int hw = h[l] * w[l];
        L.tail0[l] = tail_mod > 1 ? hw - hw % tail_mod : hw;
        const size_t nblk = (C[l] + kL0 - 1) / kL0, ngrp = (C[l] + kSliceC - 1) / kSliceC;
        L.part_off[l] = (long long)off;
        off += (size_t)B * nblk * hw;
        L.tpart_off[l] = (long long)off;
        off += (size_t)B * ngrp * (hw - L.tail0[l]) * 4;
        L.map_off[l] = (long long)off;
        off += (size_t)B * hw;
    }
    *total_floats = off;
    return WSDL_OK;
}


// ---- the class-logit head of the LayerCAM pass (wsdl_class_logit_head) ------------------------------------------------
// logits[b][k] = bias[k] + sum_c pooled[b][c] W[k][c];  cls[b] = class_idx ? class_idx[b] : argmax_k (the first maximum, as
// torch.argmax).  One workgroup per image, a wave per class in turn, lanes strided over the channels.
__global__ void class_logits_kernel(const float* __restrict__ pooled, const float* __restrict__ W, const float* __restrict__ bias,
                                    const long long* __restrict__ class_idx, float* __restrict__ logits, int* __restrict__ cls,
                                    int C, int K) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* pb = pooled + (long long)b * C;
    for (int k = wid; k < K; k += nw) {
        const float* wk = W + (long long)k * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(pb[c], wk[c], s);
        s = wave_sum(s);
        if (lane == 0) logits[(long long)b * K + k] = s + (bias ? bias[k] : 0.f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        if (class_idx) {
            const long long want = class_idx[b];
            best = (want >= 0 && want < K) ? (int)want : -1;       // out of range: the seed kernel writes NaN (torch's gather raises)
        } else {
            float m = logits[(long long)b * K];
            for (int k = 1; k < K; ++k) {
                const float v = logits[(long long)b * K + k];
                if (v > m) { m = v; best = k; }
            }
        }
        cls[b] = best;
    }
}

// d logits[b][cls[b]] / d x[b][c][i] through fc and the global average pool: W[cls[b]][c] / HW for every pixel i
__global__ void class_grad_seed_kernel(const float* __restrict__ W, const int* __restrict__ cls, float* __restrict__ dx, int C,
                                       int HW, int planes) {
    for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const int b = plane / C, c = plane - b * C;
        const int k = cls[b];
        const float g = k >= 0 ? W[(long long)k * C + c] / (float)HW : __builtin_nanf("");
        float* dp = dx + (long long)plane * HW;
        if ((HW & 3) == 0) {
            float4* dp4 = reinterpret_cast<float4*>(dp);
            for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += gridDim.x * blockDim.x)
                dp4[i] = make_float4(g, g, g, g);
            continue;
        }
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) dp[i] = g;
    }
}

}  // namespace

extern "C" {

int wsdl_class_logit_head(const float* pooled, const float* weight, const float* bias, const long long* class_idx,
                          float* logits, int* cls, float* dx, int B, int C, int K, int HW, wsdl_stream_t stream) {
    WSDL_REQUIRE(pooled && weight && logits && cls && dx && B > 0 && C > 0 && K > 0 && HW > 0, "class_logit_head: bad arguments");
    hipStream_t s = wsdl::as_stream(stream);
    hipLaunchKernelGGL(class_logits_kernel, dim3(B), dim3(256), 0, s, pooled, weight, bias, class_idx, logits, cls, C, K);
    WSDL_LAUNCH_CHECK();
    const long long planes = (long long)B * C;
    const int per = wsdl::cdiv(HW, 4 * 256);
    hipLaunchKernelGGL(class_grad_seed_kernel, dim3(per, (unsigned)std::min<long long>(planes, 32768)), dim3(256), 0, s, weight, cls,
                       dx, C, HW, (int)planes);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

size_t wsdl_layercam_workspace(int n_layers, int B, const int* C, const int* h, const int* w) {
    CamLayers L;
    size_t fl = 0;
    if (!C || !h || !w || fill_layers(L, nullptr, nullptr, C, h, w, n_layers, B, &fl)) return 0;
    return fl * sizeof(float);
}

int wsdl_layercam_epilogue(const float* const* act, const float* const* grad, const int* C, const int* h,
                           const int* w, int n_layers, int B, int outH, int outW, float alpha, int variant,
                           float* cam, float thresh, uint8_t* mask, void* ws, size_t ws_bytes,
                           wsdl_stream_t stream) {
    WSDL_REQUIRE(act && grad && C && h && w && cam && ws, "layercam: null pointer");
    WSDL_REQUIRE(outH > 0 && outW > 0 && (variant == 0 || variant == 1), "layercam: bad output size / variant");
    CamLayers L;
    size_t fl = 0;
    if (int rc = fill_layers(L, act, grad, C, h, w, n_layers, B, &fl)) return rc;
    for (int l = 0; l < n_layers; ++l) WSDL_REQUIRE(act[l] && grad[l], "layercam: null layer pointer %d", l);
    if (ws_bytes < fl * sizeof(float)) {
        wsdl::set_error("layercam: workspace %zu < %zu bytes", ws_bytes, fl * sizeof(float));
        return WSDL_EWORKSPACE;
    }
    hipStream_t s = wsdl::as_stream(stream);
    float* wsf = static_cast<float*>(ws);
    double bytes = 0.0;
    for (int l = 0; l < n_layers; ++l) bytes += 8.0 * B * C[l] * h[l] * w[l];
    bytes += (double)B * outH * outW * (4.0 + (mask ? 1.0 : 0.0));
    wsdl::ProfScope prof(WSDL_PROF_LAYERCAM, s, bytes);
    int slices = 1;
    for (int l = 0; l < n_layers; ++l) slices = std::max(slices, wsdl::cdiv(C[l], kSliceC));
    hipLaunchKernelGGL(layercam_partial_kernel, dim3(slices, n_layers, B), dim3(256), 0, s, L, wsf);
    hipLaunchKernelGGL(layercam_normalise_kernel, dim3(n_layers, B), dim3(256), 0, s, L, wsf, alpha, variant);
    int gx = wsdl::cdiv(outH * outW, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(layercam_upsample_kernel, dim3(gx, B), dim3(256), 0, s, L, wsf, cam,
                       thresh >= 0.f ? mask : nullptr, outH, outW, alpha, variant, thresh);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_plane_relu_minmax(const float* x, float* y, int planes, int hw, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && planes > 0 && hw > 0, "plane_relu_minmax: bad arguments");
    hipLaunchKernelGGL(plane_relu_minmax_kernel, dim3(planes), dim3(256), 0, wsdl::as_stream(stream), x, y, hw);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                   float beta2, float eps, int step, const int* step_dev, float grad_scale, wsdl_stream_t stream) {
    WSDL_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev), "adam_step: bad arguments");
    if (step < 1) step = 1;
    WSDL_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v)) % 16 == 0, "adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)std::sqrt(bc2);
    const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, wsdl::as_stream(stream), p, g, m, v, n, beta1,
                       beta2, eps, step_size, sqrt_bc2, grad_scale, step_dev, lr, (const float*)nullptr);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper_dev, const int* step_dev,
                       wsdl_stream_t stream) {
    WSDL_REQUIRE(p && g && m && v && n > 0 && hyper_dev && step_dev, "adam_step_dev: null pointer / empty");
    WSDL_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v)) % 16 == 0, "adam_step_dev: buffers must be 16-byte aligned");
    const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, wsdl::as_stream(stream), p, g, m, v, n, 0.f, 0.f, 0.f, 0.f, 1.f,
                       1.f, step_dev, 0.f, hyper_dev);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
