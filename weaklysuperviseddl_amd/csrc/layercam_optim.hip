// layercam_optim.hip - LayerCAM epilogue (channel-weighted sum / ReLU / min-max / bilinear / layer
// mean / clamp-pow / threshold) and the fused flat Adam step.
//
// LayerCAM epilogue, three launches for all images and layers together:
//   1. layercam_partial : grid (slices, layers, B).  Each block sums relu(g*a) over a slice of the
//      channels for every pixel of the small map; consecutive lanes = consecutive pixels of one
//      channel -> coalesced reads of the NCHW activations/gradients (the 4.8 MB/img that dominate).
//   2. layercam_normalise: grid (layers, B).  Adds the slices in fixed order, ReLU, per-image min-max
//      (second min-max after **alpha for the notebook variant) with wave shuffles.
//   3. layercam_upsample : bilinear to (outH,outW), mean over layers, clamp/pow, threshold -> uint8.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kMaxLayers = 8;
constexpr int kSlices = 16;

struct CamLayers {
    const float* act[kMaxLayers];
    const float* grad[kMaxLayers];
    int C[kMaxLayers], h[kMaxLayers], w[kMaxLayers];
    long long part_off[kMaxLayers];   // float offset of this layer's [B][kSlices][hw] partials
    long long map_off[kMaxLayers];    // float offset of this layer's [B][hw] normalised map
    int n;
};

__global__ void layercam_partial_kernel(CamLayers L, float* __restrict__ ws) {
    const int s = blockIdx.x, l = blockIdx.y, b = blockIdx.z;
    const int C = L.C[l], hw = L.h[l] * L.w[l];
    const int per = (C + kSlices - 1) / kSlices;
    const int c0 = s * per, c1 = min(c0 + per, C);
    const float* a = L.act[l] + (long long)b * C * hw;
    const float* g = L.grad[l] + (long long)b * C * hw;
    float* out = ws + L.part_off[l] + ((long long)b * kSlices + s) * hw;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        float acc = 0.f;
        for (int c = c0; c < c1; ++c) acc += fmaxf(g[(long long)c * hw + p] * a[(long long)c * hw + p], 0.f);
        out[p] = acc;
    }
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
    const int l = blockIdx.x, b = blockIdx.y;
    const int hw = L.h[l] * L.w[l];
    const float* part = ws + L.part_off[l] + (long long)b * kSlices * hw;
    float* map = ws + L.map_off[l] + (long long)b * hw;
    // pass 1: slice sum + relu (kept in the map buffer), min / max
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < kSlices; ++s) v += part[(long long)s * hw + p];
        v = fmaxf(v, 0.f);
        map[p] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, sm);
    // c -= min ; c /= (max_after_shift + 1e-8)
    const float den = (mx - mn) + 1e-8f;
    float mn2 = INFINITY, mx2 = -INFINITY;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
        float v = (map[p] - mn) / den;
        if (variant == 1) v = powf(v, alpha);
        map[p] = v;
        mn2 = fminf(mn2, v);
        mx2 = fmaxf(mx2, v);
    }
    if (variant == 1) {
        block_minmax(mn2, mx2, sm);
        const float den2 = (mx2 - mn2) + 1e-8f;
        for (int p = threadIdx.x; p < hw; p += blockDim.x) map[p] = (map[p] - mn2) / den2;
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

__device__ __forceinline__ void src_index(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
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
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            src_index(oh, (float)h / (float)outH, h, y0, y1, ly0, ly1);
            src_index(ow, (float)w / (float)outW, w, x0, x1, lx0, lx1);
            const float top = lx0 * m[y0 * w + x0] + lx1 * m[y0 * w + x1];
            const float bot = lx0 * m[y1 * w + x0] + lx1 * m[y1 * w + x1];
            sum += ly0 * top + ly1 * bot;
        }
        float v = sum / (float)L.n;
        if (variant == 0) {
            v = fmaxf(v, 0.f);
            if (alpha != 1.f) v = powf(v, alpha);
        }
        cam[(long long)b * outH * outW + o] = v;
        if (mask) mask[(long long)b * outH * outW + o] = (v >= thresh && v > 0.f) ? 1 : 0;
    }
}

// torch.optim.Adam (no amsgrad, no weight decay): exp_avg, exp_avg_sq, bias corrections as torch
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float b1, float b2, float eps, float step_size,
                            float sqrt_bc2, float gscale, const int* __restrict__ step_dev, float lr) {
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
    size_t off = 0;
    L.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        WSDL_REQUIRE(C[l] > 0 && h[l] > 0 && w[l] > 0 && (long long)h[l] * w[l] < (1 << 24), "layercam: bad layer %d shape", l);
        L.act[l] = act ? act[l] : nullptr;
        L.grad[l] = grad ? grad[l] : nullptr;
        L.C[l] = C[l]; L.h[l] = h[l]; L.w[l] = w[l];
        L.part_off[l] = (long long)off;
        off += (size_t)B * kSlices * h[l] * w[l];
        L.map_off[l] = (long long)off;
        off += (size_t)B * h[l] * w[l];
    }
    *total_floats = off;
    return WSDL_OK;
}

}  // namespace

extern "C" {

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
    hipLaunchKernelGGL(layercam_partial_kernel, dim3(kSlices, n_layers, B), dim3(256), 0, s, L, wsf);
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
                       beta2, eps, step_size, sqrt_bc2, grad_scale, step_dev, lr);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
