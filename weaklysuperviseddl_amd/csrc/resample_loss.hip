// resample_loss.hip - bilinear resampling, pixel-wise softmax cross-entropy, the pairwise-affinity
// (normalised-cut / boundary) loss with its backward, compute_affinities, KL and channel softmax.
//
// Pairwise loss formulation (differs from the reference's 24 full-tensor passes): reflect padding is
// folded into per-axis pair weights.  For pixel q and an IN-BOUNDS window pixel p,
//     Wf(q,p) = sum over offsets o with reflect(q+o) = p of g(o)      (q's own terms landing on p)
//     Wr(q,p) = sum over offsets o with reflect(p+o) = q of g(o)      (p's terms landing on q)
// both separable into row x column factors, g(o) = exp(-|o|^2 / 2 ss^2) (1 without a spatial term).
//     loss  = 1/N sum_q sum_p Wf c(p,q) sum_c (P_c(q)-P_c(p))^2 ,  c = exp(-|I(p)-I(q)|^2 / 2 sc^2)
//     dL/dP_c(q) = 2/N sum_p (Wf + Wr) c(p,q) (P_c(q) - P_c(p))
// so forward and backward are ONE sweep over the in-bounds window: image and (softmaxed) predictions
// are staged once per 32x8 tile (+halo) in LDS, 20 B/px of HBM reads + 8 B/px of gradient writes.
#include "common.h"

#include <algorithm>

namespace {

// ------------------------------------------------------------------------------- bilinear
struct Lerp {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp src_index(int o, float scale, int in) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    Lerp r;
    r.i0 = (int)s;
    if (r.i0 > in - 1) r.i0 = in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int h, int w,
                                    int H, int W, long long y_bs, int planes) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const int b = plane / C, c = plane - b * C;
        const float* xp = x + (long long)plane * h * w;
        float* yp = y + (long long)b * y_bs + (long long)c * H * W;
        for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < H * W; o += gridDim.x * blockDim.x) {
            const int oh = o / W, ow = o - oh * W;
            const Lerp a = src_index(oh, sh, h), bb = src_index(ow, sw, w);
            const float top = bb.l0 * xp[a.i0 * w + bb.i0] + bb.l1 * xp[a.i0 * w + bb.i1];
            const float bot = bb.l0 * xp[a.i1 * w + bb.i0] + bb.l1 * xp[a.i1 * w + bb.i1];
            yp[o] = a.l0 * top + a.l1 * bot;
        }
    }
}

// gather form of the transpose: each input pixel sums the output pixels that read it (no atomics)
__global__ void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int h, int w,
                                    int H, int W, long long dy_bs, int planes) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const int b = plane / C, c = plane - b * C;
        const float* gp = dy + (long long)b * dy_bs + (long long)c * H * W;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < h * w; idx += gridDim.x * blockDim.x) {
            const int ih = idx / w, iw = idx - ih * w;
            int oh_lo = (int)floorf(((float)ih - 0.5f) / sh - 0.5f) - 1;
            int oh_hi = (int)ceilf(((float)ih + 1.5f) / sh - 0.5f) + 1;
            int ow_lo = (int)floorf(((float)iw - 0.5f) / sw - 0.5f) - 1;
            int ow_hi = (int)ceilf(((float)iw + 1.5f) / sw - 0.5f) + 1;
            oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
            oh_hi = min(oh_hi, H - 1); ow_hi = min(ow_hi, W - 1);
            float s = 0.f;
            for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                const Lerp a = src_index(oh, sh, h);
                const float wh = (a.i0 == ih ? a.l0 : 0.f) + (a.i1 == ih ? a.l1 : 0.f);
                if (wh == 0.f) continue;
                float rs = 0.f;
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const Lerp bb = src_index(ow, sw, w);
                    const float ww = (bb.i0 == iw ? bb.l0 : 0.f) + (bb.i1 == iw ? bb.l1 : 0.f);
                    rs += ww * gp[oh * W + ow];
                }
                s += wh * rs;
            }
            dx[(long long)plane * h * w + idx] = s;
        }
    }
}

// Same sums for large up-sampling factors (8x logits -> image, the 1x1 -> 32x32 broadcast of the ASPP pooling
// branch): an input pixel gathers hundreds of output pixels, so ONE WAVE works on it - lanes stride over the window,
// fixed shuffle tree (deterministic).  The thread-per-pixel form above left 1 lane of 256 busy on the 1x1 maps.
__global__ void bilinear_bwd_wave_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int h, int w,
                                         int H, int W, long long dy_bs, long long total) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long item = wave0; item < total; item += nwaves) {
        const int plane = (int)(item / (h * w)), idx = (int)(item - (long long)plane * (h * w));
        const int b = plane / C, c = plane - b * C;
        const float* gp = dy + (long long)b * dy_bs + (long long)c * H * W;
        const int ih = idx / w, iw = idx - ih * w;
        int oh_lo = (int)floorf(((float)ih - 0.5f) / sh - 0.5f) - 1;
        int oh_hi = (int)ceilf(((float)ih + 1.5f) / sh - 0.5f) + 1;
        int ow_lo = (int)floorf(((float)iw - 0.5f) / sw - 0.5f) - 1;
        int ow_hi = (int)ceilf(((float)iw + 1.5f) / sw - 0.5f) + 1;
        oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
        oh_hi = min(oh_hi, H - 1); ow_hi = min(ow_hi, W - 1);
        const int nw = ow_hi - ow_lo + 1, n = (oh_hi - oh_lo + 1) * nw;
        float s = 0.f;
        for (int k = lane; k < n; k += 64) {
            const int r = k / nw, oh = oh_lo + r, ow = ow_lo + (k - r * nw);
            const Lerp a = src_index(oh, sh, h);
            const float wh = (a.i0 == ih ? a.l0 : 0.f) + (a.i1 == ih ? a.l1 : 0.f);
            const Lerp bb = src_index(ow, sw, w);
            const float ww = (bb.i0 == iw ? bb.l0 : 0.f) + (bb.i1 == iw ? bb.l1 : 0.f);
            s += wh * ww * gp[oh * W + ow];
        }
        s = wave_sum(s);
        if (lane == 0) dx[(long long)plane * h * w + idx] = s;
    }
}

// ------------------------------------------------------------------------------- reductions
__global__ void finalize_sum_kernel(const float* __restrict__ part, int n, int groups, float scale,
                                    float* __restrict__ out) {
    // out[g] = scale * sum(part[g*n .. g*n+n)) in fixed order, double accumulation
    __shared__ double sm[16];
    const int g = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[(long long)g * n + i];
    s = block_sum_d(s, sm);
    if (threadIdx.x == 0 && g < groups) out[g] = (float)(s * (double)scale);
}

// ------------------------------------------------------------------------------- cross entropy
// nn.CrossEntropyLoss() semantics: mean over the pixels whose label is not `ignore_index` (default -100); such
// pixels get zero loss and zero gradient.  Any other label outside [0, C) is an error in PyTorch (IndexError on the
// CPU, a device assert on a GPU); a kernel cannot raise, so it POISONS the loss with NaN - loud, instead of silently
// training the pixel as a real class (e.g. an un-clamped 255 of a mask PNG; the reference clamps before the loss:
// AlternatingDirectionCutLoss.py:695).  part[0..blocks) = loss partials, part[blocks..2*blocks) = valid-pixel counts.
__global__ void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                  float* __restrict__ part, float* __restrict__ dlogits, int C, int HW,
                                  long long npix, float gscale, long long ignore_index) {
    __shared__ float sm[16];
    float acc = 0.f, cnt = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW;
        const int r = (int)(i - b * HW);
        const float* lp = logits + b * C * HW + r;
        const long long lab = labels[i];
        const bool ignored = lab == ignore_index;
        const bool bad = !ignored && (lab < 0 || lab >= C);
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, lp[(long long)c * HW]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(lp[(long long)c * HW] - m);
        const float lse = m + logf(se);
        if (bad)
            acc += NAN;
        else if (!ignored) {
            acc += lse - lp[lab * HW];
            cnt += 1.f;
        }
        if (dlogits) {
            float* dp = dlogits + b * C * HW + r;
            const float inv = 1.f / se;
            for (int c = 0; c < C; ++c) {
                const float pr = expf(lp[(long long)c * HW] - m) * inv;
                dp[(long long)c * HW] = ignored ? 0.f : (pr - (c == lab ? 1.f : 0.f)) * gscale;
            }
        }
    }
    acc = block_sum(acc, sm);
    cnt = block_sum(cnt, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = acc;
        part[gridDim.x + blockIdx.x] = cnt;
    }
}

// loss = sum / count, inv_count = 1 / count (the factor the gradient still lacks); count == 0 -> NaN like PyTorch
__global__ void ce_finalize_kernel(const float* __restrict__ part, int blocks, float* __restrict__ loss,
                                   float* __restrict__ inv_count) {
    __shared__ double sm[16];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) {
        s += part[i];
        c += part[blocks + i];
    }
    s = block_sum_d(s, sm);
    c = block_sum_d(c, sm);
    if (threadIdx.x == 0) {
        *loss = (float)(s / c);
        if (inv_count) *inv_count = (float)(1.0 / c);
    }
}

// ------------------------------------------------------------------------------- pairwise loss
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// exp(-|dI|^2 / (2 sigma^2)) with the arithmetic pinned (explicit fma chain, symmetric in the sign of the differences):
// the loss kernel and the affinity-cache kernel must produce the same bits for the same pixel pair.
__device__ __forceinline__ float colour_affinity(float d0, float d1, float d2, float inv2sc) {
    const float s = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
    return __expf(-s * inv2sc);                            // v_exp_f32: 2 instructions, ~1 ulp
}

constexpr int kTileW = 32, kTileH = 8;

// Colour affinities of the "forward" half of the window - offsets (dy > 0) or (dy == 0, dx > 0), K/2 maps - for every
// pixel q whose partner p = q + offset is inside the image (0 otherwise; reflect padding is folded into the pair
// weights by the loss kernel, so only in-image pairs exist).  The affinity is symmetric: the loss kernel reads the
// backward half from the partner's forward entry.  cache[k][b][H][W].  Used when the image stays fixed over many loss
// evaluations (refine_pseudo_mask: 10 Adam steps x 5 passes on one image): 24 exp + ~200 VALU ops per pixel and
// evaluation become 24 loads.
template <int R>
__global__ void pairwise_cache_kernel(const float* __restrict__ image, float* __restrict__ cache, int B, int H, int W,
                                      float inv2sc) {
    constexpr int D = 2 * R + 1;
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), g = (int)(i - (long long)b * HW);
        const int y = g / W, x = g - y * W;
        const float* ib = image + (long long)b * 3 * HW;
        const float i0 = ib[g], i1 = ib[HW + g], i2 = ib[2 * HW + g];
        int k = 0;
#pragma unroll
        for (int jy = R; jy < D; ++jy)
#pragma unroll
            for (int jx = 0; jx < D; ++jx) {
                if (jy == R && jx <= R) continue;
                const int py = y + jy - R, px = x + jx - R;
                float cc = 0.f;
                if (py < H && px >= 0 && px < W) {
                    const int n = py * W + px;
                    const float d0 = i0 - ib[n], d1 = i1 - ib[HW + n], d2 = i2 - ib[2 * HW + n];
                    cc = colour_affinity(d0, d1, d2, inv2sc);
                }
                cache[((long long)k * B + b) * HW + g] = cc;
                ++k;
            }
    }
}

constexpr int kRegClasses = 8;     // classes whose gradients stay in registers until the single store

template <int R, bool CACHED>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ preds,
                                                       const float* __restrict__ image,
                                                       float* __restrict__ part, float* __restrict__ dpreds,
                                                       int C, int H, int W, float inv2sc, float inv2ss,
                                                       int use_space, int apply_softmax, float grad_norm,
                                                       const float* __restrict__ cache, int B) {
    constexpr int D = 2 * R + 1;
    constexpr int TW = kTileW + 2 * R, TH = kTileH + 2 * R, TS = TW * TH;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [3 + C][TH][TW]
    __shared__ float red[16];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int HW = H * W;
    const float* ib = image + (long long)b * 3 * HW;
    const float* pb = preds + (long long)b * C * HW;

    // ---- stage image + probabilities (softmax once per staged pixel)
    for (int t = threadIdx.x; t < TS; t += blockDim.x) {
        const int ty = t / TW, tx = t - ty * TW;
        const int gy = y0 - R + ty, gx = x0 - R + tx;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int g = in ? gy * W + gx : 0;
        if constexpr (!CACHED) {
#pragma unroll
            for (int c = 0; c < 3; ++c) lds[c * TS + t] = in ? ib[c * HW + g] : 0.f;
        }
        if (!in) {
            for (int c = 0; c < C; ++c) lds[(3 + c) * TS + t] = 0.f;
        } else if (apply_softmax) {
            float m = -INFINITY;
            for (int c = 0; c < C; ++c) m = fmaxf(m, pb[(long long)c * HW + g]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) {
                const float e = expf(pb[(long long)c * HW + g] - m);
                lds[(3 + c) * TS + t] = e;
                se += e;
            }
            const float inv = 1.f / se;
            for (int c = 0; c < C; ++c) lds[(3 + c) * TS + t] *= inv;
        } else {
            for (int c = 0; c < C; ++c) lds[(3 + c) * TS + t] = pb[(long long)c * HW + g];
        }
    }
    __syncthreads();

    const int lx = threadIdx.x & (kTileW - 1), ly = threadIdx.x / kTileW;
    const int qx = x0 + lx, qy = y0 + ly;
    const bool active = qx < W && qy < H;
    float loss = 0.f;
    if (active) {
        // ---- per-axis pair weights (reflect padding folded in)
        float g1[R + 1];
#pragma unroll
        for (int d = 0; d <= R; ++d) g1[d] = use_space ? expf(-(float)(d * d) * inv2ss) : 1.f;
        float wyF[D], wyR[D], wxF[D], wxR[D];
        const bool inner_y = qy >= 2 * R && qy < H - 2 * R, inner_x = qx >= 2 * R && qx < W - 2 * R;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int off = j - R, ao = off < 0 ? -off : off;
            if (inner_y) {
                wyF[j] = wyR[j] = g1[ao];
            } else {
                const int py = qy + off;
                float f = 0.f, r = 0.f;
                if (py >= 0 && py < H) {
#pragma unroll
                    for (int d = -R; d <= R; ++d) {
                        const float gd = g1[d < 0 ? -d : d];
                        if (reflect_idx(qy + d, H) == py) f += gd;
                        if (reflect_idx(py + d, H) == qy) r += gd;
                    }
                }
                wyF[j] = f;
                wyR[j] = r;
            }
            if (inner_x) {
                wxF[j] = wxR[j] = g1[ao];
            } else {
                const int px = qx + off;
                float f = 0.f, r = 0.f;
                if (px >= 0 && px < W) {
#pragma unroll
                    for (int d = -R; d <= R; ++d) {
                        const float gd = g1[d < 0 ? -d : d];
                        if (reflect_idx(qx + d, W) == px) f += gd;
                        if (reflect_idx(px + d, W) == qx) r += gd;
                    }
                }
                wxF[j] = f;
                wxR[j] = r;
            }
        }
        // ---- colour affinities for the whole window (registers), then one sweep per class
        const int tq = (ly + R) * TW + lx + R;
        float i0 = 0.f, i1 = 0.f, i2 = 0.f;
        if constexpr (!CACHED) {
            i0 = lds[tq];
            i1 = lds[TS + tq];
            i2 = lds[2 * TS + tq];
        }
        const float* cq = CACHED ? cache + (long long)b * HW + qy * W + qx : nullptr;
        const long long cstride = (long long)B * HW;       // between the cache's offset maps
        float af[D * D], ab[D * D];   // forward weight, forward+reverse weight
#pragma unroll
        for (int jy = 0; jy < D; ++jy)
#pragma unroll
            for (int jx = 0; jx < D; ++jx) {
                const int k = jy * D + jx;
                if (jy == R && jx == R) {
                    af[k] = ab[k] = 0.f;
                    continue;
                }
                float cc;
                if constexpr (CACHED) {
                    // forward half: this pixel's own entry; backward half: the partner's entry for the mirrored offset
                    const bool fwd = jy > R || (jy == R && jx > R);
                    const int my = fwd ? jy : 2 * R - jy, mx = fwd ? jx : 2 * R - jx;        // mirrored -> forward offset
                    const int kf = (my - R) * D + mx - (R + 1);                                   // index among the K/2 maps
                    const int py = qy + jy - R, px = qx + jx - R;
                    const bool inb = py >= 0 && py < H && px >= 0 && px < W;
                    cc = inb ? cq[kf * cstride + (fwd ? 0 : (jy - R) * W + (jx - R))] : 0.f;
                } else {
                    const int tp = tq + (jy - R) * TW + (jx - R);
                    const float d0 = i0 - lds[tp], d1 = i1 - lds[TS + tp], d2 = i2 - lds[2 * TS + tp];
                    cc = colour_affinity(d0, d1, d2, inv2sc);
                }
                const float wf = wyF[jy] * wxF[jx], wr = wyR[jy] * wxR[jx];
                af[k] = wf * cc;
                ab[k] = (wf + wr) * cc;
            }
        float dot = 0.f;
        float* dq = dpreds ? dpreds + (long long)b * C * HW + qy * W + qx : nullptr;
        float Gr[kRegClasses];           // per-class gradients stay in registers until the single store (C <= 8)
        for (int c0 = 0; c0 < C; c0 += kRegClasses) {
#pragma unroll
            for (int cc = 0; cc < kRegClasses; ++cc) {
                const int c = c0 + cc;
                Gr[cc] = 0.f;
                if (c < C) {
                    const float* pl = lds + (3 + c) * TS;
                    const float pq = pl[tq];
                    float G = 0.f;
#pragma unroll
                    for (int jy = 0; jy < D; ++jy)
#pragma unroll
                        for (int jx = 0; jx < D; ++jx) {
                            const int k = jy * D + jx;
                            const float df = pq - pl[tq + (jy - R) * TW + (jx - R)];
                            loss = fmaf(af[k] * df, df, loss);          // pinned: both kernel variants round alike
                            G = fmaf(ab[k], df, G);
                        }
                    if (dq) {
                        G *= grad_norm;
                        dot += pq * G;
                        Gr[cc] = G;
                        if (!apply_softmax || C > kRegClasses) dq[(long long)c * HW] = G;
                    }
                }
            }
        }
        if (dq && apply_softmax) {
            if (C <= kRegClasses) {
#pragma unroll
                for (int c = 0; c < kRegClasses; ++c)
                    if (c < C) dq[(long long)c * HW] = lds[(3 + c) * TS + tq] * (Gr[c] - dot);
            } else {
                for (int c = 0; c < C; ++c) {
                    const float pq = lds[(3 + c) * TS + tq];
                    dq[(long long)c * HW] = pq * (dq[(long long)c * HW] - dot);
                }
            }
        }
    }
    loss = block_sum(loss, red);
    if (threadIdx.x == 0)
        part[((long long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = loss;
}

template <int R>
__global__ void affinities_kernel(const float* __restrict__ image, float* __restrict__ out, int B, int H,
                                  int W, float inv2sc, float inv2ss, int use_space) {
    constexpr int D = 2 * R + 1;
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), g = (int)(i - (long long)b * HW);
        const int y = g / W, x = g - y * W;
        const float* ib = image + (long long)b * 3 * HW;
        const float i0 = ib[g], i1 = ib[HW + g], i2 = ib[2 * HW + g];
        int k = 0;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                if (dy == 0 && dx == 0) continue;
                const int n = reflect_idx(y + dy, H) * W + reflect_idx(x + dx, W);
                const float d0 = i0 - ib[n], d1 = i1 - ib[HW + n], d2 = i2 - ib[2 * HW + n];
                float e = -(d0 * d0 + d1 * d1 + d2 * d2) * inv2sc;
                if (use_space) e -= (float)(dy * dy + dx * dx) * inv2ss;
                out[((long long)k * B + b) * HW + g] = expf(e);
                ++k;
            }
        (void)D;
    }
}

// ------------------------------------------------------------------------------- KL / softmax
__global__ void kl_div_kernel(const float* __restrict__ xn, const float* __restrict__ s,
                              float* __restrict__ part, float* __restrict__ dxn, size_t n, float inv_batch) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = s[i], xe = xn[i] + 1e-8f;
        acc += (t > 0.f ? t * logf(t) : 0.f) - t * logf(xe);
        if (dxn) dxn[i] = -t / xe * inv_batch;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// per-image KL: grid (blocks, N); part[(img*gridDim.x + blockIdx.x)]
__global__ void kl_div_image_kernel(const float* __restrict__ xn, const float* __restrict__ s,
                                    float* __restrict__ part, float* __restrict__ dxn, size_t per_image) {
    __shared__ float sm[16];
    const size_t base = (size_t)blockIdx.y * per_image;
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_image; i += (size_t)gridDim.x * blockDim.x) {
        const float t = s[base + i], xe = xn[base + i] + 1e-8f;
        acc += (t > 0.f ? t * logf(t) : 0.f) - t * logf(xe);
        if (dxn) dxn[base + i] = -t / xe;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

// out = dkl + coef_i * dnc with coef_i = lambda * kl_i / (nc_i + 1e-6), all per image, all on the device
__global__ void refine_combine_kernel(const float* __restrict__ dkl, const float* __restrict__ dnc,
                                      const float* __restrict__ kl, const float* __restrict__ nc, float lambda,
                                      float nc_scale, float* __restrict__ out, size_t per_image) {
    const int img = blockIdx.y;
    const float ncv = nc[img] * nc_scale;
    const float coef = lambda * (kl[img] / (ncv + 1e-6f)) * nc_scale;
    const size_t base = (size_t)img * per_image;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_image; i += (size_t)gridDim.x * blockDim.x)
        out[base + i] = dkl[base + i] + coef * dnc[base + i];
}

__global__ void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW,
                                   long long npix) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW;
        const long long base = b * C * HW + (i - b * HW);
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, x[base + (long long)c * HW]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[base + (long long)c * HW] - m);
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) y[base + (long long)c * HW] = expf(x[base + (long long)c * HW] - m) * inv;
    }
}

__global__ void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                   float* __restrict__ dx, int C, int HW, long long npix) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW;
        const long long base = b * C * HW + (i - b * HW);
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += y[base + (long long)c * HW] * dy[base + (long long)c * HW];
        for (int c = 0; c < C; ++c)
            dx[base + (long long)c * HW] = y[base + (long long)c * HW] * (dy[base + (long long)c * HW] - dot);
    }
}

inline dim3 plane_grid(int planes, int HW) {
    int gx = wsdl::cdiv(HW, 256);
    if (gx > 64) gx = 64;
    return dim3(gx < 1 ? 1 : gx, planes > 65535 ? 65535 : planes);
}
inline int flat_blocks(long long n) { return (int)std::min<long long>((n + 255) / 256, wsdl::kReduceSlots); }

}  // namespace

extern "C" {

size_t wsdl_reduce_workspace(void) { return wsdl::kReduceSlots * sizeof(float); }

int wsdl_bilinear_fwd(const float* x, float* y, int B, int C, int h, int w, int H, int W, long long y_bs,
                      wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && B > 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bilinear_fwd: bad arguments");
    if (!y_bs) y_bs = (long long)C * H * W;
    hipLaunchKernelGGL(bilinear_fwd_kernel, plane_grid(B * C, H * W), dim3(256), 0, wsdl::as_stream(stream), x, y,
                       C, h, w, H, W, y_bs, B * C);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_bilinear_bwd(const float* dy, float* dx, int B, int C, int h, int w, int H, int W,
                      long long dy_bs, wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && dx && B > 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bilinear_bwd: bad arguments");
    if (!dy_bs) dy_bs = (long long)C * H * W;
    if ((long long)H * W >= 32ll * h * w) {       // >= 32 contributions per input pixel on average: a wave per pixel
        const long long total = (long long)B * C * h * w;
        const int blocks = (int)std::min<long long>((total + 3) / 4, 16384);
        hipLaunchKernelGGL(bilinear_bwd_wave_kernel, dim3(blocks), dim3(256), 0, wsdl::as_stream(stream), dy, dx, C, h, w,
                           H, W, dy_bs, total);
    } else {
        hipLaunchKernelGGL(bilinear_bwd_kernel, plane_grid(B * C, h * w), dim3(256), 0, wsdl::as_stream(stream), dy, dx,
                           C, h, w, H, W, dy_bs, B * C);
    }
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_softmax_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                            float* inv_count, int B, int C, int H, int W, float grad_scale, long long ignore_index,
                            void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(logits && labels && loss && ws && B > 0 && C > 0 && H > 0 && W > 0, "softmax_ce: bad arguments");
    WSDL_REQUIRE(!dlogits || inv_count, "softmax_ce: the gradient needs inv_count (it is left unnormalised)");
    if (ws_bytes < wsdl_reduce_workspace()) {
        wsdl::set_error("softmax_ce: workspace too small");
        return WSDL_EWORKSPACE;
    }
    const long long npix = (long long)B * H * W;
    const int blocks = std::min(flat_blocks(npix), wsdl::kReduceSlots / 2);
    hipStream_t s = wsdl::as_stream(stream);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(blocks), dim3(256), 0, s, logits, labels, part, dlogits, C, H * W,
                       npix, grad_scale, ignore_index);
    WSDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, part, blocks, loss, inv_count);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

size_t wsdl_pairwise_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * wsdl::cdiv(H, kTileH) * wsdl::cdiv(W, kTileW) * sizeof(float);
}

int wsdl_pairwise_affinity_loss_fwd_bwd(const float* preds, const float* image, float* loss, float* dpreds,
                                        int B, int C, int H, int W, int window, float sigma_color,
                                        float sigma_space, int apply_softmax, int normalise,
                                        const float* cache, void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(preds && (image || cache) && loss && ws, "pairwise_loss: null pointer");
    WSDL_REQUIRE(B > 0 && B <= 65535 && C > 0 && C <= 32 && H > 0 && W > 0, "pairwise_loss: bad shape (C <= 32)");
    WSDL_REQUIRE(window == 3 || window == 5 || window == 7, "pairwise_loss: window must be 3, 5 or 7");
    const int R = window / 2;
    WSDL_REQUIRE(H > R && W > R, "pairwise_loss: reflect padding needs H, W > window/2 (as F.pad)");
    WSDL_REQUIRE(sigma_color > 0.f, "pairwise_loss: sigma_color must be positive");
    if (ws_bytes < wsdl_pairwise_workspace(B, H, W)) {
        wsdl::set_error("pairwise_loss: workspace too small");
        return WSDL_EWORKSPACE;
    }
    const int K = window * window - 1;
    const dim3 grid(wsdl::cdiv(W, kTileW), wsdl::cdiv(H, kTileH), B);
    const int tiles = grid.x * grid.y;
    const size_t lds = (size_t)(3 + C) * (kTileW + 2 * R) * (kTileH + 2 * R) * sizeof(float);
    WSDL_REQUIRE(lds <= 64 * 1024, "pairwise_loss: C too large for the LDS tile");
    const double N = normalise == 0 ? (double)B * H * W * K * C : (double)H * W * K;
    const float inv2sc = 1.f / (2.f * sigma_color * sigma_color);
    const int use_space = sigma_space > 0.f;
    const float inv2ss = use_space ? 1.f / (2.f * sigma_space * sigma_space) : 0.f;
    const float gnorm = (float)(2.0 / N);
    float* part = static_cast<float*>(ws);
    hipStream_t s = wsdl::as_stream(stream);
    const double bytes = (double)B * H * W * 4.0 * ((C + 3) + (dpreds ? C : 0));
    {
        wsdl::ProfScope prof(WSDL_PROF_PAIRWISE, s, bytes);
#define WSDL_PAIRWISE(RR, CA)                                                                                     \
    hipLaunchKernelGGL((pairwise_kernel<RR, CA>), grid, dim3(256), lds, s, preds, image, part, dpreds, C, H, W, inv2sc, \
                       inv2ss, use_space, apply_softmax, gnorm, cache, B)
        if (cache) {
            if (R == 1) WSDL_PAIRWISE(1, true);
            else if (R == 2) WSDL_PAIRWISE(2, true);
            else WSDL_PAIRWISE(3, true);
        } else {
            if (R == 1) WSDL_PAIRWISE(1, false);
            else if (R == 2) WSDL_PAIRWISE(2, false);
            else WSDL_PAIRWISE(3, false);
        }
#undef WSDL_PAIRWISE
    }
    WSDL_LAUNCH_CHECK();
    if (normalise == 0)
        hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, s, part, B * tiles, 1, (float)(1.0 / N), loss);
    else
        hipLaunchKernelGGL(finalize_sum_kernel, dim3(B), dim3(256), 0, s, part, tiles, B, (float)(1.0 / N), loss);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

size_t wsdl_pairwise_cache_bytes(int B, int H, int W, int window) {
    if (B <= 0 || H <= 0 || W <= 0 || (window != 3 && window != 5 && window != 7)) return 0;
    return (size_t)((window * window - 1) / 2) * B * H * W * sizeof(float);
}

int wsdl_pairwise_cache(const float* image, float* cache, int B, int H, int W, int window, float sigma_color,
                        wsdl_stream_t stream) {
    WSDL_REQUIRE(image && cache && B > 0 && H > 0 && W > 0, "pairwise_cache: bad arguments");
    WSDL_REQUIRE(window == 3 || window == 5 || window == 7, "pairwise_cache: window must be 3, 5 or 7");
    WSDL_REQUIRE(sigma_color > 0.f, "pairwise_cache: sigma_color must be positive");
    const int R = window / 2;
    const float inv2sc = 1.f / (2.f * sigma_color * sigma_color);
    const int blocks = flat_blocks((long long)B * H * W);
    hipStream_t s = wsdl::as_stream(stream);
    if (R == 1) hipLaunchKernelGGL((pairwise_cache_kernel<1>), dim3(blocks), dim3(256), 0, s, image, cache, B, H, W, inv2sc);
    else if (R == 2) hipLaunchKernelGGL((pairwise_cache_kernel<2>), dim3(blocks), dim3(256), 0, s, image, cache, B, H, W, inv2sc);
    else hipLaunchKernelGGL((pairwise_cache_kernel<3>), dim3(blocks), dim3(256), 0, s, image, cache, B, H, W, inv2sc);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_compute_affinities(const float* image, float* out, int B, int H, int W, int window,
                            float sigma_color, float sigma_space, wsdl_stream_t stream) {
    WSDL_REQUIRE(image && out && B > 0 && H > 0 && W > 0, "compute_affinities: bad arguments");
    WSDL_REQUIRE(window == 3 || window == 5 || window == 7, "compute_affinities: window must be 3, 5 or 7");
    const int R = window / 2;
    WSDL_REQUIRE(H > R && W > R && sigma_color > 0.f, "compute_affinities: bad geometry");
    const float inv2sc = 1.f / (2.f * sigma_color * sigma_color);
    const int use_space = sigma_space > 0.f;
    const float inv2ss = use_space ? 1.f / (2.f * sigma_space * sigma_space) : 0.f;
    const int blocks = flat_blocks((long long)B * H * W);
    hipStream_t s = wsdl::as_stream(stream);
    if (R == 1) hipLaunchKernelGGL((affinities_kernel<1>), dim3(blocks), dim3(256), 0, s, image, out, B, H, W, inv2sc, inv2ss, use_space);
    else if (R == 2) hipLaunchKernelGGL((affinities_kernel<2>), dim3(blocks), dim3(256), 0, s, image, out, B, H, W, inv2sc, inv2ss, use_space);
    else hipLaunchKernelGGL((affinities_kernel<3>), dim3(blocks), dim3(256), 0, s, image, out, B, H, W, inv2sc, inv2ss, use_space);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_kl_div_fwd_bwd(const float* xn, const float* s, float* loss, float* dxn, size_t n, int batch,
                        void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(xn && s && loss && ws && n > 0 && batch > 0, "kl_div: bad arguments");
    if (ws_bytes < wsdl_reduce_workspace()) {
        wsdl::set_error("kl_div: workspace too small");
        return WSDL_EWORKSPACE;
    }
    const int blocks = flat_blocks((long long)n);
    hipStream_t st = wsdl::as_stream(stream);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(kl_div_kernel, dim3(blocks), dim3(256), 0, st, xn, s, part, dxn, n, 1.f / (float)batch);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, st, part, blocks, 1, 1.f / (float)batch, loss);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_kl_div_per_image_fwd_bwd(const float* xn, const float* s, float* loss, float* dxn, int N,
                                  size_t per_image, void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(xn && s && loss && ws && N > 0 && N <= 65535 && per_image > 0, "kl_div_per_image: bad arguments");
    int blocks = (int)std::min<size_t>((per_image + 255) / 256, 64);
    if (ws_bytes < (size_t)N * blocks * sizeof(float)) {
        wsdl::set_error("kl_div_per_image: workspace too small");
        return WSDL_EWORKSPACE;
    }
    hipStream_t st = wsdl::as_stream(stream);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(kl_div_image_kernel, dim3(blocks, N), dim3(256), 0, st, xn, s, part, dxn, per_image);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(N), dim3(256), 0, st, part, blocks, N, 1.f, loss);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_refine_combine(const float* dkl, const float* dnc, const float* kl, const float* nc, float lambda,
                        float nc_scale, float* out, int N, size_t per_image, wsdl_stream_t stream) {
    WSDL_REQUIRE(dkl && dnc && kl && nc && out && N > 0 && N <= 65535 && per_image > 0, "refine_combine: bad arguments");
    int blocks = (int)std::min<size_t>((per_image + 255) / 256, 256);
    hipLaunchKernelGGL(refine_combine_kernel, dim3(blocks, N), dim3(256), 0, wsdl::as_stream(stream), dkl, dnc, kl, nc,
                       lambda, nc_scale, out, per_image);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_softmax_fwd(const float* x, float* y, int B, int C, int HW, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "softmax_fwd: bad arguments");
    const long long npix = (long long)B * HW;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(flat_blocks(npix)), dim3(256), 0, wsdl::as_stream(stream), x, y, C,
                       HW, npix);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_softmax_bwd(const float* y, const float* dy, float* dx, int B, int C, int HW, wsdl_stream_t stream) {
    WSDL_REQUIRE(y && dy && dx && B > 0 && C > 0 && HW > 0, "softmax_bwd: bad arguments");
    const long long npix = (long long)B * HW;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(flat_blocks(npix)), dim3(256), 0, wsdl::as_stream(stream), y, dy,
                       dx, C, HW, npix);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
