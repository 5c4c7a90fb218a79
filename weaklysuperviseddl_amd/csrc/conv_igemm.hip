// conv_igemm.hip - convolution forward / input-gradient / weight-gradient as implicit GEMMs on the
// CDNA4 fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 64 FLOP/clk/SIMD).
//
// Layout choices (NCHW activations, 64-lane waves):
//   forward / dgrad : D[cout][pixel] = sum_k Wt[k][cout] * Xg[k][pixel],  k = tap*Cin + ci.
//       The MFMA D tile has the pixel on the lane (col = lane&31), so every accumulator register is
//       stored as 2 x 128-byte runs of consecutive pixels of one output channel: coalesced NCHW writes.
//       Xg is gathered on the fly (im2col never materialised): for one k the 32 lanes of a half wave
//       read consecutive pixels of one input channel -> coalesced HBM reads, conflict-free LDS writes.
//       Weights are pre-laid-out k-major ([k][cout]) so the A tile is read with 16-byte loads.
//   wgrad           : D[cout][n] = sum_pixel dY[cout][pixel] * Xg[n][pixel],  n = tap*Cin + ci,
//       both operands pixel-contiguous; LDS rows padded to 33 dwords so the MFMA operand reads
//       (32 lanes x stride 33) hit 32 distinct banks.  Split over pixel ranges into slabs, summed in
//       fixed order by a second kernel (bitwise reproducible, no float atomics).
// One code path serves forward and dgrad: the gather is  num = o*ah + t*bh + ch; src = num/sh
// (valid when divisible and in range): forward (ah,bh,ch,sh) = (stride, dil, -pad, 1),
// dgrad = (1, -dil, +pad, stride) over dY with the [tap][cout][cin] weight layout.
#include "common.h"

#include <type_traits>
#include <algorithm>
#include <cstdint>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 256;

// an additional source of a multi-source launch (conv_split.h, MS): operand tensor, weights, scale and tap geometry
struct ConvSrc {
    const float* x;
    const float* wt;
    const float* x_amax;
    int KH, KW, bh, ch, K, Cin;
    long long x_bs;
    unsigned x_bytes;
};

struct ConvP {
    const float* x;
    const float* wt;   // [K][Cout]
    float* y;
    const float* scale;
    const float* shift;
    const float* res;
    int B, Cin, H, W, Cout, OH, OW, KH, KW;
    int ah, bh, ch, sh;
    int K;
    long long x_bs, y_bs, res_bs;
    int relu, accumulate;
    int P;  // B*OH*OW
    unsigned x_bytes;   // extent of x in bytes (buffer descriptor; < 2^31 for the fast path)
    int ksplit;         // > 1: gridDim.z blocks share a tile's K chunks and write raw partial sums to `slab`
    float* slab;        // [ksplit][Cout][P]
    int ow0, own;       // output-column window: pixels are (b, oh, ow0 <= ow < ow0+own), P = B*OH*own
    // column bands inside ONE launch (fast kernel): band i owns pixel tiles [b_tile0[i], b_tile0[i+1])
    int nb;
    int b_ow0[4], b_own[4], b_tile0[5];
    int grid_x;         // host only: pixel tiles of the launch when banded (0 = cdiv(P, BN))
    int xcd_py;         // split kernels: > 0 = XCD-aware tile order with this many row groups (1, 2, 4 or 8); 0 = launch order
    int xcd_rowfast;    // ... with the XCD's tiles taken row tile fastest (multi-source launches)
    int tile_img_major; // split kernels: pixel tiles of a band taken image-fastest (conv_split.h: workgroups that run at once share their tap lists)
    const float* x_amax;   // split kernels, fp16x2 arithmetic: device scalar >= max|x| (the scale of the activation operand)
    float* y_amax;         // optional: receives max|y| of what the epilogue stores (atomicMax into a zeroed device scalar)
    // accumulate with a bit mask (dgrad of a bottleneck's first convolution): the value already in y counts only where bit
    // e % 8 of acc_mask[e / 8] is set, e = the element's index from acc_base (= y of the whole batch) - y holds the block
    // output's gradient and the mask is the block's final ReLU, so the masked gradient of the identity branch is never
    // written out by the BatchNorm backward
    const uint8_t* acc_mask;
    const float* acc_base;
    int nsrc;              // multi-source launches: number of ADDITIONAL sources in src[] (0: an ordinary convolution)
    ConvSrc src[3];
};

__device__ __forceinline__ float acc_prev(const ConvP& p, const float* ptr) {
    float o = *ptr;
    if (p.acc_mask) {
        const long long e = ptr - p.acc_base;
        if (!((p.acc_mask[e >> 3] >> (e & 7)) & 1)) o = 0.f;
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// Pipeline: LDS double buffer, ONE barrier per K-chunk.  Iteration q: registers (holding chunk q+1, whose
// global loads were issued one iteration ago) -> LDS[(q+1)&1]; issue the global loads of chunk q+2; MFMAs on
// LDS[q&1]; barrier.  Taps whose source pixels are padding for EVERY pixel of the block's tile are skipped
// (ALIGNED path): exact zeros are not multiplied (ASPP dilation 12/24/36 on 32x32 maps: 52 % of the
// nominal K-chunks never execute).
template <int BM, int BN, int WM, bool ALIGNED>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(ConvP p) {
    constexpr int BK = 16;
    constexpr int WN = 4 / WM;              // 4 waves arranged WM x WN
    constexpr int MI = BM / WM / 32;        // 32x32 MFMA tiles per wave along M
    constexpr int NI = BN / WN / 32;        // ... along N (pixels)
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN, "bad tile");
    constexpr int A_F4 = BK * BM / 4 / kThreads;   // float4 per thread for the A tile
    constexpr int B_STEP = kThreads / BN;          // k rows covered per pass
    constexpr int B_PER = BK / B_STEP;             // loads per thread for the B tile

    __shared__ float As[2][BK * BM];
    __shared__ float Bs[2][BK * BN];
    __shared__ int vtaps[64];                      // ids of the taps that touch at least one real pixel
    constexpr int kTab = ALIGNED ? 1 : 1024;
    __shared__ int ktab[kTab][2];                  // unaligned path: k -> (ci*HW, (ti*bh) << 16 | (tj*bh & 0xffff))

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int OHOW = p.OH * p.OW;
    const int HW = p.H * p.W;
    const bool use_tab = !ALIGNED && p.K <= kTab;
    if (use_tab) {
        for (int k = tid; k < p.K; k += kThreads) {
            const int tp = k / p.Cin, ci = k - tp * p.Cin;
            const int ti = tp / p.KW, tj = tp - ti * p.KW;
            ktab[k][0] = ci * HW;
            ktab[k][1] = (int)(((unsigned)(ti * p.bh) << 16) | ((unsigned)(tj * p.bh) & 0xffffu));
        }
        __syncthreads();
    }

    // ---- per-thread gather state for the B (activation) tile
    const int pl = tid % BN, kr = tid / BN;
    const int pix = n0 + pl;
    const bool pix_ok = pix < p.P;
    int pb = 0, poh = 0, pow_ = 0;
    if (pix_ok) {
        pb = pix / OHOW;
        const int r = pix - pb * OHOW;
        poh = r / p.OW;
        pow_ = r - poh * p.OW;
    }
    const float* xb = p.x + (long long)pb * p.x_bs;
    const bool cout_vec = (p.Cout & 3) == 0;

    // ---- chunk list
    const int T = p.KH * p.KW;
    int nq;                 // number of K-chunks this block executes
    int cpt = 1;            // chunks per tap (ALIGNED)
    if (ALIGNED) {
        cpt = p.Cin / BK;
        int nv = 0;
        for (int t = 0; t < T; ++t) {
            const int ti = t / p.KW, tj = t - ti * p.KW;
            const int nh = poh * p.ah + ti * p.bh + p.ch;
            const int nw = pow_ * p.ah + tj * p.bh + p.ch;
            bool ok = pix_ok && nh >= 0 && nw >= 0;
            if (ok) {
                const int ih = nh / p.sh, iw = nw / p.sh;
                ok = (ih * p.sh == nh) && (iw * p.sh == nw) && ih < p.H && iw < p.W;
            }
            if (__syncthreads_or(ok)) {
                if (tid == 0) vtaps[nv] = t;
                ++nv;
            }
        }
        nq = nv * cpt;
        __syncthreads();
    } else {
        nq = (p.K + BK - 1) / BK;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_F4];
    float rb[B_PER];

    auto load_tiles = [&](int q) {
        int k0, tap = 0;
        if (ALIGNED) {
            const int vi = q / cpt;
            tap = vtaps[vi];
            k0 = tap * p.Cin + (q - vi * cpt) * BK;
        } else {
            k0 = q * BK;
        }
        // A: weights, k-major
#pragma unroll
        for (int e = 0; e < A_F4; ++e) {
            const int qq = tid + e * kThreads;
            const int row = qq / (BM / 4), c4 = (qq % (BM / 4)) * 4;
            const int k = k0 + row, m = m0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < p.K) {
                const float* src = p.wt + (long long)k * p.Cout + m;
                if (cout_vec) {
                    if (m < p.Cout) v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (m + 0 < p.Cout) v.x = src[0];
                    if (m + 1 < p.Cout) v.y = src[1];
                    if (m + 2 < p.Cout) v.z = src[2];
                    if (m + 3 < p.Cout) v.w = src[3];
                }
            }
            ra[e] = v;
        }
        // B: gathered activations
        if (ALIGNED) {
            const int ci0 = k0 - tap * p.Cin + kr;
            const int ti = tap / p.KW, tj = tap - ti * p.KW;
            const int nh = poh * p.ah + ti * p.bh + p.ch;
            const int nw = pow_ * p.ah + tj * p.bh + p.ch;
            bool ok = pix_ok && nh >= 0 && nw >= 0;
            int ih = nh, iw = nw;
            if (p.sh != 1) {
                ih = nh / p.sh;
                iw = nw / p.sh;
                ok = ok && (ih * p.sh == nh) && (iw * p.sh == nw);
            }
            ok = ok && ih < p.H && iw < p.W;
            const float* src = xb + (long long)ci0 * HW + ih * p.W + iw;
#pragma unroll
            for (int e = 0; e < B_PER; ++e) rb[e] = ok ? src[(long long)e * B_STEP * HW] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < B_PER; ++e) {
                const int k = k0 + kr + e * B_STEP;
                float v = 0.f;
                if (use_tab) {
                    // decode table in LDS: no integer divisions per gathered element (7x7 stem: K = 147)
                    if (pix_ok && k < p.K) {
                        const int cioff = ktab[k][0], sh2 = ktab[k][1];
                        const int nh = poh * p.ah + (sh2 >> 16) + p.ch;
                        const int nw = pow_ * p.ah + (int)(short)(sh2 & 0xffff) + p.ch;
                        if (nh >= 0 && nw >= 0) {
                            int ih = nh, iw = nw;
                            bool ok = true;
                            if (p.sh != 1) {
                                ih = nh / p.sh;
                                iw = nw / p.sh;
                                ok = ih * p.sh == nh && iw * p.sh == nw;
                            }
                            if (ok && ih < p.H && iw < p.W) v = xb[(long long)cioff + ih * p.W + iw];
                        }
                    }
                } else if (pix_ok && k < p.K) {
                    const int tp = k / p.Cin, ci = k - tp * p.Cin;
                    const int ti = tp / p.KW, tj = tp - ti * p.KW;
                    const int nh = poh * p.ah + ti * p.bh + p.ch;
                    const int nw = pow_ * p.ah + tj * p.bh + p.ch;
                    if (nh >= 0 && nw >= 0) {
                        const int ih = nh / p.sh, iw = nw / p.sh;
                        if (ih * p.sh == nh && iw * p.sh == nw && ih < p.H && iw < p.W)
                            v = xb[(long long)ci * HW + ih * p.W + iw];
                    }
                }
                rb[e] = v;
            }
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < A_F4; ++e) {
            const int qq = tid + e * kThreads;
            *reinterpret_cast<float4*>(&As[buf][qq * 4]) = ra[e];   // row-major [BK][BM]
        }
#pragma unroll
        for (int e = 0; e < B_PER; ++e) Bs[buf][(kr + e * B_STEP) * BN + pl] = rb[e];
    };

    if (nq > 0) {
        load_tiles(0);
        store_tiles(0);
        if (nq > 1) load_tiles(1);
    }
    __syncthreads();
    {
        const int l31 = lane & 31, lh = lane >> 5;
        for (int q = 0; q < nq; ++q) {
            const int cur = q & 1;
            if (q + 1 < nq) store_tiles(cur ^ 1);      // chunk q+1 (registers) -> the idle LDS buffer
            if (q + 2 < nq) load_tiles(q + 2);         // in flight under the MFMAs below
            const float* Ab = As[cur];
            const float* Bb = Bs[cur];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = Ab[(kk + lh) * BM + wm * (MI * 32) + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = Bb[(kk + lh) * BN + wn * (NI * 32) + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: D row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int opix = n0 + wn * (NI * 32) + j * 32 + l31;
        if (opix >= p.P) continue;
        const int ob = opix / OHOW;
        const int orp = opix - ob * OHOW;
        float* yb = p.y + (long long)ob * p.y_bs + orp;
        const float* rbp = p.res ? p.res + (long long)ob * p.res_bs + orp : nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= p.Cout) continue;
                float v = acc[i][j][r];
                if (p.scale) v *= p.scale[co];
                if (p.shift) v += p.shift[co];
                const long long off = (long long)co * OHOW;
                if (rbp) v += rbp[off];
                if (p.accumulate) v += acc_prev(p, yb + off);
                if (p.relu) v = fmaxf(v, 0.f);
                yb[off] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path (Cin % 16 == 0, Cout % 4 == 0, x < 2 GiB): same tiling / pipeline as above, but all global
// traffic goes through buffer descriptors: per-lane byte offsets change only when the tap changes, the
// per-chunk advance (ci block, k row of the weights) rides in the scalar offset, and padding / ragged
// edges are lanes whose offset is parked beyond num_records (hardware returns 0) - no exec-mask
// branches and no 64-bit address arithmetic inside the K loop.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int BK>
__global__ __launch_bounds__(kThreads, 4) void conv_igemm_fast_kernel(ConvP p) {
    constexpr int WN = 4 / WM;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN, "bad tile");
    constexpr int A_F4 = BK * BM / 4 / kThreads;
    constexpr int B_STEP = kThreads / BN;
    constexpr int B_PER = BK / B_STEP;
    constexpr unsigned kOOB = 0x80000000u;

    __shared__ float As[2][BK * BM];
    __shared__ float Bs[2][BK * BN];
    __shared__ int vtaps[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM;
    // column bands: this block's band and its tile inside the band (scalar selects, no dynamic indexing: the
    // window must stay in SGPRs or every buffer load below turns into a waterfall loop)
    int w_ow0 = p.ow0, w_own = p.own, w_tile0 = 0;
    if (p.nb > 1) {
        w_ow0 = p.b_ow0[0];
        w_own = p.b_own[0];
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < p.nb && (int)blockIdx.x >= p.b_tile0[i]) {
                w_ow0 = p.b_ow0[i];
                w_own = p.b_own[i];
                w_tile0 = p.b_tile0[i];
            }
    }
    const int W_P = p.nb > 1 ? p.B * p.OH * w_own : p.P;
    const int n0 = ((int)blockIdx.x - w_tile0) * BN;
    const int OHOW = p.OH * p.OW;
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.K * p.Cout * 4, 0x00020000);

    const int pl = tid % BN, kr = tid / BN;
    const int pix = n0 + pl;
    const bool pix_ok = pix < W_P;
    int pb = 0, poh = 0, pow_ = 0;
    const int OHW = p.OH * w_own;       // pixels of one image inside this block's column window
    if (pix_ok) {
        pb = pix / OHW;
        const int r = pix - pb * OHW;
        poh = r / w_own;
        pow_ = w_ow0 + (r - poh * w_own);
    }
    const unsigned img_off = (unsigned)((long long)pb * p.x_bs) + (unsigned)(kr * HW);   // elements

    auto tap_src = [&](int t, int& sp) {      // source pixel of this thread's output pixel under tap t
        const int ti = t / p.KW, tj = t - ti * p.KW;
        const int nh = poh * p.ah + ti * p.bh + p.ch;
        const int nw = pow_ * p.ah + tj * p.bh + p.ch;
        bool ok = pix_ok && nh >= 0 && nw >= 0;
        int ih = nh, iw = nw;
        if (p.sh != 1) {
            ih = nh / p.sh;
            iw = nw / p.sh;
            ok = ok && (ih * p.sh == nh) && (iw * p.sh == nw);
        }
        ok = ok && ih < p.H && iw < p.W;
        sp = ih * p.W + iw;
        return ok;
    };

    const int T = p.KH * p.KW;
    const int cpt = p.Cin / BK;
    int nv = 0;
    for (int t = 0; t < T; ++t) {
        int sp;
        const bool ok = tap_src(t, sp);
        if (__syncthreads_or(ok)) {
            if (tid == 0) vtaps[nv] = t;
            ++nv;
        }
    }
    const int nq_all = nv * cpt;
    // split-K: this block's share of the tile's (valid) chunks
    const int q0 = p.ksplit > 1 ? (int)((long long)nq_all * blockIdx.z / p.ksplit) : 0;
    const int q1 = p.ksplit > 1 ? (int)((long long)nq_all * (blockIdx.z + 1) / p.ksplit) : nq_all;
    const int nq = q1 - q0;
    __syncthreads();

    // weights: per-thread constant byte offsets inside one [BK][Cout] slab
    unsigned voff_a[A_F4];
#pragma unroll
    for (int e = 0; e < A_F4; ++e) {
        const int qq = tid + e * kThreads;
        const int row = qq / (BM / 4), c4 = (qq % (BM / 4)) * 4;
        voff_a[e] = (m0 + c4 < p.Cout) ? (unsigned)(row * p.Cout + m0 + c4) * 4u : kOOB;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[A_F4];
    unsigned rb[B_PER];
    // load cursor: valid-tap index / chunk inside the tap, and this thread's byte offset for the tap
    int ld_vi = q0 / cpt, ld_c = q0 - (q0 / cpt) * cpt, ld_tap = 0;
    unsigned voff_b = kOOB;
    const unsigned chan_step = (unsigned)(B_STEP * HW) * 4u;     // bytes between this thread's k rows
    auto set_tap = [&](int vi) {
        ld_tap = __builtin_amdgcn_readfirstlane(vtaps[vi]);       // block-uniform: keep it scalar
        int sp;
        const bool ok = tap_src(ld_tap, sp);
        voff_b = ok ? (img_off + (unsigned)sp) * 4u : kOOB;
    };
    if (nq > 0) set_tap(ld_vi);

    auto load_next = [&]() {                   // issues the loads of the next chunk in sequence
        if (ld_c == cpt) {
            ld_c = 0;
            ++ld_vi;
            set_tap(ld_vi);
        }
        const int k0 = ld_tap * p.Cin + ld_c * BK;
        const unsigned soff_a = (unsigned)(k0 * p.Cout) * 4u;
#pragma unroll
        for (int e = 0; e < A_F4; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_a[e], soff_a, 0);
        const unsigned soff_b = (unsigned)(ld_c * BK * HW) * 4u;
#pragma unroll
        for (int e = 0; e < B_PER; ++e)
            rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, soff_b + e * chan_step, 0);
        ++ld_c;
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < A_F4; ++e) {
            const int qq = tid + e * kThreads;
            *reinterpret_cast<u32x4*>(&As[buf][qq * 4]) = ra[e];
        }
#pragma unroll
        for (int e = 0; e < B_PER; ++e) Bs[buf][(kr + e * B_STEP) * BN + pl] = __builtin_bit_cast(float, rb[e]);
    };

    if (nq > 0) {
        load_next();
        store_tiles(0);
        if (nq > 1) load_next();
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int q = 0; q < nq; ++q) {
        const int cur = q & 1;
        if (q + 1 < nq) store_tiles(cur ^ 1);
        if (q + 2 < nq) load_next();
        const float* Ab = As[cur];
        const float* Bb = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = Ab[(kk + lh) * BM + wm * (MI * 32) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Bb[(kk + lh) * BN + wn * (NI * 32) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    if (p.ksplit > 1) {     // raw partial sums; conv_splitk_reduce_kernel applies the epilogue
        // slabs are indexed by the pixel's position in the whole output (b, oh, ow), not inside the block's column band
        float* sl = p.slab + (long long)blockIdx.z * p.Cout * p.P;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int opix = n0 + wn * (NI * 32) + j * 32 + l31;
            if (opix >= W_P) continue;
            const int ob = opix / OHW;
            const int orr = opix - ob * OHW, ooh = orr / w_own;
            const int gpix = ob * OHOW + ooh * p.OW + w_ow0 + (orr - ooh * w_own);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < p.Cout) sl[(long long)co * p.P + gpix] = acc[i][j][r];
                }
        }
        return;
    }
    // ---- epilogue (same as the generic kernel)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int opix = n0 + wn * (NI * 32) + j * 32 + l31;
        if (opix >= W_P) continue;
        const int ob = opix / OHW;
        const int orr = opix - ob * OHW, ooh = orr / w_own;
        const int orp = ooh * p.OW + w_ow0 + (orr - ooh * w_own);
        float* yb = p.y + (long long)ob * p.y_bs + orp;
        const float* rbp = p.res ? p.res + (long long)ob * p.res_bs + orp : nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co >= p.Cout) continue;
                float v = acc[i][j][r];
                if (p.scale) v *= p.scale[co];
                if (p.shift) v += p.shift[co];
                const long long off = (long long)co * OHOW;
                if (rbp) v += rbp[off];
                if (p.accumulate) v += acc_prev(p, yb + off);
                if (p.relu) v = fmaxf(v, 0.f);
                yb[off] = v;
            }
        }
    }
}

// y = epilogue(sum_z slab[z][co][pix]) for the split-K launches (fixed summation order)
__global__ void conv_splitk_reduce_kernel(ConvP p) {
    const long long total = (long long)p.Cout * p.P;
    const int OHOW = p.OH * p.OW;
    float vmax = 0.f;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx / p.P), pix = (int)(idx - (long long)co * p.P);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.slab[(long long)z * total + idx];
        const int ob = pix / OHOW, orp = pix - ob * OHOW;
        if (p.scale) v *= p.scale[co];
        if (p.shift) v += p.shift[co];
        const long long off = (long long)co * OHOW + orp;
        if (p.res) v += p.res[(long long)ob * p.res_bs + off];
        float* y = p.y + (long long)ob * p.y_bs + off;
        if (p.accumulate) v += acc_prev(p, y);
        if (p.relu) v = fmaxf(v, 0.f);
        *y = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    if (p.y_amax) publish_amax(vmax, p.y_amax);      // (uniform branch; every thread of the workgroup reaches it)
}

// 16-byte form of the above for OH*OW % 4 == 0 (and 4-float-aligned batch strides): four pixels per thread
__global__ void conv_splitk_reduce_vec4_kernel(ConvP p) {
    const int OHOW = p.OH * p.OW;
    float vmax = 0.f;
    const long long total = (long long)p.Cout * p.P, total4 = total / 4;
    const float4* slab4 = reinterpret_cast<const float4*>(p.slab);
    for (long long i4 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i4 < total4;
         i4 += (long long)gridDim.x * blockDim.x) {
        const long long idx = i4 * 4;
        const int co = (int)(idx / p.P), pix = (int)(idx - (long long)co * p.P);
        float4 v = slab4[i4];
        for (int z = 1; z < p.ksplit; ++z) {
            const float4 t = slab4[(long long)z * total4 + i4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int ob = pix / OHOW, orp = pix - ob * OHOW;
        if (p.scale) { const float sc = p.scale[co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        if (p.shift) { const float sh = p.shift[co]; v.x += sh; v.y += sh; v.z += sh; v.w += sh; }
        const long long off = (long long)co * OHOW + orp;
        if (p.res) {
            const float4 r = *reinterpret_cast<const float4*>(p.res + (long long)ob * p.res_bs + off);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        float4* y = reinterpret_cast<float4*>(p.y + (long long)ob * p.y_bs + off);
        if (p.accumulate) {
            float4 o = *y;
            if (p.acc_mask) {
                const long long e = reinterpret_cast<const float*>(y) - p.acc_base;      // a multiple of 4
                const unsigned nb = ((unsigned)p.acc_mask[e >> 3] >> (unsigned)(e & 4)) & 0xFu;
                if (!(nb & 1u)) o.x = 0.f;
                if (!(nb & 2u)) o.y = 0.f;
                if (!(nb & 4u)) o.z = 0.f;
                if (!(nb & 8u)) o.w = 0.f;
            }
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *y = v;
        vmax = amax4(vmax, v);
    }
    if (p.y_amax) publish_amax(vmax, p.y_amax);      // the split-K path's amax: no separate read pass over the output
}

// ---------------------------------------------------------------------------------------------
struct WgradP {
    const float* x;
    const float* dy;
    float* slab;  // [S][Cout][N]
    int B, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, dil;
    int N;        // KH*KW*Cin
    int P;        // B*OH*OW
    int chunks_per_split;
    long long x_bs, dy_bs;
    unsigned x_bytes, dy_bytes;   // extents for the buffer-descriptor fast path (0 = not eligible)
    int ow0, own;                 // output-column window (fast kernel): P = B*OH*own
    int slab0;                    // first slab index of this launch
    int xcd_order;                // split32 kernel: 0 launch order, 1 XCD-contiguous (slab, row tile, N tile) order, 2 + 2x2 blocks
    int nb, splits;               // column bands inside one launch: blockIdx.z = band * splits + split
    int b_ow0[4], b_own[4], b_cps[4];   // per band: window and 32-pixel chunks per split
    const float* x_amax;                // split kernel, fp16x2 arithmetic: device scalar >= max|x|
    int xa_stride, da_stride;           // CS kernels: x_amax / dy_amax are read at [channel * stride] (0: one scale for the tensor, 1: per channel)
};

#include "conv_split.h"

// Same pipeline as the forward kernel (LDS double buffer, one barrier per 32-pixel chunk).  When the
// block's N tile lies inside one tap, pixel chunks for which that tap reads only padding are skipped
// (decided per wave with a ballot: every wave sees the same 32 pixels, so the decision is block-uniform).
template <int BM, int BN, int WM>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(WgradP p) {
    constexpr int BK = 32, LD = BK + 1;
    constexpr int WN = 4 / WM, MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int A_PER = BM / 8, B_PER = BN / 8;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN, "bad tile");

    extern __shared__ __attribute__((aligned(16))) float wg_lds[];   // As[2][BM*LD] | Bs[2][BN*LD]
    float* const As0 = wg_lds;
    float* const Bs0 = wg_lds + 2 * BM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int OHOW = p.OH * p.OW, HW = p.H * p.W;
    const int px = tid & 31, row0 = tid >> 5;

    // fixed per-thread row decode of the B tile: n -> (tap, ci) -> (dh, dw, channel offset)
    int b_shift[B_PER];   // (dh << 16) | (dw & 0xffff), dh/dw = tap offset in input pixels
    int b_coff[B_PER];    // ci*H*W, or -1 when n >= N
#pragma unroll
    for (int e = 0; e < B_PER; ++e) {
        const int n = n0 + row0 + 8 * e;
        if (n < p.N) {
            const int tap = n / p.Cin, ci = n - tap * p.Cin;
            const int ti = tap / p.KW, tj = tap - ti * p.KW;
            const int dh = ti * p.dil - p.pad, dw = tj * p.dil - p.pad;
            b_shift[e] = (int)(((unsigned)dh << 16) | ((unsigned)dw & 0xffffu));
            b_coff[e] = ci * HW;
        } else {
            b_shift[e] = 0;
            b_coff[e] = -1;
        }
    }
    // block-uniform tap of the tile (if it has one)
    const int n_last = min(n0 + BN, p.N) - 1;
    const int tap0 = n0 / p.Cin;
    const bool single_tap = tap0 == n_last / p.Cin;
    const int t_dh = (tap0 / p.KW) * p.dil - p.pad, t_dw = (tap0 % p.KW) * p.dil - p.pad;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[A_PER], rb[B_PER];
    const int chunk_begin = blockIdx.z * p.chunks_per_split;
    const int total_chunks = (p.P + BK - 1) / BK;
    const int chunk_end = min(chunk_begin + p.chunks_per_split, total_chunks);

    // first chunk >= c (and < chunk_end) whose pixels are not all padding for this tile's tap
    auto next_valid = [&](int c) {
        if (!single_tap) return min(c, chunk_end);
        for (; c < chunk_end; ++c) {
            const int pix = c * BK + px;
            bool ok = pix < p.P;
            if (ok) {
                const int pb = pix / OHOW, rp = pix - pb * OHOW;
                const int oh = rp / p.OW, ow = rp - oh * p.OW;
                const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
                ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            }
            if (__any(ok)) break;
        }
        return c;
    };

    auto load_tiles = [&](int chunk) {
        const int pix = chunk * BK + px;
        const bool ok = pix < p.P;
        int pb = 0, oh = 0, ow = 0, rp = 0;
        if (ok) {
            pb = pix / OHOW;
            rp = pix - pb * OHOW;
            oh = rp / p.OW;
            ow = rp - oh * p.OW;
        }
        const float* dyb = p.dy + (long long)pb * p.dy_bs + rp;
#pragma unroll
        for (int e = 0; e < A_PER; ++e) {
            const int co = m0 + row0 + 8 * e;
            ra[e] = (ok && co < p.Cout) ? dyb[(long long)co * OHOW] : 0.f;
        }
        const float* xb = p.x + (long long)pb * p.x_bs;
        const int bh = oh * p.stride, bw = ow * p.stride;
        if (single_tap) {
            // one validity test / spatial offset for the whole tile column (all rows share the tap)
            const int ih = bh + t_dh, iw = bw + t_dw;
            const bool v = ok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const float* src = xb + ih * p.W + iw;
#pragma unroll
            for (int e = 0; e < B_PER; ++e) rb[e] = (v && b_coff[e] >= 0) ? src[b_coff[e]] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < B_PER; ++e) {
                const int ih = bh + (b_shift[e] >> 16);
                const int iw = bw + (int)(short)(b_shift[e] & 0xffff);
                const bool v = ok && b_coff[e] >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                rb[e] = v ? xb[(long long)b_coff[e] + ih * p.W + iw] : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* A = As0 + buf * BM * LD;
        float* Bq = Bs0 + buf * BN * LD;
#pragma unroll
        for (int e = 0; e < A_PER; ++e) A[(row0 + 8 * e) * LD + px] = ra[e];
#pragma unroll
        for (int e = 0; e < B_PER; ++e) Bq[(row0 + 8 * e) * LD + px] = rb[e];
    };

    int c0 = next_valid(chunk_begin), c1 = chunk_end;
    if (c0 < chunk_end) {
        load_tiles(c0);
        store_tiles(0);
        c1 = next_valid(c0 + 1);
        if (c1 < chunk_end) load_tiles(c1);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    int cur = 0;
    while (c0 < chunk_end) {
        if (c1 < chunk_end) store_tiles(cur ^ 1);
        const int c2 = c1 < chunk_end ? next_valid(c1 + 1) : chunk_end;
        if (c2 < chunk_end) load_tiles(c2);
        const float* A = As0 + cur * BM * LD;
        const float* Bq = Bs0 + cur * BN * LD;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = A[(wm * (MI * 32) + i * 32 + l31) * LD + kk + lh];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Bq[(wn * (NI * 32) + j * 32 + l31) * LD + kk + lh];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        c0 = c1;
        c1 = c2;
        cur ^= 1;
    }

    float* slab = p.slab + (long long)blockIdx.z * p.Cout * p.N;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (NI * 32) + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) slab[(long long)co * p.N + n] = acc[i][j][r];
            }
    }
}

// Fast weight-gradient path (Cout % BM == 0, Cin % BN == 0 so every tile sits inside one tap, tensors
// < 2 GiB): buffer-descriptor loads, per-lane offsets recomputed once per 32-pixel chunk, row advance in
// the scalar offset, padding via out-of-range offsets.  Same tiling, pipeline and chunk skipping as
// conv_wgrad_kernel.
template <int BM, int BN, int WM, int BK>
__global__ __launch_bounds__(kThreads, 3) void conv_wgrad_fast_kernel(WgradP p) {
    constexpr int LD = BK + 1;
    constexpr int WN = 4 / WM, MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int RPP = kThreads / BK;                    // rows covered per pass
    constexpr int A_PER = BM / RPP, B_PER = BN / RPP;
    constexpr unsigned kOOB = 0x80000000u;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN, "bad tile");

    extern __shared__ __attribute__((aligned(16))) float wg_lds[];   // As[2][BM*LD] | Bs[2][BN*LD]
    float* const As0 = wg_lds;
    float* const Bs0 = wg_lds + 2 * BM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    // column bands: blockIdx.z = band * splits + split (scalar selects keep the window in SGPRs)
    int zsplit = blockIdx.z, w_ow0 = p.ow0, w_own = p.own, w_cps = p.chunks_per_split;
    if (p.nb > 1) {
        const int band = (int)blockIdx.z / p.splits;
        zsplit = (int)blockIdx.z - band * p.splits;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i == band) {
                w_ow0 = p.b_ow0[i];
                w_own = p.b_own[i];
                w_cps = p.b_cps[i];
            }
    }
    const int W_P = p.nb > 1 ? p.B * p.OH * w_own : p.P;
    const int OHOW = p.OH * p.OW, HW = p.H * p.W;
    const int px = tid % BK, row0 = tid / BK;

    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);

    const int tap0 = n0 / p.Cin, ci0 = n0 - tap0 * p.Cin;
    const int t_dh = (tap0 / p.KW) * p.dil - p.pad, t_dw = (tap0 % p.KW) * p.dil - p.pad;
    const unsigned a_row = (unsigned)((m0 + row0) * OHOW);       // elements
    const unsigned b_row = (unsigned)((ci0 + row0) * HW);
    const unsigned a_step = (unsigned)(RPP * OHOW) * 4u, b_step = (unsigned)(RPP * HW) * 4u;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned ra[A_PER], rb[B_PER];
    // p.chunks_per_split counts 32-pixel chunks; this kernel walks BK-pixel chunks
    const int chunk_begin = zsplit * w_cps * (32 / BK);
    const int total_chunks = (W_P + BK - 1) / BK;
    const int chunk_end = min(chunk_begin + w_cps * (32 / BK), total_chunks);

    // decode this thread's pixel of chunk c: byte offsets into dy / x (kOOB when padding / past the end)
    const bool row_chunks = (w_own % BK) == 0;     // a chunk never crosses an output row: decode it in scalars
    const int OHW = p.OH * w_own;
    auto decode = [&](int c, unsigned& va, unsigned& vb) {
        va = kOOB;
        vb = kOOB;
        if (row_chunks) {
            // c is block-uniform, so row / image index and the row's validity under the tap are SALU work;
            // per lane only the column remains (P is a multiple of OW, so no ragged tail)
            const int first = c * BK;
            const int grow = first / w_own, ow = w_ow0 + first - grow * w_own + px;
            const int pb = grow / p.OH, oh = grow - pb * p.OH;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            const unsigned img_a = (unsigned)((long long)pb * p.dy_bs) + (unsigned)(oh * p.OW);
            va = (img_a + (unsigned)ow + a_row) * 4u;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                vb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W) + (unsigned)iw + b_row) * 4u;
            return;
        }
        const int pix = c * BK + px;
        if (pix < W_P) {
            const int pb = pix / OHW, rr = pix - pb * OHW;
            const int oh = rr / w_own, ow = w_ow0 + (rr - oh * w_own);
            const int rp = oh * p.OW + ow;
            va = ((unsigned)((long long)pb * p.dy_bs) + (unsigned)rp + a_row) * 4u;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                vb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W + iw) + b_row) * 4u;
        }
    };
    unsigned nva = kOOB, nvb = kOOB;            // offsets of the chunk found by next_valid
    auto next_valid = [&](int c) {
        for (; c < chunk_end; ++c) {
            decode(c, nva, nvb);
            // every wave holds all BK pixels of the chunk (64 / BK copies), so the ballot is block-uniform
            if (__any(nvb != kOOB)) break;
        }
        return c;
    };
    auto load_tiles = [&]() {                    // loads the chunk last returned by next_valid
#pragma unroll
        for (int e = 0; e < A_PER; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b32(rdy, nva, e * a_step, 0);
#pragma unroll
        for (int e = 0; e < B_PER; ++e) rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, nvb, e * b_step, 0);
    };
    auto store_tiles = [&](int buf) {
        float* A = As0 + buf * BM * LD;
        float* Bq = Bs0 + buf * BN * LD;
#pragma unroll
        for (int e = 0; e < A_PER; ++e) A[(row0 + RPP * e) * LD + px] = __builtin_bit_cast(float, ra[e]);
#pragma unroll
        for (int e = 0; e < B_PER; ++e) Bq[(row0 + RPP * e) * LD + px] = __builtin_bit_cast(float, rb[e]);
    };

    int c0 = next_valid(chunk_begin), c1 = chunk_end;
    if (c0 < chunk_end) {
        load_tiles();
        store_tiles(0);
        c1 = next_valid(c0 + 1);
        if (c1 < chunk_end) load_tiles();
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    int cur = 0;
    while (c0 < chunk_end) {
        if (c1 < chunk_end) store_tiles(cur ^ 1);
        const int c2 = c1 < chunk_end ? next_valid(c1 + 1) : chunk_end;
        if (c2 < chunk_end) load_tiles();
        const float* A = As0 + cur * BM * LD;
        const float* Bq = Bs0 + cur * BN * LD;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = A[(wm * (MI * 32) + i * 32 + l31) * LD + kk + lh];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Bq[(wn * (NI * 32) + j * 32 + l31) * LD + kk + lh];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        c0 = c1;
        c1 = c2;
        cur ^= 1;
    }

    float* slab = p.slab + (long long)(p.slab0 + blockIdx.z) * p.Cout * p.N;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (NI * 32) + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)co * p.N + n] = acc[i][j][r];
            }
    }
}

// slab[z][co][tap*Cin+ci] summed over z in order -> dw[co][ci][tap]
// live: bit t set = tap t has slab data (taps that read only padding everywhere are not computed by the split kernel)
// Each reduction is a device body of (its arguments, which of `nblocks` 256-thread blocks this is): the per-layer kernels below
// run one body on their own grid, wgrad_reduce_multi_kernel (round 6) runs the bodies of MANY layers in one launch - the same
// sums in the same order, so the two are bit-identical.
__device__ __forceinline__ void wgrad_reduce_plain_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int Cout, int Cin,
                                                        int T, int accumulate, unsigned long long live, long long lb, long long nblocks) {
    const long long total = (long long)Cout * Cin * T;
    for (long long idx = lb * 256 + threadIdx.x; idx < total; idx += nblocks * 256) {
        const int N = Cin * T;
        const int co = (int)(idx / N), n = (int)(idx - (long long)co * N);
        const int tap = n / Cin, ci = n - tap * Cin;
        float s = 0.f;
        if (tap >= 64 || ((live >> tap) & 1ull))
            for (int z = 0; z < S; ++z) s += slab[(long long)z * total + idx];
        const long long o = ((long long)co * Cin + ci) * T + tap;
        dw[o] = accumulate ? dw[o] + s : s;
    }
}
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S,
                                    int Cout, int Cin, int T, int accumulate, unsigned long long live) {
    wgrad_reduce_plain_body(slab, dw, S, Cout, Cin, T, accumulate, live, blockIdx.x, gridDim.x);
}

// The same reduction for MANY slabs of a small matrix (the 7x7 stem: 64 x 147 values in 256 pixel slabs): four waves
// share 64 consecutive outputs, wave w adds slabs w, w + 4, ... and the four partial sums are combined in wave order -
// a fixed order again, a quarter of the dependent chain and four times the loads in flight (62 -> 20 us).
__device__ __forceinline__ void wgrad_reduce_many_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int Cout, int Cin,
                                                       int T, int accumulate, unsigned long long live, long long lb, float* part /* [4][64] */) {
    const long long total = (long long)Cout * Cin * T;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long idx = lb * 64 + l;
    const int N = Cin * T;
    int co = 0, tap = 0, ci = 0;
    float s = 0.f;
    if (idx < total) {
        co = (int)(idx / N);
        const int n = (int)(idx - (long long)co * N);
        tap = n / Cin;
        ci = n - tap * Cin;
        if (tap >= 64 || ((live >> tap) & 1ull))
            for (int z = w; z < S; z += 4) s += slab[(long long)z * total + idx];
    }
    part[w * 64 + l] = s;
    __syncthreads();
    if (w == 0 && idx < total) {
        const float r = ((part[l] + part[64 + l]) + part[128 + l]) + part[192 + l];
        const long long o = ((long long)co * Cin + ci) * T + tap;
        dw[o] = accumulate ? dw[o] + r : r;
    }
}
__global__ void wgrad_reduce_many_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S,
                                         int Cout, int Cin, int T, int accumulate, unsigned long long live) {
    __shared__ float part[4 * 64];
    wgrad_reduce_many_body(slab, dw, S, Cout, Cin, T, accumulate, live, blockIdx.x, part);
}

// MANY slabs of a SMALL matrix (round 6: the stem's 64 x 147 values in 768 slabs; 64 outputs per block would be 147 blocks whose
// waves each walk 192 slabs one dependent-looking load after the other - that reduction, not the matrix kernel, was most of the
// stem weight gradient's 155 us).  16 outputs per block and 16 slab lanes: thread (o, zl) adds slabs zl, zl + 16, ... in four
// independent chains, the 16 lane sums are combined in lane order - a fixed order again.
__device__ __forceinline__ void wgrad_reduce_many16_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int Cout, int Cin,
                                                         int T, int accumulate, unsigned long long live, long long lb, float* sm /* [16 * 17] */) {
    const long long total = (long long)Cout * Cin * T;
    const int o = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long long idx = lb * 16 + o;
    const int N = Cin * T;
    int co = 0, tap = 0, ci = 0;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
        co = (int)(idx / N);
        const int n = (int)(idx - (long long)co * N);
        tap = n / Cin;
        ci = n - tap * Cin;
        if (tap >= 64 || ((live >> tap) & 1ull)) {
            int z = zl;
            for (; z + 48 < S; z += 64) {
                s0 += slab[(long long)z * total + idx];
                s1 += slab[(long long)(z + 16) * total + idx];
                s2 += slab[(long long)(z + 32) * total + idx];
                s3 += slab[(long long)(z + 48) * total + idx];
            }
            for (; z < S; z += 16) s0 += slab[(long long)z * total + idx];
        }
    }
    sm[zl * 17 + o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zl == 0 && idx < total) {
        float r = sm[o];
#pragma unroll
        for (int j = 1; j < 16; ++j) r += sm[j * 17 + o];
        const long long out = ((long long)co * Cin + ci) * T + tap;
        dw[out] = accumulate ? dw[out] + r : r;
    }
}
__global__ void wgrad_reduce_many16_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S,
                                           int Cout, int Cin, int T, int accumulate, unsigned long long live) {
    __shared__ float sm[16 * 17];
    wgrad_reduce_many16_body(slab, dw, S, Cout, Cin, T, accumulate, live, blockIdx.x, sm);
}

// T == 1 (1x1 kernels: slab layout == dw layout): 16 bytes per thread, four slabs in flight per step.  The sum
// order is fixed (pairs of pairs), so the result stays bitwise reproducible.
__device__ __forceinline__ float4 wgrad_sum_slabs4(const float4* __restrict__ slab, int S, long long total4, long long idx) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= S; z += 4) {
        const float4 a = slab[(long long)z * total4 + idx], b = slab[(long long)(z + 1) * total4 + idx];
        const float4 c = slab[(long long)(z + 2) * total4 + idx], d = slab[(long long)(z + 3) * total4 + idx];
        s.x += (a.x + b.x) + (c.x + d.x);
        s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z);
        s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; z < S; ++z) {
        const float4 a = slab[(long long)z * total4 + idx];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    return s;
}
__device__ __forceinline__ void wgrad_reduce_vec4_body(const float4* __restrict__ slab, float4* __restrict__ dw, int S, long long total4,
                                                       int accumulate, long long lb, long long nblocks) {
    for (long long idx = lb * 256 + threadIdx.x; idx < total4; idx += nblocks * 256) {
        float4 s = wgrad_sum_slabs4(slab, S, total4, idx);
        if (accumulate) {
            const float4 o = dw[idx];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        dw[idx] = s;
    }
}
__global__ void wgrad_reduce_vec4_kernel(const float4* __restrict__ slab, float4* __restrict__ dw, int S,
                                         long long total4, int accumulate) {
    wgrad_reduce_vec4_body(slab, dw, S, total4, accumulate, blockIdx.x, gridDim.x);
}

// Same sum for 2 <= T <= 16 with both sides coalesced: a block owns (co, 32 input channels); each tap's 32
// slab values are read as one 128-byte run, the (ci,tap) transpose happens in LDS, and the 32*T results leave
// as one contiguous run of dw.  256 threads as (32, 8); block = (Cin/32 tile, co).
__device__ __forceinline__ void wgrad_reduce_tiled_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int Cout, int Cin,
                                                        int T, int accumulate, unsigned long long live, int ci_tile, int co,
                                                        float* tile /* [32 * 16] */) {
    const int ci0 = ci_tile * 32;
    const int lane = threadIdx.x & 31, row = threadIdx.x >> 5;
    const long long total = (long long)Cout * Cin * T;
    const long long N = (long long)Cin * T;
    for (int tap = row; tap < T; tap += 8) {
        float s = 0.f;
        if (ci0 + lane < Cin && ((live >> tap) & 1ull)) {
            const long long idx = (long long)co * N + (long long)tap * Cin + ci0 + lane;
            int z = 0;
            for (; z + 4 <= S; z += 4)      // four slabs in flight; fixed (pairs of pairs) order
                s += (slab[(long long)z * total + idx] + slab[(long long)(z + 1) * total + idx]) +
                     (slab[(long long)(z + 2) * total + idx] + slab[(long long)(z + 3) * total + idx]);
            for (; z < S; ++z) s += slab[(long long)z * total + idx];
        }
        tile[lane * T + tap] = s;
    }
    __syncthreads();
    const int nvalid = min(32, Cin - ci0) * T;
    float* out = dw + ((long long)co * Cin + ci0) * T;
    for (int j = row * 32 + lane; j < nvalid; j += 256) out[j] = accumulate ? out[j] + tile[j] : tile[j];
}
__global__ void wgrad_reduce_tiled_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S,
                                          int Cout, int Cin, int T, int accumulate, unsigned long long live) {
    __shared__ float tile[32 * 16];
    wgrad_reduce_tiled_body(slab, dw, S, Cout, Cin, T, accumulate, live, blockIdx.x, blockIdx.y, tile);
}

// Role-swapped 1x1 weight gradients (wgrad_role_swap): the slabs hold dW^T [Cin][Cout]; dw[co][ci] (+)= sum_z slab[z][ci][co]
// with the slabs added in wgrad_reduce_vec4_kernel's order (what the two-launch form - that kernel into a scratch matrix, then
// transpose_add_kernel - computes): 32 x 32 tiles through LDS, both sides in 128-byte runs.  256 threads as (32, 8).
__device__ __forceinline__ void wgrad_reduce_transposed_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int Cout,
                                                             int Cin, int accumulate, int co_tile, int ci_tile, float* tile /* [32 * 33] */) {
    const int co0 = co_tile * 32, ci0 = ci_tile * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long total = (long long)Cout * Cin;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        float s = 0.f;
        if (ci < Cin && co < Cout) {
            const long long idx = (long long)ci * Cout + co;
            int z = 0;
            for (; z + 4 <= S; z += 4)
                s += (slab[(long long)z * total + idx] + slab[(long long)(z + 1) * total + idx]) +
                     (slab[(long long)(z + 2) * total + idx] + slab[(long long)(z + 3) * total + idx]);
            for (; z < S; ++z) s += slab[(long long)z * total + idx];
        }
        tile[r * 33 + tx] = s;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) {
            const long long o = (long long)co * Cin + ci;
            dw[o] = accumulate ? dw[o] + tile[tx * 33 + r] : tile[tx * 33 + r];
        }
    }
}

// The slab reductions of MANY weight gradients in one launch (round 6; wsdl_wgrad_reduce_multi).  A training step ran ~60 of
// the kernels above, one behind each weight-gradient launch on the side stream: 9-10 us each for a few MB - launch and tail
// latency, not bandwidth (0.69 ms per step).  With wsdl_conv2d_wgrad_deferred the weight-gradient launches leave their slabs
// in place and hand back a descriptor; the caller runs ALL pending reductions as one grid when the gradients are needed (the
// optimiser step, a gradient bucket's all-reduce).  Block b belongs to the last entry whose block_begin <= b.
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const wsdl_wgrad_reduce_desc* __restrict__ d, int n) {
    __shared__ float sm[32 * 33];
    const int b = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].block_begin <= b) lo = mid; else hi = mid - 1;
    }
    const wsdl_wgrad_reduce_desc q = d[lo];
    const int lb = b - q.block_begin;
    switch (q.kind) {
        case WSDL_WGRAD_REDUCE_TILED:
            wgrad_reduce_tiled_body(q.slab, q.dw, q.S, q.Cout, q.Cin, q.T, q.accumulate, q.live, lb % q.grid_x, lb / q.grid_x, sm);
            break;
        case WSDL_WGRAD_REDUCE_VEC4:
            wgrad_reduce_vec4_body(reinterpret_cast<const float4*>(q.slab), reinterpret_cast<float4*>(q.dw), q.S,
                                   (long long)q.Cout * q.Cin * q.T / 4, q.accumulate, lb, q.nblocks);
            break;
        case WSDL_WGRAD_REDUCE_MANY:
            wgrad_reduce_many_body(q.slab, q.dw, q.S, q.Cout, q.Cin, q.T, q.accumulate, q.live, lb, sm);
            break;
        case WSDL_WGRAD_REDUCE_TRANSPOSED:
            wgrad_reduce_transposed_body(q.slab, q.dw, q.S, q.Cout, q.Cin, q.accumulate, lb % q.grid_x, lb / q.grid_x, sm);
            break;
        case WSDL_WGRAD_REDUCE_MANY16:
            wgrad_reduce_many16_body(q.slab, q.dw, q.S, q.Cout, q.Cin, q.T, q.accumulate, q.live, lb, sm);
            break;
        default:
            wgrad_reduce_plain_body(q.slab, q.dw, q.S, q.Cout, q.Cin, q.T, q.accumulate, q.live, lb, q.nblocks);
            break;
    }
}

// w[co][ci][tap] -> fwd[(tap*Cin+ci)][co], dgrad[(tap*Cout+co)][ci]   (generic, uncoalesced reads)
__global__ void prep_weights_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                    float* __restrict__ dg, int Cout, int Cin, int T) {
    const long long total = (long long)Cout * Cin * T;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates the fwd layout (co fastest) so the fwd write is coalesced
        const int co = (int)(idx % Cout);
        const long long k = idx / Cout;
        const int ci = (int)(k % Cin), tap = (int)(k / Cin);
        const float v = w[((long long)co * Cin + ci) * T + tap];
        if (fwd) fwd[idx] = v;
        if (dg) dg[((long long)tap * Cout + co) * Cin + ci] = v;
    }
}

// Tiled version for T <= 9: a block stages w[co0..co0+31][ci0..ci0+31][all taps] (32 contiguous runs of 32*T
// floats) in LDS and writes both layouts with 128-byte runs.  blockDim = (32, 8).
template <int TMAX>
__global__ void prep_weights_tiled_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                          float* __restrict__ dg, int Cout, int Cin, int T) {
    constexpr int LDT = 32 * TMAX + 1;
    __shared__ float tile[32 * LDT];
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int lane = threadIdx.x, row = threadIdx.y;
    const int nci = min(32, Cin - ci0), nco = min(32, Cout - co0);
    const int run = nci * T;
    for (int r = row; r < nco; r += 8) {
        const float* src = w + ((long long)(co0 + r) * Cin + ci0) * T;
        for (int j = lane; j < run; j += 32) tile[r * LDT + j] = src[j];
    }
    __syncthreads();
    if (fwd) {
        // rows (tap, ci), 32 consecutive co per row
        for (int q = row; q < nci * T; q += 8) {
            const int tap = q / nci, cil = q - tap * nci;
            if (lane < nco)
                fwd[((long long)tap * Cin + ci0 + cil) * Cout + co0 + lane] = tile[lane * LDT + cil * T + tap];
        }
    }
    if (dg) {
        // rows (tap, co), 32 consecutive ci per row
        for (int q = row; q < nco * T; q += 8) {
            const int tap = q / nco, col = q - tap * nco;
            if (lane < nci)
                dg[((long long)tap * Cout + co0 + col) * Cin + ci0 + lane] = tile[col * LDT + lane * T + tap];
        }
    }
}

__global__ void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int C,
                                 int HW, long long dy_bs, int accumulate) {
    __shared__ float sm[16];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* src = dy + (long long)b * dy_bs + (long long)c * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += src[i];
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

int check_geom(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
               int* OH, int* OW) {
    WSDL_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv: non-positive dimension");
    WSDL_REQUIRE(kh > 0 && kw > 0 && kh == kw && kh <= 15, "conv: kernel must be square, <= 15 (got %dx%d)", kh, kw);
    WSDL_REQUIRE(stride >= 1 && stride <= 8 && dil >= 1 && pad >= 0, "conv: bad stride/pad/dilation");
    const int oh = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
    const int ow = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    WSDL_REQUIRE(oh > 0 && ow > 0, "conv: empty output (%d x %d)", oh, ow);
    WSDL_REQUIRE((long long)B * oh * ow < (1ll << 31) && (long long)B * H * W < (1ll << 31),
                 "conv: pixel count exceeds int32");
    WSDL_REQUIRE((long long)Cin * H * W < (1ll << 31) && (long long)Cout * oh * ow < (1ll << 31),
                 "conv: per-image tensor exceeds int32 elements");
    WSDL_REQUIRE((long long)kh * kw * Cin < (1ll << 31) && (long long)kh * kw * Cout < (1ll << 31), "conv: K too large");
    WSDL_REQUIRE(dil * (kh - 1) < 32768 && pad < 32768, "conv: tap offset exceeds int16");
    *OH = oh;
    *OW = ow;
    return WSDL_OK;
}

// Fraction of the nominal K-chunks a launch really executes (profiling only): mirrors the kernels' skip
// rule - a tap is skipped for a pixel tile when every pixel of the tile reads padding under that tap.
// Column bands: the output columns split where the set of in-range column taps changes (ASPP dilation 12 on a
// 32-wide map: [0,12) [12,20) [20,32)).  Launching each band separately makes tap validity uniform along a
// tile's columns, so the tile-level skip also removes the column padding taps.  Only when every band is >= 8
// pixels wide and the stride-1 / no-subsampling gather applies.
struct Band { int ow0, own; };
int column_bands(int OW, int W, int ah, int bh, int ch, int sh, int KW, Band* out /* >= 8 */) {
    out[0] = Band{0, OW};
    if (sh != 1 || ah != 1 || KW < 2 || KW > 3 || (bh < 8 && bh > -8)) return 1;
    int cuts[8], nc = 0;
    for (int j = 0; j < KW; ++j) {
        const int o = j * bh + ch;                 // iw = ow + o
        const int lo = -o, hi = W - o;             // valid: lo <= ow < hi
        if (lo > 0 && lo < OW) cuts[nc++] = lo;
        if (hi > 0 && hi < OW) cuts[nc++] = hi;
    }
    if (nc == 0) return 1;
    std::sort(cuts, cuts + nc);
    nc = (int)(std::unique(cuts, cuts + nc) - cuts);
    if (nc > 3) return 1;
    int prev = 0, n = 0;
    Band tmp[8];
    for (int i = 0; i <= nc; ++i) {
        const int end = i < nc ? cuts[i] : OW;
        if (end - prev < 8) return 1;
        tmp[n++] = Band{prev, end - prev};
        prev = end;
    }
    for (int i = 0; i < n; ++i) out[i] = tmp[i];
    return n;
}

double igemm_executed_fraction(const ConvP& p, int BN) {
    const int T = p.KH * p.KW, OHW = p.OH * p.own;
    long long done = 0, all = 0;
    for (int n0 = 0; n0 < p.P; n0 += BN) {
        for (int t = 0; t < T; ++t) {
            const int ti = t / p.KW, tj = t % p.KW;
            bool any = false;
            for (int pix = n0; pix < n0 + BN && pix < p.P && !any; ++pix) {
                const int r = pix % OHW, oh = r / p.own, ow = p.ow0 + r % p.own;
                const int nh = oh * p.ah + ti * p.bh + p.ch, nw = ow * p.ah + tj * p.bh + p.ch;
                if (nh < 0 || nw < 0 || nh % p.sh || nw % p.sh) continue;
                any = nh / p.sh < p.H && nw / p.sh < p.W;
            }
            done += any;
            ++all;
        }
    }
    return all ? (double)done / (double)all : 1.0;
}

double wgrad_executed_fraction(int P, int OH, int ow0, int own, int H, int W, int KH, int KW, int stride, int pad,
                               int dil, int BK) {
    const int T = KH * KW, OHW = OH * own;
    long long done = 0, all = 0;
    for (int c0 = 0; c0 < P; c0 += BK) {
        for (int t = 0; t < T; ++t) {
            const int dh = (t / KW) * dil - pad, dw = (t % KW) * dil - pad;
            bool any = false;
            for (int pix = c0; pix < c0 + BK && pix < P && !any; ++pix) {
                const int r = pix % OHW, ih = (r / own) * stride + dh, iw = (ow0 + r % own) * stride + dw;
                any = ih >= 0 && ih < H && iw >= 0 && iw < W;
            }
            done += any;
            ++all;
        }
    }
    return all ? (double)done / (double)all : 1.0;
}

wsdl::Opt g_tile_threshold{400};   // blocks below which the half-size pixel tile is used
wsdl::Opt g_xcd_rowfast{0};               // experiment: row-tile-fastest XCD order for every XCD-mapped launch of the split kernels
wsdl::Opt g_ms_rowfast{1}, g_ms_py{0};   // multi-source launches: row-tile-fastest XCD order; forced number of row groups (0 = auto)
wsdl::Opt g_group_interleave{1};          // grouped forward: stream-interleaved workgroup order when the streams fill the XCDs evenly
wsdl::Opt g_group_tps10{45};       // "group_tps10": taps per K slice of a grouped forward launch, in tenths.  20 (round 5's default): d12 in 4 slices,
                              // d24 in 2 = eight streams, one per XCD; 45 (round 6): d12 in 2 slices, the others whole - 67 MB of slabs less, problem-major
                              // order.  Round 5: 648 / 623 / 636 us at 20 / 30 / 45 on one box, 662 / 713 at 20 / 30 on another; round 6 (image-major
                              // tile order on / off): 638.6 / 634.2 at 20, 575.7 / 601.0 at 45, 1058 at 60 (d12 whole: workgroups of 9 taps)
constexpr int kNumCU = 256, kLdsPerCU = 160 * 1024;

wsdl::Opt g_wgrad_bk{16};   // pixel chunk of the fast weight-gradient kernel: 16 or 32
wsdl::Opt g_bk32{1};        // K-chunk of 32 for the small-tile configurations (half the barriers per MFMA)

template <int BM, int BN, int WM, int BK>
int launch_fast(const ConvP& p, hipStream_t s, dim3 grid) {
    grid.z = p.ksplit > 1 ? p.ksplit : 1;
    constexpr size_t static_lds = 2 * BK * (BM + BN) * sizeof(float) + 64 * sizeof(int);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_fast_kernel<BM, BN, WM, BK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCU - (int)static_lds - 2048);
    });
    WSDL_TRACE("fp32_fast<%d,%d,%d> ks=%d nb=%d grid=%ux%ux%u", BM, BN, BK, p.ksplit, p.nb, grid.x, grid.y, grid.z);
    hipLaunchKernelGGL((conv_igemm_fast_kernel<BM, BN, WM, BK>), grid, dim3(kThreads), 0, s, p);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

// bf16x3-split kernels (conv_split.h): which (rows, K-channels, taps) shapes use them.  A function of the
// weight shape alone, so the layout kernel and the convolution agree on what the layout buffer holds.
wsdl::Opt g_conv_split{1};
wsdl::Opt g_conv_arith{1};    // arithmetic of the split kernels: 1 = fp16x2 (three MFMAs per product, per-tensor power-of-two
                         // scales), 0 = bf16x3 (six MFMAs, no scales) - see conv_split.h
wsdl::Opt g_split_bk32{1};    // K chunk 32 on the small-tile split configurations
bool split_eligible(int rows, int kc, int T) {
    return g_conv_split && kc % 16 == 0 && T <= 9 && split_layout_bytes(g_conv_arith, (long long)T * kc, rows) < (1ll << 31);
}

wsdl::Opt g_tile_img_major{1};   // pixel tiles image-fastest inside a column band (conv_split.h, "image-major tile order"): 0 off, 1 = in the grouped
                                 // forward launch (ASPP) only - the default, 2 = in every split launch with taps.  Measured (profiles/r06_notes.md):
                                 // grouped forward 575.7 us with / 601.0 without at group_tps10 = 45; multi-source input gradient 709 with / 693
                                 // without (and 2.15 / 1.85 GB past L2): there it stays off; plain dilated launches: no difference
wsdl::Opt g_xcd_map{1};        // XCD-aware tile order of the split kernels: 0 off, 1 auto (by operand bytes), 10 + py forced
// row groups of the XCD-aware tile order: minimise (weight bytes x pixel groups + activation bytes x row groups); only
// worth a re-labelling when that beats the launch order (every XCD streams all weights, 1/8 of the pixels) by > 10 %
int choose_xcd_py(const ConvP& p, int gx, int gy) {
    if (!g_xcd_map || p.ksplit > 1 || ((long long)gx * gy) % 8 != 0) return 0;
    const double wb = (g_conv_arith ? 4.0 : 6.0) * (double)p.K * p.Cout, xb = 4.0 * (double)p.B * p.Cin * p.H * p.W;
    int best = 0;
    double best_cost = 0.0;
    for (int py = 1; py <= 8; py *= 2) {
        const int px = 8 / py;
        if (gy % py || gx % px) continue;
        if (g_xcd_map > 10 && py != g_xcd_map - 10) continue;     // 11 / 12 / 14 / 18: force py = 1 / 2 / 4 / 8
        const double cost = wb * px + xb * py;
        if (!best || cost < best_cost) { best = py; best_cost = cost; }
    }
    if (g_xcd_map == 1 && (best <= 1 || best_cost > 0.9 * (wb * 8 + xb))) return 0;
    return best;
}

// fp16x2 with K chunks of 32 runs on v_mfma_f32_16x16x32_f16 (MF): the 32x32x16 form of those small-tile configurations was an
// A/B option until round 4 ("conv_mfma16 = 0": neutral alone, 2.2 % slower on the step) and is no longer compiled.
template <int BM, int BN, int WM, int BK>
int launch_split(const ConvP& p_in, hipStream_t s, dim3 grid) {
    ConvP p = p_in;
    grid.z = p.ksplit > 1 ? p.ksplit : 1;
    p.xcd_py = choose_xcd_py(p, grid.x, grid.y);
    p.tile_img_major = g_tile_img_major == 2 && p.KH * p.KW > 1;
    constexpr bool MF = BK == 32;
    WSDL_TRACE("split<%d,%d,%d> ar=%d %s ks=%d xcd_py=%d nb=%d grid=%ux%ux%u", BM, BN, BK, (int)g_conv_arith,
               (MF && g_conv_arith) ? "mfma16x16x32" : "mfma32x32x16", p.ksplit, p.xcd_py, p.nb, grid.x, grid.y, grid.z);
    if (g_conv_arith == 2)
        hipLaunchKernelGGL((conv_igemm_split_kernel<BM, BN, WM, BK, kThreads, 2, MF>), grid, dim3(kThreads), 0, s, p);
    else if (g_conv_arith)
        hipLaunchKernelGGL((conv_igemm_split_kernel<BM, BN, WM, BK, kThreads, 1, MF>), grid, dim3(kThreads), 0, s, p);
    else
        hipLaunchKernelGGL((conv_igemm_split_kernel<BM, BN, WM, BK, kThreads, 0>), grid, dim3(kThreads), 0, s, p);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

wsdl::Opt g_conv_il{1};        // 256x128 form: MFMAs and staging instructions interleaved in every wave's stream (conv_split.h, IL)
wsdl::Opt g_tile64{1};         // 64 x 64 tiles for 128-row layers that give < 400 tiles of 128 x 64 (round 6): two four-wave workgroups per CU instead of
                               // one - l2.conv1 forward / l2.conv3 input gradient 20.2 -> 18.3 us, l2.conv2 38.8 -> 37.7 / 38.1 -> 36.6; ~20 us of the step
wsdl::Opt g_tile256{1};        // 256x128 tiles, 512 threads, one workgroup per CU where that still gives >= 256 workgroups: 4-5 % faster on layer4
                          // (128x256 measured 1 % behind it)
// (The 256x128 form with K chunks of 32 - "t256_bk32", 169-228 registers and 99-111 KB of LDS - was an option until round 4: faster
// alone (+3.5 % per kernel), 1-2.6 % slower on the STEP, where it leaves no room for the weight-gradient workgroups beside it
// (818.6 -> 828.2, 824 -> 845 img/s on two boxes with the 16-deep form, profiles/r02_notes.md).  Removed.)
// (256 x 256 tiles - two pixel tiles per workgroup sharing each weight chunk in LDS, 128 accumulators per lane, 202-215 registers,
// 82 KB of LDS - were an experiment of round 5 ("tile_n256"): multi-source input gradient 738 us against 724, every eligible
// launch of the step -1.9 % img/s.  Removed; profiles/r05_notes.md.)
int launch_split_256x128(const ConvP& p_in, hipStream_t s) {
    ConvP p = p_in;
    dim3 grid(p.grid_x > 0 ? p.grid_x : wsdl::cdiv(p.P, 128), wsdl::cdiv(p.Cout, 256), p.ksplit > 1 ? p.ksplit : 1);
    p.xcd_py = choose_xcd_py(p, grid.x, grid.y);
    p.xcd_rowfast = g_xcd_rowfast;
    p.tile_img_major = g_tile_img_major == 2 && (p.KH * p.KW > 1 || p.nsrc > 0);
    if (p.nsrc > 0 && g_ms_rowfast && grid.y % 2 == 0 && ((long long)grid.x * grid.y) % 8 == 0) {
        // the sources' weights together are the larger operand and every workgroup streams all of them: let the workgroups
        // an XCD runs at once span row tiles as well as pixel tiles
        // py row groups: an XCD owns grid.y / py row tiles x grid.x * py / 8 pixel tiles and runs ~32 of them at once, row tile
        // fastest.  ASPP's 8 x 128 tiles, same box: pixel-fastest order 792 us; row-fastest py = 1 / 2 / 4: 774 / 804 / 710 us
        int py = g_ms_py > 0 ? g_ms_py : 4;
        while (py > 1 && (grid.y % py || grid.x % (8 / py))) py /= 2;
        if (grid.y % py == 0 && grid.x % (8 / py) == 0) {
            p.xcd_py = py;
            p.xcd_rowfast = 1;
        }
    }
    WSDL_TRACE("split<256,128,16> ar=%d il=%d nsrc=%d ks=%d xcd_py=%d rowfast=%d imgmajor=%d nb=%d grid=%ux%ux%u", (int)g_conv_arith,
               (int)(g_conv_arith == 1 && g_conv_il), p.nsrc, p.ksplit, p.xcd_py, p.xcd_rowfast, p.tile_img_major, p.nb, grid.x, grid.y, grid.z);
    if (p.nsrc > 0) {        // several convolutions accumulated into one output (conv_split.h, MS)
        if (g_conv_arith == 2) hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 2, false, true>), grid, dim3(512), 0, s, p);
        else if (g_conv_arith && g_conv_il)
            hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 1, false, true, true>), grid, dim3(512), 0, s, p);
        else if (g_conv_arith) hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 1, false, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 0, false, true>), grid, dim3(512), 0, s, p);
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    if (g_conv_arith == 2) hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 2>), grid, dim3(512), 0, s, p);
    else if (g_conv_arith && g_conv_il)
        hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 1, false, false, true>), grid, dim3(512), 0, s, p);
    else if (g_conv_arith) hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 1>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv_igemm_split_kernel<256, 128, 4, 16, 512, 0>), grid, dim3(512), 0, s, p);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

template <int BM, int BN, int WM>
int launch_cfg(const ConvP& p, hipStream_t s, bool aligned, bool split) {
    dim3 grid(p.grid_x > 0 ? p.grid_x : wsdl::cdiv(p.P, BN), wsdl::cdiv(p.Cout, BM));
    constexpr bool kSmallTile = BM * BN <= 128 * 64;      // K chunks of 32 only where registers / LDS allow them
    if (split) {
        if constexpr (kSmallTile) {
            if (g_split_bk32 && p.Cin % 32 == 0) {
                return launch_split<BM, BN, WM, 32>(p, s, grid);
            }
        }
        return launch_split<BM, BN, WM, 16>(p, s, grid);
    } else if (aligned) {
        if constexpr (kSmallTile) {
            if (g_bk32 && p.Cin % 32 == 0) {
                return launch_fast<BM, BN, WM, 32>(p, s, grid);
            }
        }
        return launch_fast<BM, BN, WM, 16>(p, s, grid);
    } else {
        WSDL_TRACE("fp32_generic<%d,%d> grid=%ux%u", BM, BN, grid.x, grid.y);
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, false>), grid, dim3(kThreads), 0, s, p);
    }
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

// Tile choice: 128x128 (64x64 per wave) when that already gives every CU ~2 blocks; otherwise halve the
// pixel tile (128x64) so small-map layers (Cout 256 at 32x32: 256 -> 512 blocks) keep two waves per SIMD and
// the tap-skipping imbalance averages out.  Cout <= 64: 64x256 / 64x128.
// Split-K for grids that cannot fill the chip (small batches of small maps, e.g. the CAM path at B=8: 14x14 maps
// give 25 pixel tiles): returns the number of K slices (1 = no split).  Only the fast path supports it.
wsdl::Opt g_ksplit_big{1};   // 128x128 tiles + 2 K slices for grids of 200..399 such tiles with K >= 2048 (instead of 128x64
                        // tiles): aux 3x3 505 -> 435 us, layer3 3x3 139 -> 128 us; shorter K loses to the slab reduce
wsdl::Opt g_ksplit_target{512}, g_ksplit_max{8}, g_ksplit_min_chunks{4};   // small grids: workgroups aimed at, most K slices, fewest 32-deep chunks per slice
int igemm_ksplit(int P, int Cout, int Cin, int T, int dil) {
    if (Cin % 32 != 0 || Cout % 4 != 0 || Cout <= 64) return 1;
    const long long blocks = (long long)wsdl::cdiv(P, 64) * wsdl::cdiv(Cout, 128);       // 128x64 tile
    const int nq = T * (Cin / 32);
    if (g_ksplit_big) {
        const long long b128 = (long long)wsdl::cdiv(P, 128) * wsdl::cdiv(Cout, 128);
        if (b128 >= 200 && b128 < g_tile_threshold && nq >= 64) return 2;
    }
    if (blocks >= 160 || nq < 8) return 1;
    long long s = g_ksplit_target / blocks;
    if (s > nq / g_ksplit_min_chunks) s = nq / g_ksplit_min_chunks;
    if (s > g_ksplit_max) s = g_ksplit_max;
    return s < 2 ? 1 : (int)s;
}

wsdl::Opt g_col_bands{1};   // launch dilated convs per output-column band (see column_bands)

int launch_igemm(const ConvP& p_in, hipStream_t s, double flops, void* ws, size_t ws_bytes) {
    ConvP p = p_in;
    p.ksplit = 1;
    p.slab = nullptr;
    p.ow0 = 0;
    p.own = p.OW;
    p.nb = 1;
    p.grid_x = 0;
    p.xcd_py = 0;
    {
        const int ks = igemm_ksplit(p.P, p.Cout, p.Cin, p.KH * p.KW, p.bh < 0 ? -p.bh : p.bh);
        if (ks > 1 && ws && ws_bytes >= (size_t)ks * p.Cout * p.P * sizeof(float) && p.x_bytes != 0 &&
            (long long)p.K * p.Cout * 4 < (1ll << 31)) {
            p.ksplit = ks;
            p.slab = static_cast<float*>(ws);
        }
    }
    // fast path: K chunks inside one tap, 16-byte weight rows, 31-bit byte offsets
    const bool split = split_eligible(p.Cout, p.Cin, p.KH * p.KW);     // p.wt then holds the split layout
    if (split && p.x_bytes == 0) {
        wsdl::set_error("conv: activation extent >= 2 GiB is not supported by the split-bf16 kernels "
                        "(wsdl_set_option(\"conv_split\", 0) selects the fp32 kernels)");
        return WSDL_EINVAL;
    }
    if (split && g_conv_arith && !p.x_amax) {
        wsdl::set_error("conv: the fp16x2 split kernels need the activation tensor's amax (x_amax / dy_amax is null)");
        return WSDL_EINVAL;
    }
    // who publishes max|y|: the split kernel's epilogue (no split-K), the split-K reduce kernel (any arithmetic), else one
    // read pass over the output (fp32 kernels without split-K)
    float* amax_after = nullptr;
    if (p.y_amax && !(split && p.ksplit == 1) && p.ksplit == 1) {
        amax_after = p.y_amax;
        p.y_amax = nullptr;
    }
    const bool aligned = split || ((p.Cin % 16) == 0 && (p.Cout % 4) == 0 && p.x_bytes != 0 &&
                                   (long long)p.K * p.Cout * 4 < (1ll << 31));
    const long long kWant = g_tile_threshold;   // blocks below which the smaller tile is used (256 CUs x 2.5)
    int cfg;                           // 0: 128x128, 1: 128x64, 2: 64x256, 3: 64x128
    if (p.Cout <= 64)
        cfg = (long long)wsdl::cdiv(p.P, 256) * wsdl::cdiv(p.Cout, 64) >= kWant ? 2 : 3;
    else
        cfg = ((long long)wsdl::cdiv(p.P, 128) * wsdl::cdiv(p.Cout, 128) * p.ksplit >= kWant) ? 0 : 1;
    // column bands (fast path, no split-K): one launch per band, each with uniform column-tap validity
    Band bands[8];
    int nb = 1;
    bands[0] = Band{0, p.OW};
    if (aligned && g_col_bands) nb = column_bands(p.OW, p.W, p.ah, p.bh, p.ch, p.sh, p.KW, bands);
    // 256-row tiles only where the rows fill them (>= 90 %: not for 128-channel outputs)
    const bool t256 = cfg == 0 && split && g_tile256 && p.Cout * 10 >= wsdl::cdiv(p.Cout, 256) * 256 * 9 &&
                      (long long)wsdl::cdiv(p.P, 128) * wsdl::cdiv(p.Cout, 256) * p.ksplit >= 256;
    // 64 x 64 tiles ("tile64"): a 128-row layer at 32 x 32 x 16 is 256 tiles of 128 x 64 - ONE four-wave workgroup per CU,
    // whose chunks run staging, barrier, fragment reads and MFMAs one after the other (profiles/r05_notes.md); 512 tiles of 64 x 64 put
    // two workgroups on a CU, out of phase
    if (cfg == 1 && g_tile64 && split && g_conv_arith >= 1 && g_split_bk32 && p.Cin % 32 == 0 && p.Cout % 64 == 0 && p.ksplit == 1 &&
        p.nsrc == 0 && (long long)wsdl::cdiv(p.P, 64) * wsdl::cdiv(p.Cout, 128) < kWant)
        cfg = 4;
    const int bn_tile = cfg == 0 ? 128 : (cfg == 1 || cfg == 4) ? 64 : cfg == 2 ? 256 : 128;
    double executed = flops;
    if (aligned && wsdl::prof_enabled()) {
        executed = 0.0;
        for (int i = 0; i < nb; ++i) {
            ConvP q = p;
            q.ow0 = bands[i].ow0; q.own = bands[i].own; q.P = p.B * p.OH * q.own;
            executed += flops * ((double)q.own / p.OW) * igemm_executed_fraction(q, bn_tile);
        }
    }
    const double bytes = 4.0 * ((double)p.B * p.Cin * p.H * p.W + (double)p.K * p.Cout + (double)p.P * p.Cout * (p.res ? 2 : 1));
    wsdl::ProfScope prof(t256 ? (p.nsrc > 0 ? WSDL_PROF_SPLIT_MULTI : WSDL_PROF_SPLIT_256x128)
                              : split ? WSDL_PROF_SPLIT_128x128 + (cfg == 4 ? 1 : cfg) : WSDL_PROF_IGEMM_128x128_A + cfg * 2 + (aligned ? 0 : 1),
                         s, flops, executed, bytes);
    {
        ConvP q = p;
        q.nb = nb;
        int tiles = 0;
        for (int i = 0; i < nb && nb > 1; ++i) {
            q.b_ow0[i] = bands[i].ow0;
            q.b_own[i] = bands[i].own;
            q.b_tile0[i] = tiles;
            tiles += wsdl::cdiv((long long)p.B * p.OH * bands[i].own, bn_tile);
        }
        q.b_tile0[nb > 1 ? nb : 0] = tiles;
        q.grid_x = nb > 1 ? tiles : 0;
        int rc;
        if (p.nsrc > 0 && !t256) {
            wsdl::set_error("conv: a multi-source launch needs the 256x128 split form (rows a multiple of 256, >= 256 tiles)");
            return WSDL_EINVAL;
        }
        if (t256) {
            rc = launch_split_256x128(q, s);
        } else
        switch (cfg) {
            case 0: rc = launch_cfg<128, 128, 2>(q, s, aligned, split); break;
            case 1: rc = launch_cfg<128, 64, 2>(q, s, aligned, split); break;
            case 2: rc = launch_cfg<64, 256, 1>(q, s, aligned, split); break;
            case 4: rc = launch_split<64, 64, 2, 32>(q, s, dim3(q.grid_x > 0 ? q.grid_x : wsdl::cdiv(q.P, 64), wsdl::cdiv(q.Cout, 64))); break;
            default: rc = launch_cfg<64, 128, 1>(q, s, aligned, split); break;
        }
        if (rc) return rc;
    }
    if (p.ksplit > 1) {
        const long long total = (long long)p.Cout * p.P;
        const bool vec4 = (p.OH * p.OW) % 4 == 0 && p.y_bs % 4 == 0 && (!p.res || p.res_bs % 4 == 0) &&
                          (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && (!p.res || (reinterpret_cast<uintptr_t>(p.res) & 15) == 0);
        WSDL_TRACE("splitk_reduce%s", vec4 ? "_vec4" : "");
        if (vec4)
            hipLaunchKernelGGL(conv_splitk_reduce_vec4_kernel, dim3((int)std::min<long long>((total / 4 + 255) / 256, 8192)),
                               dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((int)std::min<long long>((total + 255) / 256, 4096)),
                               dim3(256), 0, s, p);
    }
    WSDL_LAUNCH_CHECK();
    if (amax_after) {
        // the kernel that ran has no amax epilogue (fp32 kernels, split-K reduce): one read pass over the output
        const long long per = (long long)p.Cout * p.OH * p.OW, total = (long long)p.B * per;
        hipLaunchKernelGGL(amax_kernel, dim3((int)std::min<long long>((total + 1023) / 1024, 2048)), dim3(256), 0, s, p.y, per,
                           p.y_bs, total, amax_after);
        WSDL_LAUNCH_CHECK();
    }
    return WSDL_OK;
}

// ---------------------------------------------------------------------------------------------
// The stem: 7x7, stride 2, padding 3, three input channels, 64 output channels (ResNet's conv1; K = 147 is no multiple of
// 16, so it ran on the generic fp32 kernel with per-element bounds checks: 150 us at B = 16, 256 x 256 - alone at the head
// of the step).  One workgroup per 8 x 32 tile of output pixels: the 21 x 69 x 3 input patch and all 64 x 147 weights go
// to LDS once (56 KB: two workgroups per CU), the implicit GEMM runs from LDS on v_mfma_f32_32x32x2_f32 - exact fp32
// products as before, k = tap * 3 + ci as in the k-major weight layout.  The patch is stored de-interleaved by column
// parity so that the 32 pixels of a fragment (input columns 2 ox + kx) read consecutive words.
constexpr int kStemTH = 8, kStemTW = 32;
constexpr int kStemIH = 2 * kStemTH + 5;            // 21 input rows
constexpr int kStemHalf = kStemTW + 3;              // 35 columns of one parity (69 = 35 even + 34 odd)
constexpr int kStemIW = 2 * kStemHalf;              // row stride in LDS
constexpr int kStemPlane = kStemIH * kStemIW;
constexpr int kStemK = 148;                         // 147 + one zero row: K steps of 2
wsdl::Opt g_stem_kernel{1};
wsdl::Opt g_stem_wgrad{1};       // the stem's weight gradient on its own kernel (stem_wgrad7x7s2_kernel)

__global__ __launch_bounds__(256, 2) void stem_conv7x7s2_kernel(ConvP p, int tiles_w, int tiles_h) {
    __shared__ float w_s[kStemK * 64];              // [k][cout]
    __shared__ float x_s[3 * kStemPlane];
    __shared__ int koff[kStemK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tw = blockIdx.x % tiles_w, th = (blockIdx.x / tiles_w) % tiles_h, b = blockIdx.x / (tiles_w * tiles_h);
    const int oy0 = th * kStemTH, ox0 = tw * kStemTW;
    for (int i = tid; i < 147 * 64; i += 256) w_s[i] = p.wt[i];
    if (tid < 64) w_s[147 * 64 + tid] = 0.f;
    if (tid < kStemK) {
        const int k = tid < 147 ? tid : 0;
        const int tap = k / 3, ci = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        koff[tid] = ci * kStemPlane + ky * kStemIW + (kx & 1) * kStemHalf + (kx >> 1);
    }
    const float* xb = p.x + (long long)b * p.x_bs;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3, HW = p.H * p.W;
    for (int i = tid; i < 3 * kStemIH * (2 * kStemTW + 5); i += 256) {
        const int ci = i / (kStemIH * 69), r = i - ci * (kStemIH * 69);
        const int row = r / 69, col = r - row * 69;
        const int iy = iy0 + row, ix = ix0 + col;
        float v = 0.f;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = xb[(long long)ci * HW + iy * p.W + ix];
        x_s[ci * kStemPlane + row * kStemIW + (col & 1) * kStemHalf + (col >> 1)] = v;
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int base0 = (2 * (2 * wid)) * kStemIW + l31, base1 = base0 + 2 * kStemIW;     // output rows 2 wid, 2 wid + 1
#pragma unroll 2
    for (int st = 0; st < kStemK / 2; ++st) {
        const int k = 2 * st + lh;
        const int ko = koff[k];
        const float b0 = x_s[base0 + ko], b1 = x_s[base1 + ko];
        const float a0 = w_s[k * 64 + l31], a1 = w_s[k * 64 + 32 + l31];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    // D row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = lane & 31 (the pixel): 128-byte runs of one channel
    const int OHOW = p.OH * p.OW;
    float vmax = 0.f;
    const int ox = ox0 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int oy = oy0 + 2 * wid + j;
        if (oy >= p.OH || ox >= p.OW) continue;
        float* yb = p.y + (long long)b * p.y_bs + oy * p.OW + ox;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r];
                if (p.scale) v *= p.scale[co];
                if (p.shift) v += p.shift[co];
                if (p.relu) v = fmaxf(v, 0.f);
                yb[(long long)co * OHOW] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
    }
    if (p.y_amax) publish_amax(vmax, p.y_amax);
}

static bool stem_eligible(const ConvP& p) {
    return g_stem_kernel && p.KH == 7 && p.KW == 7 && p.Cin == 3 && p.Cout == 64 && p.ah == 2 && p.bh == 1 && p.ch == -3 &&
           p.sh == 1 && !p.res && !p.accumulate && p.OH * 2 >= p.H && (long long)p.B * p.OH * p.OW < (1ll << 31);
}

static int launch_stem(const ConvP& p, hipStream_t s, double flops) {
    const int tiles_w = wsdl::cdiv(p.OW, kStemTW), tiles_h = wsdl::cdiv(p.OH, kStemTH);
    const long long blocks = (long long)p.B * tiles_w * tiles_h;
    WSDL_REQUIRE(blocks < (1ll << 31), "conv2d_fwd: too many stem tiles");
    const double bytes = 4.0 * ((double)p.B * 3 * p.H * p.W + 147.0 * 64 + (double)p.P * 64);
    wsdl::ProfScope prof(WSDL_PROF_STEM, s, flops, flops, bytes);
    WSDL_TRACE("stem7x7s2 fp32 workgroups=%lld", (long long)blocks);
    hipLaunchKernelGGL(stem_conv7x7s2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, tiles_w, tiles_h);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

// The buffer-descriptor kernels address the input tensor with 32-bit byte offsets.  A batch whose input extent
// reaches 2 GiB (large batches of large maps - 288 GB of HBM invite them) is processed in batch slices that each
// stay below it; a single image of >= 2 GiB falls through to launch_igemm's own handling.
int launch_igemm_sliced(const ConvP& p, int img_out_pixels, long long img_in_elems, hipStream_t s, double flops,
                        void* ws, size_t ws_bytes) {
    const long long limit = (1ll << 31) - 4;
    const long long whole = ((long long)(p.B - 1) * p.x_bs + img_in_elems) * 4;
    if (whole <= limit || p.B == 1 || img_in_elems * 4 > limit) {
        ConvP q = p;
        q.x_bytes = whole <= limit ? (unsigned)whole : 0u;      // 0 -> generic kernel / error in launch_igemm
        return launch_igemm(q, s, flops, ws, ws_bytes);
    }
    long long per = (limit / 4 - img_in_elems) / p.x_bs + 1;    // images per slice
    if (per < 1) per = 1;
    for (int b0 = 0; b0 < p.B; b0 += (int)per) {
        ConvP q = p;
        q.B = (int)std::min<long long>(per, p.B - b0);
        q.P = q.B * img_out_pixels;
        q.x = p.x + (long long)b0 * p.x_bs;
        q.y = p.y + (long long)b0 * p.y_bs;
        if (p.res) q.res = p.res + (long long)b0 * p.res_bs;
        q.x_bytes = (unsigned)(((long long)(q.B - 1) * p.x_bs + img_in_elems) * 4);
        if (int rc = launch_igemm(q, s, flops * q.B / p.B, ws, ws_bytes)) return rc;
    }
    return WSDL_OK;
}

// split count for wgrad: enough blocks to fill 256 CUs twice, at least 8 pixel chunks per split
wsdl::Opt g_wgrad_blocks{768};   // target number of workgroups of a weight-gradient launch (tiles x pixel splits)

// tile of the weight-gradient GEMM: the buffer-descriptor kernel needs Cout % BM == 0 and Cin % BN == 0
void wgrad_tile(int Cout, int Cin, int* BM, int* BN, bool* fast) {
    const int bm = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 0);
    const int bn = Cin % 128 == 0 ? 128 : (Cin % 64 == 0 ? 64 : 0);
    *fast = bm != 0 && bn != 0;
    *BM = *fast ? bm : (Cout <= 64 ? 64 : 128);
    *BN = *fast ? bn : 128;
}

wsdl::Opt g_wgrad_split{1};       // weight gradients on the bf16x3-split 32-pixel-chunk kernel where the shape allows (conv_split.h)
wsdl::Opt g_wgrad_force_s{0};     // experiments: fixed number of pixel splits
wsdl::Opt g_wgrad_dyraw{1};       // direct-fragment kernel: dY read as fp32 and split while staged (no dy_split16_kernel pass)
wsdl::Opt g_wgrad_chan_scale{0};  // fp16x2 weight-gradient kernels: one power-of-two scale per CHANNEL of x and of dY (a pre-pass takes the maxima)
wsdl::Opt g_wgrad_direct{1};      // x fragments of the split weight-gradient kernel straight from global memory (conv_wgrad_split16d_kernel)
// (fp16x2: the split weight-gradient kernels run on v_mfma_f32_16x16x32_f16; the 32x32x16 form - "wgrad_mfma16 = 0", 2 % slower on
// the step - was an A/B option until round 4.  conv_wgrad_split32_kernel remains as the bf16x3 path.)
wsdl::Opt g_wgrad_xcd{1};         // XCD-aware tile order of the split weight-gradient kernel (0 off, 1 contiguous, 2 blocked)
// dY is split once per launch, which pays off from about six 128-wide N tiles on (measured per shape: 1x1 convs with
// Cin <= 512 are faster on the fp32 kernel)
// (that was the bf16x3 kernel; the fp16x2 kernel on 16x16x32 MFMAs wins from ONE N tile on: 1x1 convs of layer2 / layer3.0
// 76 -> 61, 41 -> 31, 63 -> 43 us.  64-row / 64-column tiles for the 64-channel layers of layer1 lost to the fp32 kernels there -
// 87 -> 121 us on the 64 -> 64 3x3: few tiles, hundreds of slabs - and were removed in round 3.)
wsdl::Opt g_wgrad_min_tiles{1};     // (rounds 2-5: 6 - from one tile on was faster per kernel and 1 % slower per step, the dY pre-split, slab reduce and
                                    // amax passes joining the chain.  Round 6: the pre-split comes from the BatchNorm backward or is not needed (dY read
                                    // as fp32), the maxima are published by the producers: same box 882.5 / 890.0 / 886.8 img/s with 1 against 882.5 /
                                    // 882.8 / 878.7 with 6, and 857.9 / 857.2 against 847.9 / 851.5 on another box)
bool wgrad_chunk32(int Cout, int Cin, int N) {
    if (!g_wgrad_split || Cout % 128 != 0 || Cin % 128 != 0) return false;
    return N / 128 >= (g_conv_arith ? g_wgrad_min_tiles : std::max((int)g_wgrad_min_tiles, 6));
}

// taps that read at least one in-range input pixel for some output pixel (bit t of the result); the others (dilation
// >= map size: ASPP d36 on 32x32 maps keeps only the centre tap) contribute exact zeros
unsigned long long live_taps(int H, int W, int OH, int OW, int kh, int kw, int stride, int pad, int dil) {
    unsigned long long m = 0;
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
            const int dh = i * dil - pad, dw = j * dil - pad, t = i * kw + j;
            const bool dead = dh >= H || (OH - 1) * stride + dh < 0 || dw >= W || (OW - 1) * stride + dw < 0;
            if (!dead && t < 64) m |= 1ull << t;
        }
    return m;
}

// The share of a tap's pixel chunks (output rows) that read anything but padding, lightest live tap over heaviest: 1 for an
// undilated or small-dilation convolution, 20 / 32 for dilation 12 on a 32 x 32 map, 8 / 32 for dilation 24.
double wgrad_tap_balance(int H, int OH, int kh, int stride, int pad, int dil) {
    int lo = OH, hi = 0;
    for (int i = 0; i < kh; ++i) {
        int cnt = 0;
        for (int oh = 0; oh < OH; ++oh) {
            const int ih = oh * stride + i * dil - pad;
            cnt += ih >= 0 && ih < H;
        }
        if (cnt > 0) { lo = std::min(lo, cnt); hi = std::max(hi, cnt); }
    }
    return hi > 0 ? (double)lo / (double)hi : 1.0;
}

wsdl::Opt g_wgrad_imbalance_split{1};   // one more pixel split for tap-imbalanced launches (wgrad_splits)
// n_live: N counted over live taps only (the split kernel's dead-tap workgroups exit at once); tap_balance: wgrad_tap_balance
int wgrad_splits(int Cout, int Cin, int N, int P, int n_live, double tap_balance = 1.0) {
    int BM, BN;
    bool fast;
    wgrad_tile(Cout, Cin, &BM, &BN, &fast);
    long long tiles = (long long)wsdl::cdiv(Cout, BM) * wsdl::cdiv(N, BN);
    const int chunks = wsdl::cdiv(P, 32);
    const long long smax = std::max<long long>(1, std::min<long long>(256, chunks / 8));
    if (g_wgrad_force_s > 0) return (int)std::min<long long>(g_wgrad_force_s, smax);
    if (wgrad_chunk32(Cout, Cin, N)) {
        tiles = (long long)wsdl::cdiv(Cout, BM) * wsdl::cdiv(std::max(n_live, BN), BN);
        // two workgroups per CU (one 53 KB LDS image each): fill whole rounds of 512 slots
        const long long slots = 2 * kNumCU;
        long long best_s = 1;
        double best = -1.0;
        for (long long c = 1; c <= smax && tiles * c <= 5 * slots / 2; ++c) {
            const long long blocks = tiles * c, rounds = (blocks + slots - 1) / slots;
            double eff = (double)blocks / (double)(rounds * slots);
            if (blocks < slots) eff *= 0.9;      // a partly filled single round also loses the co-resident partner
            if (eff > best + 0.03) { best = eff; best_s = c; }
        }
        // Workgroups of a dilated convolution's outer taps skip the chunks that read only padding: with taps at 25-62 % of the
        // centre tap's work the round model above (equal workgroups) picks too few, too long workgroups - one more split lets
        // the dispatcher even them out (same box: ASPP d12 463 -> 442 us, d24 357 -> 287 at 32 x 32; d24 895 -> 819, d36 714 ->
        // 541 at 64 x 64; the balanced shapes lose 5-20 % with it)
        if (g_wgrad_imbalance_split && tap_balance < 0.7 && n_live > BN && best_s + 1 <= smax) ++best_s;
        return (int)best_s;
    }
    long long s0 = (g_wgrad_blocks + tiles - 1) / tiles;
    if (s0 > smax) s0 = smax;
    if (s0 < 1) s0 = 1;
    // among split counts near the target pick the one whose block count fills whole rounds of 256 CUs best
    // (720 blocks = 2.8 per CU run at the pace of the CUs holding 3)
    long long s = s0;
    double best = 0.0;
    for (long long c = std::max<long long>(1, s0 - 1); c <= std::min<long long>(smax, s0 + 3); ++c) {
        const long long blocks = tiles * c, rounds = (blocks + kNumCU - 1) / kNumCU;
        const double eff = (double)blocks / (double)(rounds * kNumCU);
        if (eff > best + 0.02) { best = eff; s = c; }
    }
    return (int)s;
}


// dw[co][ci] (+)= t[ci][co]: 32x32 tiles through LDS, both sides in 128-byte runs.  blockDim = (32, 8).
__global__ void transpose_add_kernel(const float* __restrict__ t, float* __restrict__ dw, int Cout, int Cin,
                                     int accumulate) {
    __shared__ float tile[32][33];
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + threadIdx.x;
        tile[r][threadIdx.x] = (ci < Cin && co < Cout) ? t[(long long)ci * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + threadIdx.x;
        if (co < Cout && ci < Cin) {
            const long long o = (long long)co * Cin + ci;
            dw[o] = accumulate ? dw[o] + tile[threadIdx.x][r] : tile[threadIdx.x][r];
        }
    }
}

// 1x1 / stride 1 convolutions whose OWN weight gradient has too few N tiles for the split kernel (Cin <= 640) but
// whose transpose has enough (Cout >= 768): the two operands are interchangeable there (no taps, no padding), so the
// library computes dW^T = wgrad(dy as input, x as output gradient) - pre-splitting the SMALLER tensor x, re-read by
// Cout/128 row tiles - and transposes the 1 M-element result.
bool wgrad_role_swap(int Cin, int Cout, int kh, int kw, int stride, int pad) {
    return kh == 1 && kw == 1 && stride == 1 && pad == 0 && !wgrad_chunk32(Cout, Cin, Cin) && wgrad_chunk32(Cin, Cout, Cout);
}

// one reduction as its own launch (the per-layer form)
int wgrad_reduce_one(const wsdl_wgrad_reduce_desc& d, hipStream_t s) {
    switch (d.kind) {
        case WSDL_WGRAD_REDUCE_TILED:
            hipLaunchKernelGGL(wgrad_reduce_tiled_kernel, dim3(d.grid_x, d.Cout), dim3(256), 0, s, d.slab, d.dw, d.S, d.Cout, d.Cin, d.T,
                               d.accumulate, d.live);
            break;
        case WSDL_WGRAD_REDUCE_VEC4:
            hipLaunchKernelGGL(wgrad_reduce_vec4_kernel, dim3(d.nblocks), dim3(256), 0, s, reinterpret_cast<const float4*>(d.slab),
                               reinterpret_cast<float4*>(d.dw), d.S, (long long)d.Cout * d.Cin * d.T / 4, d.accumulate);
            break;
        case WSDL_WGRAD_REDUCE_MANY:
            hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(d.nblocks), dim3(256), 0, s, d.slab, d.dw, d.S, d.Cout, d.Cin, d.T,
                               d.accumulate, d.live);
            break;
        case WSDL_WGRAD_REDUCE_MANY16:
            hipLaunchKernelGGL(wgrad_reduce_many16_kernel, dim3(d.nblocks), dim3(256), 0, s, d.slab, d.dw, d.S, d.Cout, d.Cin, d.T,
                               d.accumulate, d.live);
            break;
        default:
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(d.nblocks), dim3(256), 0, s, d.slab, d.dw, d.S, d.Cout, d.Cin, d.T, d.accumulate,
                               d.live);
            break;
    }
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

template <int BM, int BN, int WM>
int launch_wgrad_fast(const WgradP& p, hipStream_t s, int S) {
    dim3 grid(p.N / BN, p.Cout / BM, S);
    hipLaunchKernelGGL((conv_wgrad_fast_kernel<BM, BN, WM, 16>), grid, dim3(kThreads), 2 * (BM + BN) * 17 * sizeof(float), s, p);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // namespace

extern "C" {

int wsdl_set_option(const char* name, int value) {
    WSDL_REQUIRE(name, "set_option: null name");
    static std::mutex mu;                       // one writer at a time; readers see the old or the new value (wsdl::Opt)
    std::lock_guard<std::mutex> lock(mu);
    WSDL_REQUIRE(wsdl::g_plans_recording.load() == 0,
                 "set_option(%s): a launch plan is being recorded (its launches are chosen under ONE option set)", name);
    if (!strcmp(name, "tile_threshold")) { g_tile_threshold = value; return WSDL_OK; }
    if (!strcmp(name, "bk32")) { g_bk32 = value; return WSDL_OK; }
    if (!strcmp(name, "col_bands")) { g_col_bands = value; return WSDL_OK; }
    if (!strcmp(name, "conv_split")) { g_conv_split = value != 0; return WSDL_OK; }
    if (!strcmp(name, "stem_kernel")) { g_stem_kernel = value != 0; return WSDL_OK; }
    if (!strcmp(name, "stem_wgrad")) { g_stem_wgrad = value != 0; return WSDL_OK; }
    if (!strcmp(name, "wgrad_direct")) { g_wgrad_direct = value; return WSDL_OK; }
    if (!strcmp(name, "wgrad_imbalance_split")) { g_wgrad_imbalance_split = value != 0; return WSDL_OK; }
    if (!strcmp(name, "wgrad_chan_scale")) { g_wgrad_chan_scale = value != 0; return WSDL_OK; }
    if (!strcmp(name, "wgrad_dyraw")) { g_wgrad_dyraw = value; return WSDL_OK; }
    if (!strcmp(name, "split_bk32")) { g_split_bk32 = value != 0; return WSDL_OK; }
    if (!strcmp(name, "ksplit_big")) { g_ksplit_big = value; return WSDL_OK; }
    if (!strcmp(name, "tile256")) { g_tile256 = value; return WSDL_OK; }
    if (!strcmp(name, "tile64")) { g_tile64 = value; return WSDL_OK; }
    if (!strcmp(name, "conv_il")) { g_conv_il = value != 0; return WSDL_OK; }
    if (!strcmp(name, "tile_img_major")) { g_tile_img_major = value; return WSDL_OK; }
    if (!strcmp(name, "xcd_map")) { g_xcd_map = value; return WSDL_OK; }
    if (!strcmp(name, "ksplit_target")) { g_ksplit_target = value; return WSDL_OK; }
    if (!strcmp(name, "ksplit_max")) { g_ksplit_max = value; return WSDL_OK; }
    if (!strcmp(name, "ksplit_min_chunks")) { g_ksplit_min_chunks = value > 0 ? value : 1; return WSDL_OK; }
    if (!strcmp(name, "conv_arith")) {
        WSDL_REQUIRE(value >= 0 && value <= 2, "conv_arith: 0 (bf16x3), 1 (fp16x2), 2 (fp16x2, low piece at 2^11)");
        g_conv_arith = value;
        return WSDL_OK;
    }
    if (!strcmp(name, "range_sentinel")) { wsdl::g_range_sentinel = value != 0; return WSDL_OK; }
    if (!strcmp(name, "group_tps10")) { g_group_tps10 = value > 0 ? value : 45; return WSDL_OK; }
    if (!strcmp(name, "group_interleave")) { g_group_interleave = value; return WSDL_OK; }
    if (!strcmp(name, "ms_rowfast")) { g_ms_rowfast = value; return WSDL_OK; }
    if (!strcmp(name, "xcd_rowfast")) { g_xcd_rowfast = value; return WSDL_OK; }
    if (!strcmp(name, "ms_py")) { g_ms_py = value; return WSDL_OK; }
    if (!strcmp(name, "bn_coop")) { wsdl::g_bn_coop = value; return WSDL_OK; }
    if (!strcmp(name, "bn_coop_wide")) { wsdl::g_bn_coop_wide = value; return WSDL_OK; }
    if (!strcmp(name, "bn_resident")) { wsdl::g_bn_resident = value; return WSDL_OK; }
    if (!strcmp(name, "bn_wide_c")) { wsdl::g_bn_wide_c = value; return WSDL_OK; }
    if (!strcmp(name, "layercam_tail_mod")) {
        WSDL_REQUIRE(value >= 0 && value <= 32, "layercam_tail_mod: 0..32");
        wsdl::g_layercam_tail_mod = value;
        return WSDL_OK;
    }
    if (!strcmp(name, "wgrad_force_s")) { g_wgrad_force_s = value; return WSDL_OK; }
    if (!strcmp(name, "wgrad_split")) { g_wgrad_split = value != 0; return WSDL_OK; }
    if (!strcmp(name, "wgrad_xcd")) { g_wgrad_xcd = value; return WSDL_OK; }
    if (!strcmp(name, "wgrad_min_tiles")) { g_wgrad_min_tiles = value > 0 ? value : 1; return WSDL_OK; }
    if (!strcmp(name, "wgrad_blocks")) { g_wgrad_blocks = value > 0 ? value : 768; return WSDL_OK; }
    if (!strcmp(name, "wgrad_bk")) { g_wgrad_bk = value == 32 ? 32 : 16; return WSDL_OK; }
    wsdl::set_error("set_option: unknown option %s", name);
    return WSDL_EINVAL;
}

size_t wsdl_conv2d_weight_layout_bytes(int Cout, int Cin, int kh, int kw, int dgrad, int* is_plain) {
    const int T = kh * kw;
    const bool split = dgrad ? split_eligible(Cin, Cout, T) : split_eligible(Cout, Cin, T);
    if (is_plain) *is_plain = split ? 0 : 1;
    const size_t n = (size_t)Cout * Cin * T;
    return split ? (size_t)split_layout_bytes(g_conv_arith, (long long)T * (dgrad ? Cout : Cin), dgrad ? Cin : Cout) : n * 4;
}

int wsdl_conv2d_prep_weights(const float* w, void* wt_fwd, void* wt_dgrad, int Cout, int Cin,
                             int kh, int kw, const float* w_amax, wsdl_stream_t stream) {
    WSDL_REQUIRE(w && (wt_fwd || wt_dgrad), "prep_weights: null pointer");
    WSDL_REQUIRE(Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "prep_weights: bad shape");
    const long long total = (long long)Cout * Cin * kh * kw;
    const int T = kh * kw;
    const bool fwd_split = wt_fwd && split_eligible(Cout, Cin, T);
    const bool dg_split = wt_dgrad && split_eligible(Cin, Cout, T);
    float* pf = fwd_split ? nullptr : static_cast<float*>(wt_fwd);
    float* pd = dg_split ? nullptr : static_cast<float*>(wt_dgrad);
    if (fwd_split || dg_split) {
        dim3 grid(wsdl::cdiv(Cin, 32), wsdl::cdiv(Cout, 32));
        unsigned char* f8 = fwd_split ? static_cast<unsigned char*>(wt_fwd) : nullptr;
        unsigned char* d8 = dg_split ? static_cast<unsigned char*>(wt_dgrad) : nullptr;
        if (g_conv_arith) {
            // the tensor's max|w| is reduced into the trailer of one layout; the layout kernel scales by it and copies it
            // into the other layout's trailer
            const float* amax = w_amax;
            if (!amax) {
                const long long body = split_layout_bytes(1, (long long)T * Cin, Cout) - 16;
                float* tr = reinterpret_cast<float*>((f8 ? f8 : d8) + body);
                WSDL_HIP_CHECK(hipMemsetAsync(tr, 0, 16, wsdl::as_stream(stream)));
                hipLaunchKernelGGL(amax_kernel, dim3((int)std::min<long long>((total + 1023) / 1024, 1024)), dim3(256), 0,
                                   wsdl::as_stream(stream), w, total, total, total, tr);
                amax = tr;
            }
            if (g_conv_arith == 2)
                hipLaunchKernelGGL(prep_weights_split_kernel<2>, grid, dim3(256), 0, wsdl::as_stream(stream), w, f8, d8, Cout, Cin,
                                   T, amax);
            else
                hipLaunchKernelGGL(prep_weights_split_kernel<1>, grid, dim3(256), 0, wsdl::as_stream(stream), w, f8, d8, Cout, Cin,
                                   T, amax);
        } else {
            hipLaunchKernelGGL(prep_weights_split_kernel<0>, grid, dim3(256), 0, wsdl::as_stream(stream), w, f8, d8, Cout, Cin,
                               T, static_cast<const float*>(nullptr));
        }
        WSDL_LAUNCH_CHECK();
    }
    if (pf || pd) {
        if (T <= 9 && Cout <= 65535 * 32) {
            dim3 grid(wsdl::cdiv(Cin, 32), wsdl::cdiv(Cout, 32));
            if (T == 1)
                hipLaunchKernelGGL((prep_weights_tiled_kernel<1>), grid, dim3(32, 8), 0, wsdl::as_stream(stream), w, pf,
                                   pd, Cout, Cin, T);
            else
                hipLaunchKernelGGL((prep_weights_tiled_kernel<9>), grid, dim3(32, 8), 0, wsdl::as_stream(stream), w, pf,
                                   pd, Cout, Cin, T);
        } else {
            const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
            hipLaunchKernelGGL(prep_weights_kernel, dim3(blocks), dim3(256), 0, wsdl::as_stream(stream), w, pf, pd,
                               Cout, Cin, T);
        }
    }
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_conv2d_fwd(const float* x, const void* wt_fwd, float* y, int B, int Cin, int H, int W,
                    int Cout, int kh, int kw, int stride, int pad, int dil, const float* scale,
                    const float* shift, const float* residual, int relu, long long x_bs,
                    long long y_bs, long long res_bs, const float* x_amax, float* y_amax, void* ws, size_t ws_bytes,
                    wsdl_stream_t stream) {
    WSDL_REQUIRE(x && wt_fwd && y, "conv2d_fwd: null pointer");
    int OH, OW;
    if (int rc = check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return rc;
    ConvP p{};
    p.x = x; p.wt = static_cast<const float*>(wt_fwd); p.y = y; p.scale = scale; p.shift = shift; p.res = residual;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.KH = kh; p.KW = kw;
    p.ah = stride; p.bh = dil; p.ch = -pad; p.sh = 1;
    p.K = kh * kw * Cin;
    p.x_bs = x_bs ? x_bs : (long long)Cin * H * W;
    p.y_bs = y_bs ? y_bs : (long long)Cout * OH * OW;
    p.res_bs = res_bs ? res_bs : (long long)Cout * OH * OW;
    WSDL_REQUIRE(p.x_bs >= (long long)Cin * H * W && p.y_bs >= (long long)Cout * OH * OW, "conv2d_fwd: batch stride smaller than an image");
    p.relu = relu; p.accumulate = 0; p.P = B * OH * OW;
    p.x_amax = x_amax; p.y_amax = y_amax;
    if (stem_eligible(p)) return launch_stem(p, wsdl::as_stream(stream), 2.0 * p.P * (double)Cout * p.K);
    return launch_igemm_sliced(p, OH * OW, (long long)Cin * H * W, wsdl::as_stream(stream),
                               2.0 * p.P * (double)Cout * p.K, ws, ws_bytes);
}

int wsdl_conv2d_dgrad(const float* dy, const void* wt_dgrad, float* dx, int B, int Cin, int H, int W,
                      int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, const uint8_t* acc_mask,
                      long long dy_bs, const float* dy_amax, void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && wt_dgrad && dx, "conv2d_dgrad: null pointer");
    WSDL_REQUIRE(!acc_mask || accumulate, "conv2d_dgrad: acc_mask masks the accumulated value (accumulate = 1)");
    int OH, OW;
    if (int rc = check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return rc;
    ConvP p{};
    p.x = dy; p.wt = static_cast<const float*>(wt_dgrad); p.y = dx;
    // roles swap: "input" is dY (Cout channels, OH x OW), "output" is dX (Cin channels, H x W)
    p.B = B; p.Cin = Cout; p.H = OH; p.W = OW; p.Cout = Cin; p.OH = H; p.OW = W; p.KH = kh; p.KW = kw;
    p.ah = 1; p.bh = -dil; p.ch = pad; p.sh = stride;
    p.K = kh * kw * Cout;
    p.x_bs = dy_bs ? dy_bs : (long long)Cout * OH * OW;
    p.y_bs = (long long)Cin * H * W;
    p.res_bs = p.y_bs;
    WSDL_REQUIRE(p.x_bs >= (long long)Cout * OH * OW, "conv2d_dgrad: batch stride smaller than an image");
    p.relu = 0; p.accumulate = accumulate; p.P = B * H * W;
    p.acc_mask = acc_mask; p.acc_base = dx;
    p.x_amax = dy_amax; p.y_amax = nullptr;
    return launch_igemm_sliced(p, H * W, (long long)Cout * OH * OW, wsdl::as_stream(stream),
                               2.0 * (double)B * OH * OW * (double)Cout * kh * kw * Cin, ws, ws_bytes);
}

// ---- several forward convolutions of ONE input in one launch (conv_split.h: conv_igemm_split_group_kernel) ----------------
// K slices per problem: a tile executes 1 .. T taps (padding taps are skipped); slices of about two taps each keep the
// longest workgroup of the launch short against the launch itself (the dispatcher then balances the CUs)
static int group_ksplit(int max_taps) { return std::max(1, std::min(4, (max_taps * 10) / g_group_tps10)); }

static int group_max_taps(int H, int W, int k, int dil) {
    if (k == 1) return 1;
    int rows = 1, cols = 1;                      // centre tap + the shifted ones that can reach a real pixel ('same' padding)
    if (dil < H) rows += 2;
    if (dil < W) cols += 2;
    return rows * cols;
}

int wsdl_conv2d_fwd_group_ok(int n, int B, int Cin, int H, int W, int Cout) {
    if (n < 2 || n > 4 || !g_conv_split || !g_tile256) return 0;
    if (Cout % 256 != 0 || Cin % 16 != 0 || (H * W) % 4 != 0) return 0;
    if (((long long)(B - 1) * Cin * H * W + (long long)Cin * H * W) * 4 > (1ll << 31) - 4) return 0;
    if (wsdl::cdiv((long long)B * H * W, 128) % 8 != 0) return 0;        // every problem starts on a multiple of 8 workgroups
    return 1;
}

size_t wsdl_conv2d_fwd_group_workspace(int n, const int* k, const int* dil, int B, int Cin, int H, int W, int Cout) {
    if (!k || !dil || !wsdl_conv2d_fwd_group_ok(n, B, Cin, H, W, Cout)) return 0;
    size_t total = 256;
    for (int i = 0; i < n; ++i) {
        const int ks = group_ksplit(group_max_taps(H, W, k[i], dil[i]));
        if (ks > 1) total += wsdl::align_up((size_t)ks * Cout * B * H * W * sizeof(float), 256);
    }
    return total;
}

int wsdl_conv2d_fwd_group(int n, const float* x, const void* const* wt_fwd, float* const* y, const int* k, const int* dil,
                          int B, int Cin, int H, int W, int Cout, long long x_bs, const long long* y_bs,
                          const float* x_amax, void* ws, size_t ws_bytes, wsdl_stream_t stream) {
    WSDL_REQUIRE(x && wt_fwd && y && k && dil, "conv2d_fwd_group: null pointer");
    WSDL_REQUIRE(wsdl_conv2d_fwd_group_ok(n, B, Cin, H, W, Cout), "conv2d_fwd_group: geometry not supported (see "
                 "wsdl_conv2d_fwd_group_ok): %d problems, B %d, %d -> %d channels, %d x %d", n, B, Cin, Cout, H, W);
    WSDL_REQUIRE(!g_conv_arith || x_amax, "conv2d_fwd_group: the fp16x2 kernels need x_amax");
    WSDL_REQUIRE(ws_bytes >= wsdl_conv2d_fwd_group_workspace(n, k, dil, B, Cin, H, W, Cout) && (ws || ws_bytes == 0),
                 "conv2d_fwd_group: workspace too small");
    hipStream_t s = wsdl::as_stream(stream);
    ConvGroup grp{};
    grp.n = n;
    // heaviest problem first: most taps per tile first (the dispatcher starts workgroups in index order)
    int order[4] = {0, 1, 2, 3};
    std::stable_sort(order, order + n, [&](int a, int b) {
        return group_max_taps(H, W, k[a], dil[a]) > group_max_taps(H, W, k[b], dil[b]); });
    unsigned char* wsp = static_cast<unsigned char*>(ws);
    int start = 0;
    double flops = 0.0, executed = 0.0, bytes = 4.0 * (double)B * Cin * H * W;
    const long long xbs = x_bs ? x_bs : (long long)Cin * H * W;
    for (int j = 0; j < n; ++j) {
        const int i = order[j];
        WSDL_REQUIRE(wt_fwd[i] && y[i], "conv2d_fwd_group: null problem %d", i);
        WSDL_REQUIRE((k[i] == 1 || k[i] == 3) && dil[i] >= 1, "conv2d_fwd_group: problem %d: 1x1 or 3x3, stride 1", i);
        WSDL_REQUIRE(split_eligible(Cout, Cin, k[i] * k[i]), "conv2d_fwd_group: problem %d is not on the split kernels", i);
        ConvP& p = grp.p[j];
        const int pad = dil[i] * (k[i] - 1) / 2;
        p.x = x; p.wt = static_cast<const float*>(wt_fwd[i]); p.y = y[i];
        p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = H; p.OW = W; p.KH = p.KW = k[i];
        p.ah = 1; p.bh = dil[i]; p.ch = -pad; p.sh = 1;
        p.K = k[i] * k[i] * Cin;
        p.x_bs = xbs;
        p.y_bs = (y_bs && y_bs[i]) ? y_bs[i] : (long long)Cout * H * W;
        p.res_bs = p.y_bs;
        WSDL_REQUIRE(p.x_bs >= (long long)Cin * H * W && p.y_bs >= (long long)Cout * H * W, "conv2d_fwd_group: batch stride smaller than an image");
        p.P = B * H * W;
        p.x_bytes = (unsigned)(((long long)(B - 1) * p.x_bs + (long long)Cin * H * W) * 4);
        p.x_amax = x_amax;
        p.ow0 = 0; p.own = W; p.nb = 1;
        Band bands[8];
        int nb = 1;
        bands[0] = Band{0, W};
        if (g_col_bands) nb = column_bands(W, W, p.ah, p.bh, p.ch, p.sh, p.KW, bands);
        int tiles = 0;
        for (int b = 0; b < nb && nb > 1; ++b) {
            p.b_ow0[b] = bands[b].ow0;
            p.b_own[b] = bands[b].own;
            p.b_tile0[b] = tiles;
            tiles += wsdl::cdiv((long long)B * H * bands[b].own, 128);
        }
        p.nb = nb;
        p.b_tile0[nb > 1 ? nb : 0] = tiles;
        const int gx = nb > 1 ? tiles : wsdl::cdiv(p.P, 128), gy = Cout / 256;
        const int ks = group_ksplit(group_max_taps(H, W, k[i], dil[i]));
        p.ksplit = ks;
        if (ks > 1) {
            p.slab = reinterpret_cast<float*>(wsp);
            wsp += wsdl::align_up((size_t)ks * Cout * p.P * sizeof(float), 256);
        }
        p.xcd_py = (start % 8 == 0) ? choose_xcd_py(p, gx, gy) : 0;
        p.tile_img_major = g_tile_img_major >= 1 && k[i] > 1;
        grp.start[j] = start;
        grp.gx[j] = gx;
        grp.gy[j] = gy;
        start += gx * gy * ks;
        const double f = 2.0 * p.P * (double)Cout * p.K;
        flops += f;
        if (wsdl::prof_enabled()) {
            for (int b = 0; b < nb; ++b) {
                ConvP q = p;
                q.ow0 = bands[b].ow0; q.own = bands[b].own; q.P = B * H * q.own;
                executed += f * ((double)q.own / W) * igemm_executed_fraction(q, 128);
            }
        }
        bytes += 4.0 * ((double)p.K * Cout + (double)p.P * Cout);
    }
    grp.start[n] = start;
    // stream-interleaved order: one (problem, K slice) stream per residue of the workgroup index - when the streams number a
    // multiple or a divisor of the 8 XCDs and every problem has the same tile count, each XCD runs whole streams
    {
        int ns = 0;
        bool same = true;
        for (int j = 0; j < n; ++j) {
            ns += grp.p[j].ksplit;
            same = same && grp.gx[j] * grp.gy[j] == grp.gx[0] * grp.gy[0];
        }
        if (g_group_interleave && same && ns <= 8 && (8 % ns == 0)) {
            grp.ns = ns;
            int s = 0;
            for (int j = 0; j < n; ++j)
                for (int z = 0; z < grp.p[j].ksplit; ++z) {
                    grp.s_prob[s] = j;
                    grp.s_slice[s] = z;
                    ++s;
                }
            for (int j = 0; j < n; ++j) grp.p[j].xcd_py = 0;      // the streams ARE the XCD assignment
        }
    }
    {
        wsdl::ProfScope prof(WSDL_PROF_SPLIT_GROUP, s, flops, wsdl::prof_enabled() ? executed : flops, bytes);
        WSDL_TRACE("group split<256,128,16> ar=%d il=%d problems=%d streams=%d ks=%d/%d/%d/%d workgroups=%d", (int)g_conv_arith,
                   (int)(g_conv_arith == 1 && g_conv_il), n, grp.ns, grp.p[0].ksplit, n > 1 ? grp.p[1].ksplit : 0,
                   n > 2 ? grp.p[2].ksplit : 0, n > 3 ? grp.p[3].ksplit : 0, start);
        if (g_conv_arith == 2)
            hipLaunchKernelGGL((conv_igemm_split_group_kernel<256, 128, 4, 16, 512, 2>), dim3(start), dim3(512), 0, s, grp);
        else if (g_conv_arith && g_conv_il)
            hipLaunchKernelGGL((conv_igemm_split_group_kernel<256, 128, 4, 16, 512, 1, false, true>), dim3(start), dim3(512), 0, s, grp);
        else if (g_conv_arith)
            hipLaunchKernelGGL((conv_igemm_split_group_kernel<256, 128, 4, 16, 512, 1>), dim3(start), dim3(512), 0, s, grp);
        else
            hipLaunchKernelGGL((conv_igemm_split_group_kernel<256, 128, 4, 16, 512, 0>), dim3(start), dim3(512), 0, s, grp);
        WSDL_LAUNCH_CHECK();
    }
    for (int j = 0; j < n; ++j) {
        const ConvP& p = grp.p[j];
        if (p.ksplit <= 1) continue;
        const long long total = (long long)p.Cout * p.P;
        const bool vec4 = p.y_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0;
        if (vec4)
            hipLaunchKernelGGL(conv_splitk_reduce_vec4_kernel, dim3((int)std::min<long long>((total / 4 + 255) / 256, 8192)),
                               dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((int)std::min<long long>((total + 255) / 256, 4096)),
                               dim3(256), 0, s, p);
        WSDL_LAUNCH_CHECK();
    }
    return WSDL_OK;
}

// Is the one-launch input gradient of n convolutions over the same input available for this geometry (what
// wsdl_conv2d_dgrad_multi requires beyond its arguments being well-formed)?
int wsdl_conv2d_dgrad_multi_ok(int n, int B, int Cin, int H, int W, int Cout) {
    if (n < 2 || n > 4 || !g_conv_split || !g_tile256) return 0;
    if (Cin % 256 != 0 || Cout % 16 != 0) return 0;
    if ((long long)wsdl::cdiv((long long)B * H * W, 128) * (Cin / 256) < 256) return 0;                  // the 256 x 128 form's own rule
    if ((long long)wsdl::cdiv((long long)B * H * W, 128) * wsdl::cdiv(Cin, 128) < g_tile_threshold) return 0;
    if (igemm_ksplit(B * H * W, Cin, Cout, 9, 1) != 1) return 0;
    if (((long long)(B - 1) * Cout * H * W + (long long)Cout * H * W) * 4 > (1ll << 31) - 4) return 0;      // one batch slice
    return 1;
}

int wsdl_conv2d_dgrad_multi(int n, const float* const* dy, const void* const* wt_dgrad, const float* const* dy_amax,
                            const int* k, const int* dil, const long long* dy_bs, float* dx, int B, int Cin, int H, int W,
                            int Cout, int accumulate, wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && wt_dgrad && dy_amax && k && dil && dx, "conv2d_dgrad_multi: null pointer");
    WSDL_REQUIRE(wsdl_conv2d_dgrad_multi_ok(n, B, Cin, H, W, Cout), "conv2d_dgrad_multi: geometry not supported (see "
                 "wsdl_conv2d_dgrad_multi_ok): %d sources, B %d, %d -> %d channels, %d x %d", n, B, Cin, Cout, H, W);
    ConvP p{};
    double flops = 0.0;
    for (int i = 0; i < n; ++i) {
        WSDL_REQUIRE(dy[i] && wt_dgrad[i], "conv2d_dgrad_multi: null source %d", i);
        WSDL_REQUIRE((k[i] == 1 || k[i] == 3) && dil[i] >= 1, "conv2d_dgrad_multi: source %d: 1x1 or 3x3, stride 1", i);
        WSDL_REQUIRE(split_eligible(Cin, Cout, k[i] * k[i]), "conv2d_dgrad_multi: source %d is not on the split kernels", i);
        WSDL_REQUIRE(!g_conv_arith || dy_amax[i], "conv2d_dgrad_multi: source %d: the fp16x2 kernels need dy_amax", i);
        const int pad = dil[i] * (k[i] - 1) / 2;                  // 'same' padding: every source's dY has the input's H x W
        const long long bs = (dy_bs && dy_bs[i]) ? dy_bs[i] : (long long)Cout * H * W;
        WSDL_REQUIRE(bs >= (long long)Cout * H * W, "conv2d_dgrad_multi: batch stride smaller than an image");
        const unsigned xbytes = (unsigned)(((long long)(B - 1) * bs + (long long)Cout * H * W) * 4);
        WSDL_REQUIRE(((long long)(B - 1) * bs + (long long)Cout * H * W) * 4 <= (1ll << 31) - 4, "conv2d_dgrad_multi: dY extent >= 2 GiB");
        flops += 2.0 * (double)B * H * W * (double)Cout * k[i] * k[i] * Cin;
        if (i == 0) {
            p.x = dy[0]; p.wt = static_cast<const float*>(wt_dgrad[0]); p.x_amax = dy_amax[0];
            p.KH = p.KW = k[0]; p.bh = -dil[0]; p.ch = pad; p.K = k[0] * k[0] * Cout; p.x_bs = bs; p.x_bytes = xbytes;
        } else {
            ConvSrc& c = p.src[i - 1];
            c.x = dy[i]; c.wt = static_cast<const float*>(wt_dgrad[i]); c.x_amax = dy_amax[i];
            c.KH = c.KW = k[i]; c.bh = -dil[i]; c.ch = pad; c.K = k[i] * k[i] * Cout; c.Cin = Cout; c.x_bs = bs; c.x_bytes = xbytes;
        }
    }
    p.nsrc = n - 1;
    p.y = dx;
    p.B = B; p.Cin = Cout; p.H = H; p.W = W; p.Cout = Cin; p.OH = H; p.OW = W;
    p.ah = 1; p.sh = 1;
    p.y_bs = (long long)Cin * H * W;
    p.res_bs = p.y_bs;
    p.relu = 0; p.accumulate = accumulate; p.P = B * H * W;
    p.acc_mask = nullptr; p.acc_base = dx;
    p.y_amax = nullptr;
    return launch_igemm(p, wsdl::as_stream(stream), flops, nullptr, 0);
}

int wsdl_conv2d_prep_weights_multi(const wsdl_prep_desc* desc, int n, int total_blocks, wsdl_stream_t stream) {
    WSDL_REQUIRE(desc && n > 0 && total_blocks > 0, "prep_weights_multi: bad arguments");
    WSDL_REQUIRE(g_conv_arith >= 1 && g_conv_split, "prep_weights_multi: only the fp16x2 split layouts");
    if (g_conv_arith == 2)
        hipLaunchKernelGGL(prep_weights_split_multi_kernel<2>, dim3(total_blocks), dim3(256), 0, wsdl::as_stream(stream), desc, n);
    else
        hipLaunchKernelGGL(prep_weights_split_multi_kernel<1>, dim3(total_blocks), dim3(256), 0, wsdl::as_stream(stream), desc, n);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

size_t wsdl_conv2d_igemm_workspace(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                                   int pad, int dil, int dgrad) {
    int OH, OW;
    if (check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return 0;
    // forward: P = B*OH*OW output pixels, Cout rows; dgrad: roles swapped
    const int P = dgrad ? B * H * W : B * OH * OW, M = dgrad ? Cin : Cout, Kc = dgrad ? Cout : Cin;
    const int ks = igemm_ksplit(P, M, Kc, kh * kw, dil);
    return ks > 1 ? (size_t)ks * M * P * sizeof(float) : 0;
}

// bytes of the pre-split dY image of the 32-pixel-chunk weight-gradient kernel (0: kernel not used)
// max |t| of every channel of t[B][C][HW] (batch stride bs) -> out[C] (zeroed by the caller); workgroup (c, image group)
__global__ void channel_amax_kernel(const float* __restrict__ t, int B, int C, int HW, long long bs, float* __restrict__ out) {
    const int c = blockIdx.x;
    float m = 0.f;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* src = t + (long long)b * bs + (long long)c * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) m = fmaxf(m, fabsf(src[i]));
    }
    publish_amax(m, out + c);
}

// ---------------------------------------------------------------------------------------------
// The stem's WEIGHT GRADIENT (round 6): dW[64][3][7][7] = sum over pixels of dY[co][pixel] x patch[(ky, kx, ci)][pixel].  K = 147
// is no multiple of 16 and Cin = 3, so it ran on the generic fp32 kernel with per-element bounds checks: 164 us at B = 16,
// 256 x 256 - the LAST kernel of the backward pass, on the main stream, with nothing left to overlap it (the fp32-MFMA bound of
// its 5 GFLOP is 34 us).  Same idea as the forward stem kernel: a workgroup takes 4 x 32 output pixels, stages the
// 13 x 69 x 3 input patch (de-interleaved by column parity) and the 64 x 128 tile of dY in LDS once, and runs the GEMM
// D[co][k] over the tile's 128 pixels on v_mfma_f32_32x32x2_f32 - exact fp32 products as before.  The 2 x 5 output tiles
// (64 channels x 147 -> 160 k) are dealt to the four waves 3 / 3 / 2 / 2; a workgroup walks several pixel tiles with its
// accumulators in registers (grid = at most 512 workgroups = 512 slabs of 37 KB, summed in fixed order by the slab reduction).
constexpr int kSwTH = 4;                              // output rows of a tile
constexpr int kSwIH = 2 * kSwTH + 5;                  // 13 input rows
constexpr int kSwPlane = kSwIH * kStemIW;
constexpr int kSwLD = kSwTH * kStemTW + 1;            // dY rows of 128 pixels + 1: lanes of different channels on different banks
constexpr int kSwMaxGrid = 512;                     // two workgroups per CU (178 registers): one round, four tiles each at B = 16, 256 x 256

__global__ __launch_bounds__(256, 2) void stem_wgrad7x7s2_kernel(WgradP p, int tiles_w, int tiles_h, int total_tiles) {
    __shared__ float dy_s[64 * kSwLD];
    __shared__ float x_s[3 * kSwPlane + 1];           // + one zero word: what the 13 padding columns of the k dimension read
    __shared__ int koff[160];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    if (tid < 160) {
        int off = 3 * kSwPlane;                       // k >= 147: the zero word (and no pixel offset: see the loop)
        if (tid < 147) {
            const int tap = tid / 3, ci = tid - tap * 3, ky = tap / 7, kx = tap - ky * 7;
            off = ci * kSwPlane + ky * kStemIW + (kx & 1) * kStemHalf + (kx >> 1);
        }
        koff[tid] = off;
    }
    if (tid == 0) x_s[3 * kSwPlane] = 0.f;
    // this wave's output tiles: (channel tile 0, k tile nt0), (channel tile 1, k tile nt0) and, for waves 0 / 1, (channel
    // tile wid, k tile 4)
    const int nt0 = wid;
    const bool third = wid < 2;
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.f;
    const int HW = p.H * p.W, OHOW = p.OH * p.OW;
    __syncthreads();
    const int kb0 = koff[nt0 * 32 + l31], kb4 = koff[128 + l31];
    const bool pad0 = nt0 * 32 + l31 >= 147, pad4 = 128 + l31 >= 147;
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, b = t / (tiles_w * tiles_h);
        const int oy0 = th * kSwTH, ox0 = tw * kStemTW;
        // the input patch
        const float* xb = p.x + (long long)b * p.x_bs;
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        // (all of a thread's loads are issued before its first LDS store: a load-store-load-store loop waits out the memory
        // latency once per element - 43 times per tile)
        constexpr int kXN = (3 * kSwIH * 69 + 255) / 256;          // 11 patch elements per thread
        float xv[kXN];
#pragma unroll
        for (int e = 0; e < kXN; ++e) {
            const int i = tid + e * 256;
            const int ci = i / (kSwIH * 69), r = i - ci * (kSwIH * 69);
            const int row = r / 69, col = r - row * 69;
            const int iy = iy0 + row, ix = ix0 + col;
            xv[e] = (i < 3 * kSwIH * 69 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? xb[(long long)ci * HW + iy * p.W + ix] : 0.f;
        }
        // the tile of dY: 64 channels x 4 rows x 32 columns (zeros beyond the map): eight 16-byte loads per thread where the rows
        // allow them (OW a multiple of 4: whole 4-column groups are inside or outside the map)
        const float* dyb = p.dy + (long long)b * p.dy_bs;
        float4 dv[8];
        const bool vec = (p.OW & 3) == 0 && (p.dy_bs & 3) == 0 && (OHOW & 3) == 0 && (reinterpret_cast<unsigned long long>(p.dy) & 15) == 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = tid + e * 256;                       // (channel, row, 4-column group): 64 x 4 x 8
            const int co = i >> 5, row = (i >> 3) & 3, col = (i & 7) * 4;
            const int oy = oy0 + row, ox = ox0 + col;
            dv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < p.OH) {
                const float* src = dyb + (long long)co * OHOW + oy * p.OW + ox;
                if (vec) {
                    if (ox < p.OW) dv[e] = *reinterpret_cast<const float4*>(src);
                } else {
                    if (ox < p.OW) dv[e].x = src[0];
                    if (ox + 1 < p.OW) dv[e].y = src[1];
                    if (ox + 2 < p.OW) dv[e].z = src[2];
                    if (ox + 3 < p.OW) dv[e].w = src[3];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < kXN; ++e) {
            const int i = tid + e * 256;
            if (i < 3 * kSwIH * 69) {
                const int ci = i / (kSwIH * 69), r = i - ci * (kSwIH * 69);
                const int row = r / 69, col = r - row * 69;
                x_s[ci * kSwPlane + row * kStemIW + (col & 1) * kStemHalf + (col >> 1)] = xv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = tid + e * 256;
            const int co = i >> 5, r = ((i >> 3) & 3) * 32 + (i & 7) * 4;
            float* d = dy_s + co * kSwLD + r;
            d[0] = dv[e].x; d[1] = dv[e].y; d[2] = dv[e].z; d[3] = dv[e].w;
        }
        __syncthreads();
#pragma unroll 4
        for (int st = 0; st < kSwTH * kStemTW / 2; ++st) {
            const int px = 2 * st + lh;                                      // pixel of the tile: row px / 32, column px % 32
            const int po = (px >> 5) * 2 * kStemIW + (px & 31);              // its offset inside a patch plane
            const float a0 = dy_s[l31 * kSwLD + px], a1 = dy_s[(32 + l31) * kSwLD + px];
            const float b0 = x_s[pad0 ? kb0 : kb0 + po];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc1, 0, 0, 0);
            if (third) {
                const float b4 = x_s[pad4 ? kb4 : kb4 + po];
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wid == 0 ? a0 : a1, b4, acc2, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (the channel inside its tile), column = lane & 31 (k inside its tile)
    float* slab = p.slab + (long long)blockIdx.x * 64 * 147;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = nt0 * 32 + l31;
        if (n < 147) {
            slab[(long long)row * 147 + n] = acc0[r];
            slab[(long long)(32 + row) * 147 + n] = acc1[r];
        }
        if (third && 128 + l31 < 147) slab[(long long)(wid * 32 + row) * 147 + 128 + l31] = acc2[r];
    }
}

static bool stem_wgrad_eligible(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil) {
    return g_stem_wgrad && Cin == 3 && Cout == 64 && kh == 7 && kw == 7 && stride == 2 && pad == 3 && dil == 1 && H >= 7 && W >= 7;
}
static int stem_wgrad_grid(int B, int OH, int OW) {
    const long long tiles = (long long)B * wsdl::cdiv(OH, kSwTH) * wsdl::cdiv(OW, kStemTW);
    return (int)std::min<long long>(tiles, kSwMaxGrid);
}

static size_t wgrad_dys_bytes(int Cout, int Cin, int N, int P) {
    if (!wgrad_chunk32(Cout, Cin, N)) return 0;
    const size_t n = (size_t)wsdl::cdiv(P, 32) * Cout * w2row_bytes(g_conv_arith);
    return n < (1ull << 31) ? n : 0;
}

// column bands of a weight-gradient launch (fast kernel only): same rule as the forward kernel
static int wgrad_bands(int Cout, int Cin, int OW, int W, int kw, int stride, int pad, int dil, Band* bands) {
    int BM, BN;
    bool fast;
    wgrad_tile(Cout, Cin, &BM, &BN, &fast);
    bands[0] = Band{0, OW};
    if (!fast || !g_col_bands || stride != 1) return 1;
    if (wgrad_chunk32(Cout, Cin, kw * kw * Cin)) return 1;     // 32-pixel-chunk kernel: no bands
    return column_bands(OW, W, 1, dil, -pad, 1, kw, bands);
}

// Bytes of the pre-split dY rows ([ceil(P / 32)][Cout][128 B]) this geometry's weight gradient reads when its producer writes
// them (wsdl_bn_train_bwd dy_presplit) - 0 when the launch would not run dy_split16_kernel anyway: shapes off the fp16x2
// split kernels, the role-swapped 1x1 form, the direct-fragment kernel reading dY as fp32 (few N tiles).
size_t wsdl_conv2d_wgrad_presplit_bytes(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil) {
    int OH, OW;
    if (check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return 0;
    if (g_conv_arith != 1 && g_conv_arith != 2) return 0;
    if (wgrad_role_swap(Cin, Cout, kh, kw, stride, pad)) return 0;
    const int N = kh * kw * Cin, P = B * OH * OW;
    const size_t bytes = wgrad_dys_bytes(Cout, Cin, N, P);
    if (!bytes || P % 32 != 0) return 0;
    const long long xb = (long long)B * Cin * H * W * 4, dyb = (long long)B * Cout * OH * OW * 4;
    if (xb >= (1ll << 31) || dyb >= (1ll << 31)) return 0;          // processed in batch slices
    bool taps_aligned = true;
    for (int kx = 0; kx < kw; ++kx) taps_aligned = taps_aligned && (((kx * dil - pad) & 3) == 0);
    const bool direct = g_wgrad_direct && taps_aligned && stride == 1 && OW % 32 == 0 && W % 4 == 0 && (H * W) % 4 == 0;
    const bool dyraw = direct && g_wgrad_dyraw && (N / 128 <= 10 || g_wgrad_dyraw == 2) && (OH * OW) % 4 == 0;
    return dyraw ? 0 : bytes;
}

size_t wsdl_conv2d_wgrad_workspace(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                                   int pad, int dil) {
    int OH, OW;
    if (check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return 0;
    if (wgrad_role_swap(Cin, Cout, kh, kw, stride, pad))
        return wsdl::align_up((size_t)Cout * Cin * sizeof(float), 256) +
               wsdl_conv2d_wgrad_workspace(B, Cout, OH, OW, Cin, 1, 1, 1, 0, 1);
    const int N = kh * kw * Cin;
    Band bands[8];
    const int nb = wgrad_bands(Cout, Cin, OW, W, kw, stride, pad, dil, bands);
    const int n_live = __builtin_popcountll(live_taps(H, W, OH, OW, kh, kw, stride, pad, dil)) * Cin;
    size_t slabs = (size_t)nb * wgrad_splits(Cout, Cin, N, B * OH * OW, n_live, wgrad_tap_balance(H, OH, kh, stride, pad, dil)) *
                   Cout * N * sizeof(float);
    if (stem_wgrad_eligible(B, Cin, H, W, Cout, kh, kw, stride, pad, dil))
        slabs = std::max(slabs, (size_t)stem_wgrad_grid(B, OH, OW) * Cout * N * sizeof(float));
    return wsdl::align_up(slabs, 256) + wsdl::align_up(wgrad_dys_bytes(Cout, Cin, N, B * OH * OW), 256) +
           (g_wgrad_chan_scale ? wsdl::align_up((size_t)(Cin + Cout) * sizeof(float), 256) : 0);
}

// `defer` (wsdl_conv2d_wgrad_deferred): the slabs are left un-reduced and *defer describes the reduction (kind < 0: nothing is
// pending - the call reduced by itself, e.g. a batch processed in slices).
// per-channel operands (optional; wsdl_conv2d_wgrad_ex): x_chan / dy_chan = one maximum per channel of x / dY as published by the
// channel-resident BatchNorm kernels; dy_presplit = dY already as the kernel's fp16 rows, scaled per channel by dy_chan
struct WgradExtra {
    const float* x_chan;
    const float* dy_chan;
    const void* dy_presplit;
};
static int wgrad_impl(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                      int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                      long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax, void* ws,
                      size_t ws_bytes, wsdl_stream_t stream, wsdl_wgrad_reduce_desc* defer, WgradExtra ex = WgradExtra{}) {
    WSDL_REQUIRE(x && dy && dw && ws, "conv2d_wgrad: null pointer");
    if (defer) defer->kind = -1;
    int OH, OW;
    if (int rc = check_geom(B, Cin, H, W, Cout, kh, kw, stride, pad, dil, &OH, &OW)) return rc;
    if (wgrad_role_swap(Cin, Cout, kh, kw, stride, pad)) {
        const size_t t_bytes = wsdl::align_up((size_t)Cout * Cin * sizeof(float), 256);
        WSDL_REQUIRE(ws_bytes > t_bytes, "conv2d_wgrad: workspace too small");
        float* dwt = static_cast<float*>(ws);
        if (defer) {
            // the transposed problem's slabs, reduced AND transposed by the deferred launch (no scratch matrix in between)
            wsdl_wgrad_reduce_desc inner{};
            if (int rc = wgrad_impl(dy, x, dwt, B, Cout, OH, OW, Cin, 1, 1, 1, 0, 1, 0,
                                    dy_bs ? dy_bs : (long long)Cout * OH * OW, x_bs ? x_bs : (long long)Cin * H * W,
                                    dy_amax, x_amax, static_cast<char*>(ws) + t_bytes, ws_bytes - t_bytes, stream, &inner,
                                    WgradExtra{ex.dy_chan, ex.x_chan, nullptr}))
                return rc;
            if (inner.kind == WSDL_WGRAD_REDUCE_VEC4) {
                *defer = inner;                      // slab, S as recorded: slab[z][ci][co]
                defer->kind = WSDL_WGRAD_REDUCE_TRANSPOSED;
                defer->dw = dw;
                defer->Cout = Cout; defer->Cin = Cin; defer->T = 1;
                defer->accumulate = accumulate;
                defer->grid_x = wsdl::cdiv(Cout, 32);
                defer->nblocks = defer->grid_x * wsdl::cdiv(Cin, 32);
                WSDL_TRACE("role-swapped (dW^T), reduce + transpose deferred");
                return WSDL_OK;
            }
            // (another reduction kind, or already reduced into dwt by the inner call: finish the two-launch form here)
            if (inner.kind >= 0) {
                wsdl_wgrad_reduce_desc one = inner;
                one.block_begin = 0;
                if (int rc = wgrad_reduce_one(one, wsdl::as_stream(stream))) return rc;
            }
        } else if (int rc = wgrad_impl(dy, x, dwt, B, Cout, OH, OW, Cin, 1, 1, 1, 0, 1, 0,
                                       dy_bs ? dy_bs : (long long)Cout * OH * OW, x_bs ? x_bs : (long long)Cin * H * W,
                                       dy_amax, x_amax, static_cast<char*>(ws) + t_bytes, ws_bytes - t_bytes, stream, nullptr,
                                       WgradExtra{ex.dy_chan, ex.x_chan, nullptr}))
            return rc;
        WSDL_TRACE("role-swapped (dW^T) + transpose");
        hipLaunchKernelGGL(transpose_add_kernel, dim3(wsdl::cdiv(Cout, 32), wsdl::cdiv(Cin, 32)), dim3(32, 8), 0,
                           wsdl::as_stream(stream), dwt, dw, Cout, Cin, accumulate);
        WSDL_LAUNCH_CHECK();
        return WSDL_OK;
    }
    WgradP p{};
    p.x = x; p.dy = dy; p.slab = static_cast<float*>(ws);
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.KH = kh; p.KW = kw;
    p.stride = stride; p.pad = pad; p.dil = dil;
    p.N = kh * kw * Cin; p.P = B * OH * OW;
    p.x_bs = x_bs ? x_bs : (long long)Cin * H * W;
    p.dy_bs = dy_bs ? dy_bs : (long long)Cout * OH * OW;
    p.x_amax = x_amax;
    if (stem_wgrad_eligible(B, Cin, H, W, Cout, kh, kw, stride, pad, dil) &&
        (long long)B * wsdl::cdiv(OH, kSwTH) * wsdl::cdiv(OW, kStemTW) < (1ll << 31)) {
        const int G = stem_wgrad_grid(B, OH, OW);
        const size_t need = (size_t)G * Cout * p.N * sizeof(float);
        if (ws_bytes < need) {
            wsdl::set_error("conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
            return WSDL_EWORKSPACE;
        }
        hipStream_t s = wsdl::as_stream(stream);
        const int tiles_w = wsdl::cdiv(OW, kStemTW), tiles_h = wsdl::cdiv(OH, kSwTH);
        {
            const double flops = 2.0 * p.P * (double)Cout * p.N;
            wsdl::ProfScope prof(WSDL_PROF_WGRAD_64x128, s, flops, flops, 4.0 * ((double)B * Cin * H * W + (double)p.P * Cout + (double)G * Cout * p.N));
            WSDL_TRACE("stem_wgrad7x7s2 fp32 workgroups=%d slabs=%d", G, G);
            hipLaunchKernelGGL(stem_wgrad7x7s2_kernel, dim3(G), dim3(256), 0, s, p, tiles_w, tiles_h, B * tiles_w * tiles_h);
            WSDL_LAUNCH_CHECK();
        }
        wsdl_wgrad_reduce_desc rd{};
        rd.slab = p.slab; rd.dw = dw; rd.S = G; rd.Cout = Cout; rd.Cin = Cin; rd.T = kh * kw; rd.accumulate = accumulate;
        rd.live = ~0ull;
        const long long total = (long long)Cout * p.N;
        if (G >= 64) {
            rd.kind = WSDL_WGRAD_REDUCE_MANY16;
            rd.nblocks = (int)((total + 15) / 16);
        } else if (G >= 16) {
            rd.kind = WSDL_WGRAD_REDUCE_MANY;
            rd.nblocks = (int)((total + 63) / 64);
        } else {
            rd.kind = WSDL_WGRAD_REDUCE_PLAIN;
            rd.nblocks = (int)std::min<long long>((total + 255) / 256, 4096);
        }
        if (defer) {
            *defer = rd;
            WSDL_TRACE("wgrad_reduce slabs=%d deferred", G);
            return WSDL_OK;
        }
        WSDL_TRACE("wgrad_reduce slabs=%d", G);
        return wgrad_reduce_one(rd, s);
    }
    const unsigned long long live_all = live_taps(H, W, OH, OW, kh, kw, stride, pad, dil);
    const int S = wgrad_splits(Cout, Cin, p.N, p.P, __builtin_popcountll(live_all) * Cin, wgrad_tap_balance(H, OH, kh, stride, pad, dil));
    Band bands[8];
    const int nb = (g_wgrad_bk == 32) ? 1 : wgrad_bands(Cout, Cin, OW, W, kw, stride, pad, dil, bands);
    if (nb == 1) bands[0] = Band{0, OW};
    const int S_total = nb * S;
    p.ow0 = 0; p.own = OW; p.slab0 = 0; p.nb = 1; p.splits = S;
    const size_t need = (size_t)S_total * Cout * p.N * sizeof(float);
    if (ws_bytes < need) {
        wsdl::set_error("conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return WSDL_EWORKSPACE;
    }
    const int chunks = wsdl::cdiv(p.P, 32);
    p.chunks_per_split = wsdl::cdiv(chunks, S);
    hipStream_t s = wsdl::as_stream(stream);
    const long long xb = ((long long)(B - 1) * p.x_bs + (long long)Cin * H * W) * 4;
    const long long dyb = ((long long)(B - 1) * p.dy_bs + (long long)Cout * OH * OW) * 4;
    int tBM, tBN;
    bool fast;
    wgrad_tile(Cout, Cin, &tBM, &tBN, &fast);
    if (fast && (xb >= (1ll << 31) || dyb >= (1ll << 31))) {
        // 32-bit byte offsets in the buffer-descriptor kernels: process the batch in slices below 2 GiB, accumulating
        const long long img = std::max((long long)Cin * H * W, (long long)Cout * OH * OW) * 4;
        const long long bs = std::max(p.x_bs, p.dy_bs) * 4;
        WSDL_REQUIRE(B > 1 && img < (1ll << 31), "conv2d_wgrad: a single image of 2 GiB or more is not supported");
        long long per = ((1ll << 31) - 4 - img) / bs + 1;
        if (per < 1) per = 1;
        for (int b0 = 0; b0 < B; b0 += (int)per) {
            const int nb_img = (int)std::min<long long>(per, B - b0);
            if (int rc = wgrad_impl(x + (long long)b0 * p.x_bs, dy + (long long)b0 * p.dy_bs, dw, nb_img, Cin, H, W,
                                    Cout, kh, kw, stride, pad, dil, (accumulate || b0 > 0) ? 1 : 0, p.x_bs, p.dy_bs,
                                    x_amax, dy_amax, ws, ws_bytes, stream, nullptr, WgradExtra{ex.x_chan, ex.dy_chan, nullptr}))
                return rc;
        }
        return WSDL_OK;
    }
    unsigned long long live_mask = ~0ull;
    {
        const double flops = 2.0 * p.P * (double)Cout * p.N;
        double executed = flops;
        const size_t dys_bytes = fast ? wgrad_dys_bytes(Cout, Cin, p.N, p.P) : 0;
        const size_t dys_off = wsdl::align_up(need, 256);
        const bool chunk32 = dys_bytes && ws_bytes >= dys_off + dys_bytes;
        if (wsdl::prof_enabled() && fast) {   // tiles inside one tap: all-padding pixel chunks are skipped
            executed = 0.0;
            for (int i = 0; i < nb; ++i)
                executed += flops * ((double)bands[i].own / OW) *
                            wgrad_executed_fraction(B * OH * bands[i].own, OH, bands[i].ow0, bands[i].own, H, W, kh, kw,
                                                    stride, pad, dil, (chunk32 || g_wgrad_bk == 32) ? 32 : 16);
        }
        // which fp16x2 kernel a chunk32 launch takes (the condition of the dispatch below): the direct-fragment kernel has a
        // timing class of its own
        bool taps_aligned = true;
        for (int kx = 0; kx < kw; ++kx) taps_aligned = taps_aligned && (((kx * dil - pad) & 3) == 0);
        const bool direct = chunk32 && g_conv_arith && g_wgrad_direct && taps_aligned &&
                            stride == 1 && OW % 32 == 0 && W % 4 == 0 && (H * W) % 4 == 0 && p.x_bs % 4 == 0 &&
                            reinterpret_cast<uintptr_t>(x) % 16 == 0;
        wsdl::ProfScope prof(direct ? WSDL_PROF_WGRAD_SPLIT16D
                                    : chunk32 ? WSDL_PROF_WGRAD_SPLIT32
                                     : fast ? WSDL_PROF_WGRAD_FAST_128x128
                                            : (Cout <= 64 ? WSDL_PROF_WGRAD_64x128 : WSDL_PROF_WGRAD_128x128),
                             s, flops, executed,
                             4.0 * ((double)B * Cin * H * W + (double)p.P * Cout + (double)S_total * Cout * p.N));
        constexpr size_t lds64 = 2 * (64 + 128) * 33 * sizeof(float), lds128 = 2 * (128 + 128) * 33 * sizeof(float);
        static std::once_flag once;
        static hipError_t attr_rc = hipSuccess;
        std::call_once(once, [] {
            attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<64, 128, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
            if (attr_rc == hipSuccess)
                attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<128, 128, 2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
            if (attr_rc == hipSuccess)
                attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_fast_kernel<128, 128, 2, 32>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        });
        WSDL_HIP_CHECK(attr_rc);
        if (fast) {
            p.x_bytes = (unsigned)xb;
            p.dy_bytes = (unsigned)dyb;
            // 16-pixel chunks (35 KB of LDS, 4 workgroups per CU) beat 32-pixel chunks (68 KB, 2 per CU) on all but two
            // ASPP shapes (profiles/r01_notes.md)
            if (chunk32) {
                live_mask = live_all;            // dead taps: no workgroup writes their slab columns, the reduce skips them
                unsigned char* dys = static_cast<unsigned char*>(ws) + dys_off;
                const long long total = 2ll * wsdl::cdiv(p.P, 32) * Cout;
                dim3 grid(p.N / 128, Cout / 128, S);
                p.xcd_order = ((long long)grid.x * grid.y * grid.z) % 8 == 0 ? g_wgrad_xcd : 0;
                const dim3 sgrid((int)std::min<long long>((total + 255) / 256, 16384));
                // Per-channel scales (the range guard of the weight gradient; exact: a channel is a row / column of this GEMM's
                // output).  Where the caller has the channels' maxima - the channel-resident BatchNorm kernels publish them
                // for nothing - they are used as they are; "wgrad_chan_scale" takes the missing ones with a pre-pass (one more
                // read of that operand); an operand without either keeps its one scale per tensor (stride 0).
                const size_t cm_off = wsdl::align_up(dys_off + dys_bytes, 256);
                const bool cm_fits = ws_bytes >= cm_off + (size_t)(Cin + Cout) * sizeof(float);
                p.xa_stride = 0;
                int dy_stride = 0;
                if (g_conv_arith) {
                    float* cm = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + cm_off);
                    if (ex.x_chan) {
                        p.x_amax = x_amax = ex.x_chan;
                        p.xa_stride = 1;
                    } else if (g_wgrad_chan_scale && cm_fits) {
                        WSDL_HIP_CHECK(hipMemsetAsync(cm, 0, (size_t)Cin * sizeof(float), s));
                        hipLaunchKernelGGL(channel_amax_kernel, dim3(Cin, std::min(B, std::max(1, 1024 / Cin))), dim3(256), 0, s, x, B, Cin,
                                           H * W, p.x_bs, cm);
                        WSDL_LAUNCH_CHECK();
                        p.x_amax = x_amax = cm;
                        p.xa_stride = 1;
                    }
                    if (ex.dy_chan) {
                        dy_amax = ex.dy_chan;
                        dy_stride = 1;
                    } else if (g_wgrad_chan_scale && cm_fits) {
                        WSDL_HIP_CHECK(hipMemsetAsync(cm + Cin, 0, (size_t)Cout * sizeof(float), s));
                        hipLaunchKernelGGL(channel_amax_kernel, dim3(Cout, std::min(B, std::max(1, 1024 / Cout))), dim3(256), 0, s, dy, B,
                                           Cout, OH * OW, p.dy_bs, cm + Cin);
                        WSDL_LAUNCH_CHECK();
                        dy_amax = cm + Cin;
                        dy_stride = 1;
                    }
                }
                p.da_stride = dy_stride;
                const bool cs = p.xa_stride || dy_stride;
                // dY already in the kernel's row format (written by the BatchNorm backward that produced it, scaled per channel)
                const bool presplit = ex.dy_presplit && ex.dy_chan && g_conv_arith;
                if (presplit) dys = static_cast<unsigned char*>(const_cast<void*>(ex.dy_presplit));
                if (g_conv_arith) {
                    WSDL_REQUIRE(x_amax && dy_amax, "conv2d_wgrad: the fp16x2 split kernel needs x_amax and dy_amax");
                    // the direct-fragment kernel where every tap's column shift is a multiple of 4 elements (its two
                    // 16-byte loads per tile are then aligned): 1x1 and dilation 4 / 12 / 24 / 36 - 7-10 % faster there;
                    // with a third load for misaligned taps (dilation 1, 2) it was 0-7 % slower than the LDS-staged kernel (256
                    // registers + spills): that instantiation was removed in round 4
                    // (dilation 2 through two loops - aligned and shifted by two - measured 128 us against 112 for the LDS-staged
                    // kernel on l3.conv2: the third load's registers spill; not used)
                    WSDL_TRACE("wgrad_split16%s S=%d xcd=%d cs=%d grid=%ux%ux%u", direct ? "d" : "", S, p.xcd_order, (int)cs, grid.x, grid.y, grid.z);
                    if (direct) {
                        // dY straight from the fp32 tensor (no pre-split pass) where its rows are 16-byte aligned too
                        // - where few N tiles share a row tile of dY (1x1 convolutions up to 1280 input channels: every N tile's
                        // workgroup splits its slice again; 7-11 % faster there, 7-21 % slower on the 3x3 shapes with 36-72 N tiles)
                        const bool dyraw = !presplit && taps_aligned && g_wgrad_dyraw && (p.N / 128 <= 10 || g_wgrad_dyraw == 2) &&
                                           (OH * OW) % 4 == 0 && p.dy_bs % 4 == 0 &&
                                           reinterpret_cast<uintptr_t>(dy) % 16 == 0 && p.dy_bytes != 0;
                        WSDL_TRACE(dyraw ? "dyraw" : presplit ? "dY pre-split by its producer" : "dy_split16");
                        if (!dyraw && !presplit) {
                            hipLaunchKernelGGL(dy_split16_kernel, sgrid, dim3(256), 0, s, dy, dys, B, Cout, OH * OW, p.dy_bs, p.P, dy_amax,
                                               dy_stride);
                            WSDL_LAUNCH_CHECK();
                        }
                        if (dyraw && cs)
                            hipLaunchKernelGGL((conv_wgrad_split16d_kernel<0, true, true>), grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                        else if (dyraw)
                            hipLaunchKernelGGL((conv_wgrad_split16d_kernel<0, true>), grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                        else if (cs)
                            hipLaunchKernelGGL((conv_wgrad_split16d_kernel<0, false, true>), grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                        else
                            hipLaunchKernelGGL(conv_wgrad_split16d_kernel<0>, grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                    } else {
                        WSDL_TRACE(presplit ? "dY pre-split by its producer" : "dy_split16");
                        if (!presplit) {
                            hipLaunchKernelGGL(dy_split16_kernel, sgrid, dim3(256), 0, s, dy, dys, B, Cout, OH * OW, p.dy_bs, p.P, dy_amax,
                                               dy_stride);
                            WSDL_LAUNCH_CHECK();
                        }
                        if (cs)
                            hipLaunchKernelGGL((conv_wgrad_split16_kernel<128, 128, true>), grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                        else
                            hipLaunchKernelGGL((conv_wgrad_split16_kernel<128, 128>), grid, dim3(kThreads), 0, s, p, dys,
                                               (unsigned)dys_bytes, dy_amax);
                    }
                } else {
                    WSDL_TRACE("wgrad_split32 bf16x3 S=%d xcd=%d grid=%ux%ux%u", S, p.xcd_order, grid.x, grid.y, grid.z);
                    hipLaunchKernelGGL(dy_split_kernel<0>, sgrid, dim3(256), 0, s, dy, dys, B, Cout, OH * OW, p.dy_bs, p.P,
                                       static_cast<const float*>(nullptr));
                    WSDL_LAUNCH_CHECK();
                    hipLaunchKernelGGL((conv_wgrad_split32_kernel<128, 128, 0>), grid, dim3(kThreads), 0, s, p, dys,
                                       (unsigned)dys_bytes, static_cast<const float*>(nullptr));
                }
            } else if (tBM == 128 && tBN == 128 && g_wgrad_bk == 32) {
                dim3 grid(p.N / 128, Cout / 128, S);
                hipLaunchKernelGGL((conv_wgrad_fast_kernel<128, 128, 2, 32>), grid, dim3(kThreads), lds128, s, p);
            } else {
                WgradP q = p;                            // ONE launch: blockIdx.z = band * S + split
                q.nb = nb;
                q.splits = S;
                for (int i = 0; i < nb && nb > 1; ++i) {
                    q.b_ow0[i] = bands[i].ow0;
                    q.b_own[i] = bands[i].own;
                    q.b_cps[i] = wsdl::cdiv(wsdl::cdiv((long long)B * OH * bands[i].own, 32), S);
                }
                int rc;
                WSDL_TRACE("wgrad_fp32_fast<%d,%d> S=%d nb=%d", tBM, tBN, S, nb);
                if (tBM == 128 && tBN == 128) rc = launch_wgrad_fast<128, 128, 2>(q, s, S_total);
                else if (tBM == 128) rc = launch_wgrad_fast<128, 64, 2>(q, s, S_total);
                else if (tBN == 128) rc = launch_wgrad_fast<64, 128, 1>(q, s, S_total);
                else rc = launch_wgrad_fast<64, 64, 2>(q, s, S_total);
                if (rc) return rc;
            }
        } else if (Cout <= 64) {
            WSDL_TRACE("wgrad_fp32_generic<64,128> S=%d", S);
            dim3 grid(wsdl::cdiv(p.N, 128), wsdl::cdiv(Cout, 64), S);
            hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 1>), grid, dim3(kThreads), lds64, s, p);
        } else {
            WSDL_TRACE("wgrad_fp32_generic<128,128> S=%d", S);
            dim3 grid(wsdl::cdiv(p.N, 128), wsdl::cdiv(Cout, 128), S);
            hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2>), grid, dim3(kThreads), lds128, s, p);
        }
    }
    WSDL_LAUNCH_CHECK();
    // the reduction over the pixel slabs: which kernel, on how many 256-thread blocks
    wsdl_wgrad_reduce_desc rd{};
    rd.slab = p.slab; rd.dw = dw; rd.S = S_total; rd.Cout = Cout; rd.Cin = Cin; rd.T = kh * kw; rd.accumulate = accumulate;
    rd.live = live_mask;
    const long long total = (long long)Cout * p.N;
    const int T = kh * kw;
    if (T >= 2 && T <= 16 && Cout <= 65535) {
        rd.kind = WSDL_WGRAD_REDUCE_TILED;
        rd.grid_x = wsdl::cdiv(Cin, 32);
        rd.nblocks = rd.grid_x * Cout;
    } else if (T == 1 && total % 4 == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
        rd.kind = WSDL_WGRAD_REDUCE_VEC4;
        rd.nblocks = (int)std::min<long long>((total / 4 + 255) / 256, 8192);
    } else if (S_total >= 64 && total < 64 * 512) {
        rd.kind = WSDL_WGRAD_REDUCE_MANY16;        // few outputs, very many slabs: more blocks, 16 slab lanes each
        rd.nblocks = (int)((total + 15) / 16);
    } else if (S_total >= 16 && total <= (1ll << 24)) {
        rd.kind = WSDL_WGRAD_REDUCE_MANY;
        rd.nblocks = (int)((total + 63) / 64);
    } else {
        rd.kind = WSDL_WGRAD_REDUCE_PLAIN;
        rd.nblocks = (int)std::min<long long>((total + 255) / 256, 4096);
    }
    if (defer) {
        *defer = rd;
        WSDL_TRACE("wgrad_reduce slabs=%d deferred", S_total);
        return WSDL_OK;
    }
    WSDL_TRACE("wgrad_reduce slabs=%d", S_total);
    return wgrad_reduce_one(rd, s);
}

int wsdl_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                      int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                      long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax, void* ws,
                      size_t ws_bytes, wsdl_stream_t stream) {
    return wgrad_impl(x, dy, dw, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, accumulate, x_bs, dy_bs, x_amax, dy_amax, ws,
                      ws_bytes, stream, nullptr);
}

int wsdl_conv2d_wgrad_deferred(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                               int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax, void* ws,
                               size_t ws_bytes, wsdl_wgrad_reduce_desc* desc, wsdl_stream_t stream) {
    WSDL_REQUIRE(desc != nullptr, "conv2d_wgrad_deferred: null descriptor");
    return wgrad_impl(x, dy, dw, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, accumulate, x_bs, dy_bs, x_amax, dy_amax, ws,
                      ws_bytes, stream, desc);
}

int wsdl_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                         int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                         long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax,
                         const float* x_chan_amax, const float* dy_chan_amax, const void* dy_presplit, void* ws,
                         size_t ws_bytes, wsdl_wgrad_reduce_desc* desc_or_null, wsdl_stream_t stream) {
    WSDL_REQUIRE(!dy_presplit || dy_chan_amax, "conv2d_wgrad_ex: dy_presplit comes with dy_chan_amax (the scales it was written with)");
    WSDL_REQUIRE(!dy_presplit || wsdl_conv2d_wgrad_presplit_bytes(B, Cin, H, W, Cout, kh, kw, stride, pad, dil) != 0,
                 "conv2d_wgrad_ex: this geometry's weight gradient does not read pre-split dY (wsdl_conv2d_wgrad_presplit_bytes)");
    return wgrad_impl(x, dy, dw, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, accumulate, x_bs, dy_bs, x_amax, dy_amax, ws,
                      ws_bytes, stream, desc_or_null, WgradExtra{x_chan_amax, dy_chan_amax, dy_presplit});
}

int wsdl_wgrad_reduce_multi(const wsdl_wgrad_reduce_desc* desc, int n, int total_blocks, wsdl_stream_t stream) {
    WSDL_REQUIRE(desc && n > 0 && total_blocks > 0, "wgrad_reduce_multi: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(256), 0, wsdl::as_stream(stream), desc, n);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_multi_amax(const float* const* ptrs, const long long* counts, int n, float* out, wsdl_stream_t stream) {
    WSDL_REQUIRE(ptrs && counts && out && n > 0 && n <= 65535, "multi_amax: bad arguments");
    WSDL_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * (size_t)n, wsdl::as_stream(stream)));
    hipLaunchKernelGGL(multi_amax_kernel, dim3(64, n), dim3(256), 0, wsdl::as_stream(stream), ptrs, counts, out);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_amax(const float* x, int B, long long per_image, long long x_bs, float* out, int zero_first,
              wsdl_stream_t stream) {
    WSDL_REQUIRE(x && out && B > 0 && per_image > 0, "amax: bad arguments");
    if (!x_bs) x_bs = per_image;
    const long long total = (long long)B * per_image;
    if (zero_first) WSDL_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float), wsdl::as_stream(stream)));
    hipLaunchKernelGGL(amax_kernel, dim3((int)std::min<long long>((total + 1023) / 1024, 2048)), dim3(256), 0,
                       wsdl::as_stream(stream), x, per_image, x_bs, total, out);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

int wsdl_bias_grad(const float* dy, float* dbias, int B, int C, int HW, long long dy_bs, int accumulate,
                   wsdl_stream_t stream) {
    WSDL_REQUIRE(dy && dbias && B > 0 && C > 0 && HW > 0, "bias_grad: bad arguments");
    if (!dy_bs) dy_bs = (long long)C * HW;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, wsdl::as_stream(stream), dy, dbias, B, C,
                       HW, dy_bs, accumulate);
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
