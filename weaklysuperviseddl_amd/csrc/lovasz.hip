// lovasz.hip - Lovasz-softmax loss, forward + gradient (reference TraditionalModel/LossFunctions/Lovasz-Softmax_Loss.py:
// lovasz_grad :11-23, lovasz_softmax_flat :164-192, flatten_probas :195-211; the optional loss of
// train_segmentation_model, SegmentationModel.py:103-105).
//
// Per class c:  e_i = |[label_i == c] - p_{i,c}|  over all P = B*H*W pixels; sort descending; with G = #foreground and
// the running counts F_k (foreground) / N_k (background) of the first k+1 sorted pixels, J_k = 1 - (G - F_k) / (G + N_k);
// loss_c = sum_k e_(k) (J_k - J_{k-1}) (J_{-1} = 0); the Jaccard differences are constants of the sort order, so
// d loss_c / d p_{i,c} = -sign([label_i == c] - p_{i,c}) (J_k - J_{k-1}) at the pixel's rank k.  The loss is the mean over
// the classes that occur ('present') or over all of them.
//
// On the device: one stable radix sort of (key = e as its bit pattern - non-negative floats order like their bits,
// value = pixel index + flag bits) per class (rocPRIM: a sort is library work, like a plain GEMM), one inclusive scan of
// the packed (F, N) counts, one pass that forms the differences, the dot product (fixed-order two-stage sum: bitwise
// reproducible) and scatters the gradient to the pixels.  HBM-bound: ~70 bytes per pixel and class.
// Ignored pixels (label == ignore) get the smallest key and count for neither F nor N: the sums never see them and
// their gradient is exactly zero - the same as removing them (flatten_probas).  Ties in e are ordered by pixel index
// (the reference: whatever torch.sort does); the loss does not depend on the order inside a tie.
#include <algorithm>
#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "common.h"

namespace {

constexpr int kThreadsL = 256;
constexpr unsigned kFg = 0x80000000u, kValid = 0x40000000u, kIdx = 0x3fffffffu;
constexpr int kParts = 1024;

// counts[c] = pixels of class c (labels outside [0, C) and the ignored label are in no class).  One global atomic per
// class and workgroup: per-thread counters for up to 8 classes (a pixel-wise atomic onto two addresses took 16 ms for
// two million pixels), a workgroup histogram in LDS beyond that.
__global__ void lov_hist_kernel(const int64_t* __restrict__ labels, long long P, int C, long long ignore, int* __restrict__ counts) {
    __shared__ int h[1024];
    const bool small = C <= 8;
    int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!small)
        for (int c = threadIdx.x; c < C && c < 1024; c += blockDim.x) h[c] = 0;
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const long long l = labels[i];
        if (l < 0 || l >= C || l == ignore) continue;
        if (small) {
#pragma unroll
            for (int c = 0; c < 8; ++c) mine[c] += (l == c) ? 1 : 0;
        } else if (l < 1024) {
            atomicAdd(&h[(int)l], 1);
        } else {
            atomicAdd(&counts[(int)l], 1);
        }
    }
    if (small) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int v = mine[c];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((threadIdx.x & 63) == 0 && c < C && v) atomicAdd(&counts[c], v);
        }
    } else {
        __syncthreads();
        for (int c = threadIdx.x; c < C && c < 1024; c += blockDim.x)
            if (h[c]) atomicAdd(&counts[c], h[c]);
    }
}

// norm[0] = 1 / (number of classes averaged over), norm[1 + c] = that if class c takes part, else 0
__global__ void lov_norm_kernel(const int* __restrict__ counts, int C, int all, float* __restrict__ norm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int n = 0;
        for (int c = 0; c < C; ++c) n += (all || counts[c] > 0) ? 1 : 0;
        const float inv = n > 0 ? 1.f / (float)n : 0.f;
        norm[0] = inv;
        for (int c = 0; c < C; ++c) norm[1 + c] = (all || counts[c] > 0) ? inv : 0.f;
    }
}

__global__ void lov_keys_kernel(const float* __restrict__ probas, const int64_t* __restrict__ labels, long long P, int HW,
                                int C, int c, long long ignore, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const long long l = labels[i];
        const long long b = i / HW, r = i - b * HW;
        const float p = probas[(b * C + c) * HW + r];
        const bool valid = l != ignore;
        const bool fg = valid && l == c;
        const float e = fabsf((fg ? 1.f : 0.f) - p);
        keys[i] = valid ? __float_as_uint(e) : 0u;
        vals[i] = (unsigned)i | (fg ? kFg : 0u) | (valid ? kValid : 0u);
    }
}

// packed running counts: foreground in the low 32 bits, valid background in the high 32 bits
__global__ void lov_flags_kernel(const unsigned* __restrict__ vals, long long P, unsigned long long* __restrict__ packed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const unsigned v = vals[i];
        packed[i] = (v & kFg) ? 1ull : ((v & kValid) ? (1ull << 32) : 0ull);
    }
}

__device__ __forceinline__ float jaccard(unsigned long long cum, float G) {
    const float F = (float)(unsigned)(cum & 0xffffffffull), N = (float)(unsigned)(cum >> 32);
    return 1.f - (G - F) / (G + N);
}

__global__ void lov_apply_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                 const unsigned long long* __restrict__ cum, long long P, int HW, int C, int c,
                                 const int* __restrict__ counts, const float* __restrict__ norm,
                                 const float* __restrict__ probas, float* __restrict__ dprobas, double* __restrict__ parts) {
    __shared__ double sm[16];
    const float G = (float)counts[c];
    const float w = norm[1 + c];
    double acc = 0.0;
    // contiguous runs per block: the partial sums add neighbours first
    const long long per = (P + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per, hi = lo + per < P ? lo + per : P;
    for (long long k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        const unsigned v = vals[k];
        if (!(v & kValid)) continue;                       // ignored pixel: no term, zero gradient (dprobas pre-zeroed)
        const float jk = jaccard(cum[k], G);
        const float jp = k > 0 ? jaccard(cum[k - 1], G) : 0.f;
        const float g = k > 0 ? jk - jp : jk;
        const float e = __uint_as_float(keys[k]);
        acc += (double)e * (double)g;
        if (dprobas) {
            const long long i = v & kIdx;
            const long long b = i / HW, r = i - b * HW;
            const long long at = (b * C + c) * HW + r;
            const float d = ((v & kFg) ? 1.f : 0.f) - probas[at];
            dprobas[at] = d > 0.f ? -g * w : (d < 0.f ? g * w : 0.f);
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)blockDim.x / 64; ++i) s += sm[i];
        parts[blockIdx.x] = s;
    }
}

__global__ void lov_finalize_kernel(const double* __restrict__ parts, int nparts, int c, const float* __restrict__ norm,
                                    float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nparts; ++i) s += parts[i];
        *loss += (float)(s * (double)norm[1 + c]);
    }
}

struct LovLayout {
    size_t keys_in, keys_out, vals_in, vals_out, packed, cum, parts, counts, norm, temp, temp_bytes, total;
};

int lov_layout(long long P, int C, LovLayout* L) {
    size_t sort_bytes = 0, scan_bytes = 0;
    if (rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                       (unsigned*)nullptr, (size_t)P, 0, 32, nullptr) != hipSuccess)
        return -1;
    if (rocprim::inclusive_scan(nullptr, scan_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)P,
                                rocprim::plus<unsigned long long>(), nullptr) != hipSuccess)
        return -1;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += wsdl::align_up(bytes, 256);
        return at;
    };
    L->keys_in = take((size_t)P * 4);
    L->keys_out = take((size_t)P * 4);
    L->vals_in = take((size_t)P * 4);
    L->vals_out = take((size_t)P * 4);
    L->packed = take((size_t)P * 8);
    L->cum = take((size_t)P * 8);
    L->parts = take(kParts * sizeof(double));
    L->counts = take((size_t)C * sizeof(int));
    L->norm = take((size_t)(C + 1) * sizeof(float));
    L->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    L->temp = take(L->temp_bytes);
    L->total = o;
    return 0;
}

}  // namespace

extern "C" {

size_t wsdl_lovasz_softmax_workspace(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    LovLayout L;
    if (lov_layout((long long)B * H * W, C, &L)) return 0;
    return L.total;
}

int wsdl_lovasz_softmax_fwd_bwd(const float* probas, const int64_t* labels, float* loss, float* dprobas, int B, int C,
                                int H, int W, int classes_all, long long ignore_label, void* ws, size_t ws_bytes,
                                wsdl_stream_t stream) {
    WSDL_REQUIRE(probas && labels && loss && ws, "lovasz_softmax: null pointer");
    WSDL_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "lovasz_softmax: bad shape");
    const long long P = (long long)B * H * W;
    WSDL_REQUIRE(P < (1ll << 30), "lovasz_softmax: at most 2^30 - 1 pixels (the pixel index shares a word with two flags)");
    LovLayout L;
    WSDL_REQUIRE(lov_layout(P, C, &L) == 0, "lovasz_softmax: rocPRIM size query failed");
    if (ws_bytes < L.total) {
        wsdl::set_error("lovasz_softmax: workspace %zu < %zu bytes", ws_bytes, L.total);
        return WSDL_EWORKSPACE;
    }
    hipStream_t s = wsdl::as_stream(stream);
    char* base = static_cast<char*>(ws);
    unsigned* keys_in = reinterpret_cast<unsigned*>(base + L.keys_in);
    unsigned* keys_out = reinterpret_cast<unsigned*>(base + L.keys_out);
    unsigned* vals_in = reinterpret_cast<unsigned*>(base + L.vals_in);
    unsigned* vals_out = reinterpret_cast<unsigned*>(base + L.vals_out);
    unsigned long long* packed = reinterpret_cast<unsigned long long*>(base + L.packed);
    unsigned long long* cum = reinterpret_cast<unsigned long long*>(base + L.cum);
    double* parts = reinterpret_cast<double*>(base + L.parts);
    int* counts = reinterpret_cast<int*>(base + L.counts);
    float* norm = reinterpret_cast<float*>(base + L.norm);
    void* temp = base + L.temp;
    const int HW = H * W;
    const int blocks = (int)std::min<long long>((P + kThreadsL - 1) / kThreadsL, 4096);
    const int ablocks = (int)std::min<long long>((P + kThreadsL - 1) / kThreadsL, kParts);

    wsdl::plan_poison("wsdl_lovasz_softmax_fwd_bwd sorts and scans through rocPRIM, whose launches a plan does not see");
    WSDL_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)C * sizeof(int), s));
    WSDL_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), s));
    if (dprobas) WSDL_HIP_CHECK(hipMemsetAsync(dprobas, 0, (size_t)P * C * sizeof(float), s));
    hipLaunchKernelGGL(lov_hist_kernel, dim3(std::min(blocks, 1024)), dim3(kThreadsL), 0, s, labels, P, C, ignore_label, counts);
    hipLaunchKernelGGL(lov_norm_kernel, dim3(1), dim3(64), 0, s, counts, C, classes_all, norm);
    WSDL_LAUNCH_CHECK();
    for (int c = 0; c < C; ++c) {
        hipLaunchKernelGGL(lov_keys_kernel, dim3(blocks), dim3(kThreadsL), 0, s, probas, labels, P, HW, C, c, ignore_label,
                           keys_in, vals_in);
        WSDL_LAUNCH_CHECK();
        size_t tb = L.temp_bytes;
        WSDL_HIP_CHECK(rocprim::radix_sort_pairs_desc(temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)P, 0, 32, s));
        hipLaunchKernelGGL(lov_flags_kernel, dim3(blocks), dim3(kThreadsL), 0, s, vals_out, P, packed);
        WSDL_LAUNCH_CHECK();
        tb = L.temp_bytes;
        WSDL_HIP_CHECK(rocprim::inclusive_scan(temp, tb, packed, cum, (size_t)P, rocprim::plus<unsigned long long>(), s));
        hipLaunchKernelGGL(lov_apply_kernel, dim3(ablocks), dim3(kThreadsL), 0, s, keys_out, vals_out, cum, P, HW, C, c,
                           counts, norm, probas, dprobas, parts);
        hipLaunchKernelGGL(lov_finalize_kernel, dim3(1), dim3(64), 0, s, parts, ablocks, c, norm, loss);
        WSDL_LAUNCH_CHECK();
    }
    return WSDL_OK;
}

}  // extern "C"
