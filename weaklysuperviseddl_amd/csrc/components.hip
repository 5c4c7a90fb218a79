// components.hip - keep_largest (TraditionalModel/PsuedoMasks.py:15-21) for a batch of masks on the device.
//
// The reference labels the 8-connected components of one mask on the host (skimage `label` + `regionprops`) and keeps
// the one with the largest area, the FIRST such label on ties - labels are numbered in raster order of a component's
// first pixel - and returns an all-zero mask unchanged.  Here: one workgroup per image, the label image resident in LDS
// (16-bit pixel indices: 224 x 224 = 50176 of them are 98 KB of the CU's 160 KB; larger masks keep 32-bit labels in the
// caller's workspace and go through the same code).
//
//   1. every foreground pixel gets the index of the first pixel of its horizontal run (one thread per row);
//   2. union-find over the run labels: a pixel whose lower neighbours (SW, S, SE) belong to another tree hangs the larger
//      root under the smaller one.  Parents always have smaller indices than their children, so the plain (non-atomic)
//      stores can at worst lose a union to a concurrent one - it is found again in the next sweep; sweeps repeat until one
//      changes nothing (each changing sweep lowers at least one parent: it terminates), then every pixel points at its
//      root = the smallest pixel index of its component = the reference's label order;
//   3. areas: the first pixel of every run adds the run length to its root's counter (integer atomics: order-free);
//   4. the largest (area, then smallest root) wins by a workgroup reduction; out = (label == winner).
// Bit-exact against the reference's function (tests/golden/keep_largest.npz) and scipy.ndimage on random masks.
#include "common.h"

namespace {

constexpr int kThreads = 1024;

template <typename LT>
__global__ __launch_bounds__(kThreads) void keep_largest_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out,
                                                                 int H, int W, LT* __restrict__ glabels,
                                                                 unsigned* __restrict__ gcount) {
    extern __shared__ unsigned char smem_raw[];
    __shared__ int s_changed;
    __shared__ unsigned long long s_key[kThreads / 64];
    const int HW = H * W, t = threadIdx.x;
    const long long base = (long long)blockIdx.x * HW;
    LT* L = glabels ? glabels + base : reinterpret_cast<LT*>(smem_raw);
    unsigned* cnt = gcount + base;
    const LT kNone = (LT)~(LT)0;
    mask += base;
    out += base;

    for (int i = t; i < HW; i += kThreads) L[i] = mask[i] ? (LT)i : kNone;
    __syncthreads();
    // 1. horizontal runs
    for (int r = t; r < H; r += kThreads) {
        LT start = kNone;
        for (int x = 0, i = r * W; x < W; ++x, ++i) {
            if (L[i] == kNone) {
                start = kNone;
            } else if (start == kNone) {
                start = (LT)i;
            } else {
                L[i] = start;
            }
        }
    }
    __syncthreads();
    // 2. unions across rows, to a fixed point
    auto find = [&](int i) {
        int p = (int)L[i];
        while (p != i) {
            i = p;
            p = (int)L[i];
        }
        return i;
    };
    for (int sweep = 0; sweep <= HW; ++sweep) {      // a changing sweep lowers a parent: far fewer than HW of them
        if (t == 0) s_changed = 0;
        __syncthreads();
        for (int i = t; i < HW - W; i += kThreads) {
            if (L[i] == kNone) continue;
            const int x = i % W;
            int ri = -1;
            auto join = [&](int j) {
                if (L[j] == kNone) return;
                if (ri < 0) ri = find(i);
                const int rj = find(j);
                if (ri == rj) return;
                const int lo = ri < rj ? ri : rj, hi = ri < rj ? rj : ri;
                L[hi] = (LT)lo;
                ri = lo;
                s_changed = 1;
            };
            if (L[i + W] != kNone) {
                join(i + W);                          // SW and SE are then in S's run
            } else {
                if (x > 0) join(i + W - 1);
                if (x < W - 1) join(i + W + 1);
            }
        }
        __syncthreads();
        const int changed = s_changed;
        for (int i = t; i < HW; i += kThreads)
            if (L[i] != kNone) L[i] = (LT)find(i);     // concurrent readers see a parent or the root: both ancestors
        __syncthreads();
        if (!changed) break;
    }
    // 3. areas per root
    for (int i = t; i < HW; i += kThreads)
        if ((int)L[i] == i) __hip_atomic_store(&cnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = t; i < HW; i += kThreads) {
        if (L[i] == kNone) continue;
        const int x = i % W;
        if (x > 0 && L[i - 1] != kNone) continue;       // not the first pixel of its run
        int len = 1;
        while (x + len < W && L[i + len] != kNone) ++len;
        atomicAdd(&cnt[(int)L[i]], (unsigned)len);
    }
    __syncthreads();
    // 4. the winner: largest area, smallest root on ties
    unsigned long long key = 0;
    for (int i = t; i < HW; i += kThreads) {
        if ((int)L[i] != i) continue;
        const unsigned a = __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long k = ((unsigned long long)a << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
    }
    if ((t & 63) == 0) s_key[t >> 6] = key;
    __syncthreads();
    key = s_key[0];
    for (int i = 1; i < kThreads / 64; ++i) key = s_key[i] > key ? s_key[i] : key;
    const bool any = (key >> 32) != 0;
    const int winner = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    for (int i = t; i < HW; i += kThreads) out[i] = (any && L[i] != kNone && (int)L[i] == winner) ? 1 : 0;
}

constexpr size_t kLdsLabelLimit = 65535;      // 16-bit labels with 0xFFFF as "background"

}  // namespace

extern "C" {

size_t wsdl_keep_largest_workspace(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0 || (long long)h * w > (1ll << 30)) return 0;
    const size_t hw = (size_t)h * w;
    return (size_t)n * hw * sizeof(unsigned) * (hw > kLdsLabelLimit ? 2 : 1);
}

int wsdl_keep_largest(const uint8_t* mask, uint8_t* out, int n, int h, int w, void* ws, size_t ws_bytes,
                      wsdl_stream_t stream) {
    WSDL_REQUIRE(mask && out && ws && n > 0 && h > 0 && w > 0, "keep_largest: bad arguments");
    const size_t need = wsdl_keep_largest_workspace(n, h, w);
    WSDL_REQUIRE(need != 0, "keep_largest: mask of %d x %d pixels is too large", h, w);
    if (ws_bytes < need) {
        wsdl::set_error("keep_largest: workspace %zu < %zu bytes", ws_bytes, need);
        return WSDL_EWORKSPACE;
    }
    WSDL_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 4 == 0, "keep_largest: workspace must be 4-byte aligned");
    hipStream_t s = wsdl::as_stream(stream);
    const size_t hw = (size_t)h * w;
    unsigned* cnt = static_cast<unsigned*>(ws);
    if (hw <= kLdsLabelLimit) {
        static bool attr_set = false;
        if (!attr_set) {
            WSDL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(keep_largest_kernel<unsigned short>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsLabelLimit * 2 + 2)));
            attr_set = true;
        }
        hipLaunchKernelGGL(keep_largest_kernel<unsigned short>, dim3(n), dim3(kThreads), hw * sizeof(unsigned short), s,
                           mask, out, h, w, static_cast<unsigned short*>(nullptr), cnt);
    } else {
        hipLaunchKernelGGL(keep_largest_kernel<unsigned>, dim3(n), dim3(kThreads), 0, s, mask, out, h, w,
                           cnt + (size_t)n * hw, cnt);
    }
    WSDL_LAUNCH_CHECK();
    return WSDL_OK;
}

}  // extern "C"
