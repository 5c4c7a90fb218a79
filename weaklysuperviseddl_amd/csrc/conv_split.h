// conv_split.h - the implicit-GEMM convolution on the 16-bit matrix core with fp32-equivalent accuracy
// (included by conv_igemm.hip inside its anonymous namespace; shares ConvP and the host-side tile logic).
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate.  Every fp32 operand is split EXACTLY into a few
// 16-bit pieces and a product a*b is evaluated as the partial products that matter, each an exact 16-bit x 16-bit
// product accumulated in fp32 by v_mfma_f32_32x32x16_{f16,bf16}.  Three arithmetics (template parameter AR):
//
//   AR = 1, "fp16x2" (default): x*s = h + l, h = fp16(x*s), l = fp16(x*s - h) - 2 x 11 significant bits, residual
//       <= 2^-22 |x| - and  a*b ~= ah*bh + (ah*bl + al*bh): THREE MFMAs per fp32 product.  fp16 has a 5-bit exponent, so
//       each operand tensor is scaled by a power of two s (exact) that puts its largest magnitude in [2^14, 2^15): the
//       host passes a device scalar amax >= max|x| per tensor (produced by the kernel that wrote the tensor), the
//       kernel derives s from its exponent and the epilogue multiplies the accumulator by 1/(s_a*s_b).  Elements
//       within 2^17 of the tensor's maximum keep the full 22 bits; smaller ones are represented to an ABSOLUTE
//       2^-39 of the maximum (l drops into fp16's subnormal range) - far below an fp32 rounding of the dot product
//       they enter.  Measured against float64 (tools/conv_accuracy.py, tests/test_hip_ops.py): rms error at or below that
//       of the exact-fp32 MFMA chain for K >= 64 - the representation error (2^-22 per product, random sign) is smaller
//       than what K fp32 accumulator roundings contribute, and three accumulations per k-step round less than six.
//   AR = 2, "fp16x2s" (conv_arith = 2, forward / input gradient only - the range guard): the low piece is carried at 2^11
//       times its value, l' = fp16((x*s - h) * 2^11), which is a normal fp16 number whenever h is one: an element keeps its
//       22 bits down to 2^-29 of the tensor's maximum and is exact to 2^-50 of it below that.  The cross products then
//       carry a factor 2^11 and get their own accumulator, folded in by the epilogue: a*b ~= ah*bh + 2^-11 (ah*bl' + al'*bh).
//       Same three MFMAs; 64 more accumulator registers (183 instead of 119 in the 256x128 form), which is what it costs:
//       a weight-gradient workgroup no longer fits beside it on the CU (-6.6 % on the training step, +3.7 / 4.9 % on the
//       isolated forward / input-gradient kernels; profiles/r04_notes.md).  Measured per region against float64
//       (tests/test_hip_ops.py::test_split_arithmetic_max_norm_per_channel_and_per_block): one 2^30 outlier among unit
//       activations 9e-7 (AR = 1: 1.3e-3), 2^35: 2.6e-5 (4.8e-2), images graded by 2^30 across the batch 2.8e-6 (5e-3);
//       on unit data 3.7e-7 against 5.7e-7 - the cross products no longer round into the large accumulator.
//   AR = 0, "bf16x3": x = h + m + l (3 x 8 bits cover fp32's 24; no scaling: bf16 has fp32's exponent range) and the SIX
//       products of weight >= 2^-16: ah*bh, ah*bm, am*bh, ah*bl, al*bh, am*bm (dropped terms <= 2^-23 |a*b|).  Round 1's
//       kernel; kept as wsdl_set_option("conv_arith", 0) and as the A/B partner.
//
// The kernels are power-limited (same binary 30 % faster on all-zero operands; halving the MFMAs of the bf16x3
// kernel at unchanged staging removed 26 % of its time; deeper pipelines, staggered wave roles and a three-image LDS
// ring with fragments read one chunk ahead all measured SLOWER or equal - profiles/r02_notes.md): what pays is energy per
// fp32-equivalent FLOP, which is what AR = 1 halves (MFMAs) and cuts by a third (LDS bytes, weight bytes).
//
// Data path: weights are split once per optimiser step by the layout kernel ([k/16][row][piece][16], so a K-chunk of
// a row tile is one contiguous run copied to LDS with 16-byte loads); activations are split by the loading thread
// between the global load and the LDS store.  LDS rows are (BK/16) x NP x 32 bytes + 16 bytes of padding: a row stride
// that is an odd multiple of 16 bytes makes every ds_read_b128 lane group (16 lanes: 16 different rows) hit 16 distinct
// 16-byte slots of the 256-byte read bank row.  The STORES bank over 128 bytes and go by groups of 8 (b128) / 16 (b64)
// consecutive lanes: the weight stores are conflict-free because of which unit each lane carries - 8 consecutive rows per
// b128 group (round 6; rounds 1-5 stored in unit order: a 2-way conflict per store).  The 8-byte activation stores of the
// 256 x 128 form keep their 2-way conflict (lanes i and i + 8 of a group): the mapping that removes it was built and measured
// slower - see kPairB in conv_igemm_split_body.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) -> packed bf16 pairs of the three pieces
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 s = {r[0] - __builtin_bit_cast(float, m << 16), r[1] - __builtin_bit_cast(float, m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2));
}

// fp16x2 with the LOW piece carried at 2^11 times its value ("AR = 2"): l' = fp16((x - h) * 2^11).  |x - h| <= 2^-11 |h|, so l'
// is a normal fp16 number whenever h is one - an element keeps its 22 bits down to 2^-29 of the tensor's maximum (2^-18 with the
// unscaled l of AR = 1, whose low piece drops into fp16's subnormals there) and is represented to an absolute 2^-50 of the
// maximum below that (2^-39).  The cross products ah*bl' + al'*bh then carry a factor 2^11 and are accumulated in a second
// accumulator that the epilogue folds in:  a*b ~= ah*bh + 2^-11 (ah*bl' + al'*bh).  Same three MFMAs per product.
__device__ __forceinline__ void split2hs(float x0, float x1, unsigned& h, unsigned& l) {
    f32x2 v = {x0, x1};
    const half2v hv = __builtin_convertvector(v, half2v);
    h = __builtin_bit_cast(unsigned, hv);
    f32x2 r = {(x0 - (float)hv[0]) * 2048.f, (x1 - (float)hv[1]) * 2048.f};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
}

template <int AR> struct SplitArith;
template <> struct SplitArith<0> {
    static constexpr int NP = 3;
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct SplitArith<1> {
    static constexpr int NP = 2;
    using frag = half8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct SplitArith<2> : SplitArith<1> {};
// the partial products of one k-step, smallest terms first
template <int AR, typename F>
__device__ __forceinline__ f32x16 split_products(const F (&a)[SplitArith<AR>::NP], const F (&b)[SplitArith<AR>::NP], f32x16 c) {
    using A = SplitArith<AR>;
    if constexpr (AR == 0) {
        c = A::mma(a[2], b[0], c);
        c = A::mma(a[0], b[2], c);
        c = A::mma(a[1], b[1], c);
        c = A::mma(a[1], b[0], c);
        c = A::mma(a[0], b[1], c);
        c = A::mma(a[0], b[0], c);
    } else {
        c = A::mma(a[1], b[0], c);
        c = A::mma(a[0], b[1], c);
        c = A::mma(a[0], b[0], c);
    }
    return c;
}

constexpr int split_k16_bytes(int AR) { return AR ? 64 : 96; }     // one row's 16 k values: NP pieces x 16 x 2 bytes
// the layout buffers end in a 16-byte trailer holding the tensor's amax (AR = 1)
__host__ __device__ constexpr long long split_layout_bytes(int AR, long long k_total, long long rows) {
    return (k_total / 16) * rows * split_k16_bytes(AR) + 16;
}

// One chunk of global loads is in flight in registers while the previous one is computed.
// NT threads: 256 (4 waves, two workgroups per CU) or 512 (8 waves, one 256x128 workgroup per CU: every activation
// element is split by half as many workgroups and the weight tile is shared by twice the pixels... per FLOP).
//
// XCD-aware tile order (p.xcd_py > 0): workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.
// The kernel re-labels its workgroup so that XCD c owns a (pixel-group, row-group) rectangle of the tile grid:
// py row groups x 8/py pixel groups.  Each XCD then streams 1/py of the weights and py/8 of the activations instead of
// all the weights and 1/8 of the activations; the host picks py per launch from the two operands' byte counts.
//
// (Weights by LDS-DMA - `buffer_load_dwordx4 ... lds` - were a template option in round 2: bit-identical, 2-3 % slower; round 3
// measured why: that path feeds a CU ~30 GB/s, the tile needs more.  Removed; profiles/r02_notes.md, r03_notes.md.)
//
// MF (fp16x2, K chunk 32): v_mfma_f32_16x16x32_f16 instead of 32x32x16 - one MFMA spans the chunk's 32 k values; the chip
// holds a higher clock on this shape under the power limit (weight gradient: -8 % per launch, see conv_wgrad_split16_kernel).
// LDS rows become 128 bytes, unit u = 4 piece + (k / 8) of 16 bytes stored at unit u ^ (row & 7): conflict-free for the
// operand reads (lane l: row l & 15, k-group l >> 4 - checked against the b128 lane groups) and for the activations'
// stores (8 consecutive pixel rows, one unit).
//
// MS (multi-source; round 5): the output tile accumulates over SEVERAL convolutions that share their output - the input
// gradient of a tensor that fed several convolutions (ASPP: dX = sum over the 1x1 and the three dilated branches of
// dgrad(dY_b, W_b)).  One launch, one accumulator, one store instead of a chain of launches that each read the previous
// sum back and write it again (4 x 134 MB written, 3 x 134 MB read at B = 16), and the branches' uneven padding-tap
// counts average out inside every workgroup.  Source 0 is ConvP's own (x, wt, ...), sources 1 .. p.nsrc are p.src[]; each
// has its operand tensor, weights, scales and tap geometry; rows (Cout), the output and the input map's size are shared.
// Between sources the pipeline drains and the accumulators move to the next source's power-of-two units (exact).
//
// The body is a device function of (problem, tile indices, grid extent): conv_igemm_split_kernel runs it on its own grid,
// conv_igemm_split_group_kernel (round 5) on a grid that strings several problems together (ASPP's four branches in one launch).
// IL (interleaved steady state; 32x32x16 form, K chunk 16, AR = 1): counters of the 256 x 128 kernel (profiles/r05_notes.md) show its
// waves issuing 29 % of their cycles, waiting for the issue port 41 % (the SIMD's other wave holds the matrix pipe) and parked 30 %,
// with the pipe busy 40 %: both waves of a SIMD run the SAME phase (they meet at the chunk's barrier), so staging never falls
// into the other wave's MFMAs.  IL makes every wave's instruction stream alternate - one MFMA, a few staging instructions
// (the next chunk's split and LDS stores, the loads of the one after) - so that whichever wave does not hold the pipe has
// non-matrix work to issue.  Branch-free steady state (the scheduler cannot interleave across branches); the last two chunks
// run the plain loop.
template <int BM, int BN, int WM, int BK, int NT, int AR, bool MF, bool MS, bool IL>
__device__ __forceinline__ void conv_igemm_split_body(const ConvP& p, const int bx_in, const int by_in, const int bz,
                                                      const int grid_x, const int grid_y) {
    static_assert(!MF || (AR >= 1 && BK == 32), "16x16x32 form: fp16x2, K chunk 32");
    using Ar = SplitArith<AR>;
    using frag = typename Ar::frag;
    constexpr int NP = Ar::NP;
    constexpr int K16B = split_k16_bytes(AR);
    constexpr int WN = (NT / 64) / WM;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN, "bad tile");
    static_assert(BK == 16 || BK == 32, "K chunk");
    constexpr int KS = BK / 16;                          // MFMA k-steps per chunk
    constexpr int ROW = MF ? 128 : KS * K16B + 16;       // LDS row stride in bytes (odd multiple of 16; MF: swizzled units)
    constexpr int UPR = K16B / 16;                       // 16-byte units per row of one k16 slab
    constexpr int A_UPS = BM * UPR;                      // 16-byte units of one k16 slab of the row tile
    constexpr int A_U = (KS * A_UPS + NT - 1) / NT;      // units per thread per chunk
    constexpr bool A_EXACT = A_U * NT == KS * A_UPS;
    constexpr int B_STEP = NT / BN;                      // threads sharing one pixel
    constexpr int B_PER = BK / B_STEP;                   // consecutive k (input channels) per thread
    static_assert(B_PER == 4 || B_PER == 8 || B_PER == 16 || B_PER == 32, "B tile");
    constexpr unsigned kOOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) unsigned char As[2][BM * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BN * ROW];
    __shared__ int vtaps[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    int bx = bx_in, by = by_in;
    if (p.xcd_py > 0) {
        // host guarantees: grid_x * grid_y % 8 == 0, grid_y % py == 0, grid_x % (8 / py) == 0 (and, in a group, that the
        // problem's first workgroup has a linear index that is a multiple of 8: workgroup L runs on XCD L % 8)
        const int gx = grid_x;
        const int L = by * gx + bx;
        const int xcd = L & 7, idx = L >> 3;
        const int py = p.xcd_py, px = 8 / py;
        const int lx = gx / px, ly = grid_y / py;
        if (p.xcd_rowfast) {
            // the workgroups an XCD runs at once (consecutive idx) cover ly row tiles x a few pixel tiles: they share the pixel
            // tiles' activations AND the row tiles' weights (pixel-fastest order gives them one row tile and 32 pixel tiles)
            bx = (xcd / py) * lx + idx / ly;
            by = (xcd % py) * ly + idx % ly;
        } else {
            bx = (xcd / py) * lx + idx % lx;
            by = (xcd % py) * ly + idx / lx;
        }
    }
    const int m0 = by * BM;
    int w_ow0 = p.ow0, w_own = p.own, w_tile0 = 0;
    if (p.nb > 1) {
        w_ow0 = p.b_ow0[0];
        w_own = p.b_own[0];
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < p.nb && bx >= p.b_tile0[i]) {
                w_ow0 = p.b_ow0[i];
                w_own = p.b_own[i];
                w_tile0 = p.b_tile0[i];
            }
    }
    const int W_P = p.nb > 1 ? p.B * p.OH * w_own : p.P;
    // Image-major tile order (round 6).  Which taps a pixel tile executes depends on WHERE in the image it lies (a dilated tap
    // reads only padding for the rows / columns near one border), not on which image: tiles at the same place of different
    // images run the same tap list and the same number of chunks - in lockstep, if they start together.  In pixel order the ~32
    // workgroups an XCD runs at once are the 8 places of 4 images: 8 different tap lists, the weight chunks (2 MB per tap of a
    // 2048-channel ASPP branch) are streamed by workgroups that drift apart, and every one of them misses in L2 - the
    // grouped ASPP forward fetched 3.6 GB past L2 for 0.26 GB of operands (profiles/r05_notes.md).  Taking a band's tiles image
    // fastest puts the same place of up to B images side by side: one fetch of a weight chunk serves them all.  Same tiles,
    // another order: bit-identical results.
    int lt = bx - w_tile0;
    if (p.tile_img_major) {
        const int ohw = p.OH * w_own;                        // pixels of one image inside this band
        const int tpi = ohw / BN;
        if (tpi > 1 && tpi * BN == ohw) lt = (lt % p.B) * tpi + lt / p.B;
    }
    const int n0 = lt * BN;
    const int OHOW = p.OH * p.OW;
    const int HW = p.H * p.W;

    // thread -> (pixel pl of the tile, k run kr) of the activation staging.  Four threads per pixel (the 256 x 128 form: runs of
    // 4 k, 8-byte LDS stores): lane PAIRS share a pixel, so that the 16 lanes of a ds_write_b64 group cover 8 pixels x one whole
    // 16-byte slot each - 8 distinct slots of the 128-byte store bank row.  With one k run per wave (64 consecutive pixels, the
    // round 1-5 mapping) lanes i and i + 8 hit the same 8 bytes (8 rows of 80 bytes = 5 bank rows): a 2-way conflict on every
    // store, part of the 30 % of LDS-active cycles the counters showed (profiles/r05_pmc_dominant_kernels.txt).
    // MEASURED, NOT ADOPTED (round 6, same box, three builds): the pairing removes the store conflict and costs the global side
    // more than that - a lane quad then reads 2 pixels x 2 channels instead of 16 contiguous bytes.  Training step 847.8 / 852.1
    // img/s with it alone against 853.1 / 853.4 without; the 256 x 128 launches 1.5-2.5 % slower with both mappings
    // (l4.conv2 d4 221.5 against 217.8 us).  Kept as a build option (-DWSDL_EXP_PAIRB, tools/build_variant.sh).
#if defined(WSDL_EXP_PAIRB) && !defined(WSDL_EXP_OLD_LDS_MAP)
    constexpr bool kPairB = !MF && B_PER == 4 && B_STEP == 4;
#else
    constexpr bool kPairB = false;
#endif
    const int pl = kPairB ? (tid % (2 * BN)) / 2 : tid % BN;
    const int kr = kPairB ? (tid & 1) + 2 * (tid / (2 * BN)) : tid / BN;
    const int pix = n0 + pl;
    const bool pix_ok = pix < W_P;
    int pb = 0, poh = 0, pow_ = 0;
    const int OHW = p.OH * w_own;
    if (pix_ok) {
        pb = pix / OHW;
        const int r = pix - pb * OHW;
        poh = r / w_own;
        pow_ = w_ow0 + (r - poh * w_own);
    }

    constexpr int TS = MF ? 16 : 32;                     // side of an MFMA output tile
    constexpr int TMI = BM / WM / TS, TNI = BN / WN / TS;  // tiles of a wave
    constexpr int RT = TS * TS / 64;                     // accumulator registers per tile
    using acc_t = std::conditional_t<MF, f32x4, f32x16>;
    acc_t acc[TMI][TNI];
    constexpr int LO = AR == 2 ? 1 : 0;                  // AR = 2: the cross products (2^11 too large) have their own accumulator
    acc_t acc_lo[LO ? TMI : 1][LO ? TNI : 1];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (LO) acc_lo[i][j][r] = 0.f;
            }
    // lane -> (column of the output tile, row group): D[row][col] of v_mfma_f32_32x32x16 / 16x16x32
    const int l31 = MF ? (lane & 15) : (lane & 31), lh = MF ? (lane >> 4) : (lane >> 5);
    auto acc_row = [&](int r) { return MF ? lh * 4 + r : (r & 3) + 8 * (r >> 2) + 4 * lh; };

    float out_scale = 1.f;
    int e_units = 0;                                     // MS: the accumulators hold sum * 2^e_units
    const int n_src = MS ? p.nsrc + 1 : 1;
    for (int src = 0; src < n_src; ++src) {
    // this source's operands and tap geometry (MS = false: ConvP's own, folded at compile time)
    // (selected field by field with constant indices: a pointer into the by-value parameter would move it to scratch)
#define WSDL_SRC(f) ((!MS || src == 0) ? p.f : src == 1 ? p.src[0].f : src == 2 ? p.src[1].f : p.src[2].f)
    const float* const s_x = WSDL_SRC(x);
    const float* const s_wt = WSDL_SRC(wt);
    const float* const s_amax = WSDL_SRC(x_amax);
    const int s_KH = WSDL_SRC(KH), s_KW = WSDL_SRC(KW), s_bh = WSDL_SRC(bh), s_ch = WSDL_SRC(ch);
    const int s_K = WSDL_SRC(K), s_Cin = WSDL_SRC(Cin);
    const long long s_x_bs = WSDL_SRC(x_bs);
    const unsigned s_x_bytes = WSDL_SRC(x_bytes);
#undef WSDL_SRC

    const int wbytes = (s_K / 16) * p.Cout * K16B;       // the layout without its trailer
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s_x), 0, (int)s_x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s_wt), 0, wbytes, 0x00020000);
    float xs = 1.f;
    if constexpr (AR >= 1) {
        int ex, ew;
        xs = pow2_scale(*s_amax, ex);
        (void)pow2_scale(*reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(s_wt) + wbytes), ew);
        out_scale = pow2(-(ex + ew));
        if constexpr (MS) {
            if (src > 0 && ex + ew != e_units) {         // the sum so far, in this source's units (a power of two: exact)
                const float f = pow2(ex + ew - e_units);
#pragma unroll
                for (int i = 0; i < TMI; ++i)
#pragma unroll
                    for (int j = 0; j < TNI; ++j)
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            acc[i][j][r] *= f;
                            if constexpr (LO) acc_lo[i][j][r] *= f;
                        }
            }
            e_units = ex + ew;
        }
    }
    const unsigned img_off = (unsigned)((long long)pb * s_x_bs) + (unsigned)(kr * B_PER * HW);   // elements

    auto tap_src = [&](int t, int& sp) {
        const int ti = t / s_KW, tj = t - ti * s_KW;
        const int nh = poh * p.ah + ti * s_bh + s_ch;
        const int nw = pow_ * p.ah + tj * s_bh + s_ch;
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

    const int T = s_KH * s_KW;
    const int cpt = s_Cin / BK;
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
    const int q0 = p.ksplit > 1 ? (int)((long long)nq_all * bz / p.ksplit) : 0;
    const int q1 = p.ksplit > 1 ? (int)((long long)nq_all * (bz + 1) / p.ksplit) : nq_all;
    const int nq = q1 - q0;
    __syncthreads();

    // weights: unit u of the chunk = (k-step, row, 16-byte part); a k16 slab of the row tile is contiguous
    // Which unit a lane carries: 64 consecutive units = 16 rows x 4 parts (fp16x2: 64 bytes per row and k16 slab).  The 8 lanes of
    // a ds_write_b128 group take the SAME part of 8 consecutive rows - 8 distinct 16-byte slots of the 128-byte store bank row,
    // for the padded rows (80 bytes: slot 5 row + part) and for the swizzled 128-byte rows of the MF form (unit ^ (row & 7)) alike.
    // In unit order (a lane group = two rows x four parts) the group's first and last lane are 128 bytes apart: a 2-way
    // conflict on every weight store (conv_split.h said "conflict-free" until round 6; the counters did not).  The global side
    // is unchanged: a wave still loads one contiguous 1 KB run, its lanes permuted inside it.  Same box, three builds: the
    // 16 x 16 x 32 forms 2-3 % faster per launch (l2.conv2 41.3 -> 40.1 us), the training step 853.5 / 855.1 img/s against
    // 853.1 / 853.4 in unit order: kept.
#if !defined(WSDL_EXP_OLD_LDS_MAP) && !defined(WSDL_EXP_NO_ROWA)
    constexpr bool kRowMajorA = UPR == 4 && (A_UPS % 64) == 0;
#else
    constexpr bool kRowMajorA = false;
#endif
    unsigned voff_a[A_U], lds_a[A_U];
#pragma unroll
    for (int e = 0; e < A_U; ++e) {
        const int u = tid + e * NT;
        const int ks = u / A_UPS, v = u - ks * A_UPS;
        const int j = v & 63;
        const int row = kRowMajorA ? (v >> 6) * 16 + (j & 7) + 8 * (j >> 5) : v / UPR;
        const int part = kRowMajorA ? (j >> 3) & 3 : v - row * UPR;
        voff_a[e] = (m0 + row < p.Cout && (A_EXACT || u < KS * A_UPS))
                        ? (unsigned)(ks * p.Cout * K16B + ((m0 + row) * UPR + part) * 16) : kOOB;
        if constexpr (MF)                               // part = 2 piece + half of the k16 slab -> unit 4 piece + 2 ks + half
            lds_a[e] = (unsigned)(row * ROW + (((4 * (part >> 1) + 2 * ks + (part & 1)) ^ (row & 7)) << 4));
        else
            lds_a[e] = (unsigned)(row * ROW + ks * K16B + part * 16);
    }

    u32x4 ra[A_U];
    unsigned rb[B_PER];
    int ld_vi = q0 / cpt, ld_c = q0 - (q0 / cpt) * cpt, ld_tap = 0;
    unsigned voff_b = kOOB;
    auto set_tap = [&](int vi) {
        ld_tap = __builtin_amdgcn_readfirstlane(vtaps[vi]);
        int sp;
        const bool ok = tap_src(ld_tap, sp);
        voff_b = ok ? (img_off + (unsigned)sp) * 4u : kOOB;
    };
    if (nq > 0) set_tap(ld_vi);

    auto load_next = [&]() {
        if (ld_c == cpt) {
            ld_c = 0;
            ++ld_vi;
            set_tap(ld_vi);
        }
        const int c16 = (ld_tap * s_Cin + ld_c * BK) / 16;
        const unsigned soff_a = (unsigned)(c16 * p.Cout * K16B);
#pragma unroll
        for (int e = 0; e < A_U; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_a[e], soff_a, 0);
        const unsigned soff_b = (unsigned)(ld_c * BK * HW) * 4u;
#pragma unroll
        for (int e = 0; e < B_PER; ++e)
            rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, soff_b + (unsigned)(e * HW) * 4u, 0);
        ++ld_c;
    };
    // this thread's k run [kr*B_PER, kr*B_PER + B_PER) inside the chunk -> (k-step, offset inside the 16)
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < A_U; ++e)
            if (A_EXACT || tid + e * NT < KS * A_UPS) *reinterpret_cast<u32x4*>(&As[buf][lds_a[e]]) = ra[e];
        unsigned pc[NP][B_PER / 2];
#pragma unroll
        for (int e = 0; e < B_PER / 2; ++e) {
            const float x0 = __builtin_bit_cast(float, rb[2 * e]), x1 = __builtin_bit_cast(float, rb[2 * e + 1]);
#ifdef WSDL_EXP_NOSPLIT      // timing-only build: the activations as if they arrived pre-split (no VALU between load and LDS store)
            pc[0][e] = rb[2 * e] & 0x3fff3fffu;
            pc[NP - 1][e] = rb[2 * e + 1] & 0x3fff3fffu;
#else
            if constexpr (AR == 0)
                split3(x0, x1, pc[0][e], pc[1][e], pc[2][e]);
            else if constexpr (AR == 1)
                split2h(x0 * xs, x1 * xs, pc[0][e], pc[1][e]);
            else
                split2hs(x0 * xs, x1 * xs, pc[0][e], pc[1][e]);
#endif
        }
        unsigned char* rowp = &Bs[buf][pl * ROW];
        if constexpr (MF) {
            static_assert(!MF || B_PER == 8 || B_PER == 16 || B_PER == 32, "a thread's run is whole k-groups");
#pragma unroll
            for (int g = 0; g < B_PER / 8; ++g) {
                const int kg = kr * (B_PER / 8) + g;
#pragma unroll
                for (int c = 0; c < NP; ++c)
                    *reinterpret_cast<u32x4*>(rowp + (((4 * c + kg) ^ (pl & 7)) << 4)) =
                        u32x4{pc[c][4 * g], pc[c][4 * g + 1], pc[c][4 * g + 2], pc[c][4 * g + 3]};
            }
        } else if constexpr (B_PER == 4) {
            const int k = kr * 4, off = (k / 16) * K16B + (k % 16) * 2;
#pragma unroll
            for (int c = 0; c < NP; ++c) *reinterpret_cast<u32x2*>(rowp + off + c * 32) = u32x2{pc[c][0], pc[c][1]};
        } else {
#pragma unroll
            for (int g = 0; g < B_PER / 8; ++g) {
                const int k = kr * B_PER + g * 8, off = (k / 16) * K16B + (k % 16) * 2;
#pragma unroll
                for (int c = 0; c < NP; ++c)
                    *reinterpret_cast<u32x4*>(rowp + off + c * 32) =
                        u32x4{pc[c][4 * g], pc[c][4 * g + 1], pc[c][4 * g + 2], pc[c][4 * g + 3]};
            }
        }
    };

    if (nq > 0) {
        load_next();
        store_tiles(0);
        if (nq > 1) load_next();
    }
    __syncthreads();
    auto mfma_chunk = [&](int cur) {
        if constexpr (MF) {
            const unsigned char* Ab = As[cur] + (wm * (BM / WM) + l31) * ROW;
            const unsigned char* Bb = Bs[cur] + (wn * (BN / WN) + l31) * ROW;
            const unsigned f0 = (unsigned)((lh ^ (l31 & 7)) << 4), f1 = (unsigned)(((4 + lh) ^ (l31 & 7)) << 4);
            half8 a[TMI][2], b[TNI][2];
#pragma unroll
            for (int i = 0; i < TMI; ++i) {
                a[i][0] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + f0);
                a[i][1] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + f1);
            }
#pragma unroll
            for (int j = 0; j < TNI; ++j) {
                b[j][0] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + f0);
                b[j][1] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + f1);
            }
#pragma unroll
            for (int i = 0; i < TMI; ++i)
#pragma unroll
                for (int j = 0; j < TNI; ++j) {
                    if constexpr (LO) {
                        f32x4 c = acc_lo[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
                        acc_lo[i][j] = c;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                    } else {
                        f32x4 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
        } else {
            const unsigned char* Ab = As[cur] + (wm * (MI * 32) + l31) * ROW + lh * 16;
            const unsigned char* Bb = Bs[cur] + (wn * (NI * 32) + l31) * ROW + lh * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                frag a[MI][NP], b[NI][NP];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int c = 0; c < NP; ++c)
                        a[i][c] = *reinterpret_cast<const frag*>(Ab + i * 32 * ROW + ks * K16B + c * 32);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int c = 0; c < NP; ++c)
                        b[j][c] = *reinterpret_cast<const frag*>(Bb + j * 32 * ROW + ks * K16B + c * 32);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if constexpr (!MF && LO) {
                            acc_lo[i][j] = Ar::mma(a[i][1], b[j][0], acc_lo[i][j]);
                            acc_lo[i][j] = Ar::mma(a[i][0], b[j][1], acc_lo[i][j]);
                            acc[i][j] = Ar::mma(a[i][0], b[j][0], acc[i][j]);
                        } else if constexpr (!MF) {
                            acc[i][j] = split_products<AR>(a[i], b[j], acc[i][j]);
                        }
                    }
            }
        }
    };
#ifdef WSDL_EXP_NOSTAGE          // timing-only build: both LDS images filled once (real data), the loop is barrier + ds_read + MFMA
    if (nq > 1) store_tiles(1);
    __syncthreads();
#endif
    int q_start = 0;
#if defined(WSDL_EXP_NOSTAGE) || defined(WSDL_EXP_NOMFMA)
    constexpr bool kTimingBuild = true;                  // the timing-only builds take the plain loop apart, not this one
#else
    constexpr bool kTimingBuild = false;
#endif
    if constexpr (IL && !kTimingBuild) {
        static_assert(!IL || (!MF && KS == 1 && AR == 1), "interleaved loop: 32x32x16 form, K chunk 16, fp16x2");
        // the loads of load_next() split in two: `advance` (which chunk comes next: scalar bookkeeping, a branch) at the END of an
        // iteration, `issue` (the loads themselves) inside the interleaved block
        unsigned il_soff_a = 0, il_soff_b = 0;
        auto advance = [&]() {
            if (ld_c == cpt) {
                ld_c = 0;
                ++ld_vi;
                set_tap(ld_vi);
            }
            il_soff_a = (unsigned)(((ld_tap * s_Cin + ld_c * BK) / 16) * p.Cout * K16B);
            il_soff_b = (unsigned)(ld_c * BK * HW) * 4u;
            ++ld_c;
        };
        const unsigned fra = (unsigned)((wm * (MI * 32) + l31) * ROW + lh * 16), frb = (unsigned)((wn * (NI * 32) + l31) * ROW + lh * 16);
        int q = 0;
        if (nq > 2) advance();                           // chunk 2's offsets
        for (; q + 2 < nq; ++q) {
            const int cur = q & 1;
            frag a[MI][NP], b[NI][NP];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int c = 0; c < NP; ++c) a[i][c] = *reinterpret_cast<const frag*>(&As[cur][fra + i * 32 * ROW + c * 32]);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int c = 0; c < NP; ++c) b[j][c] = *reinterpret_cast<const frag*>(&Bs[cur][frb + j * 32 * ROW + c * 32]);
            store_tiles(cur ^ 1);                        // the registers hold chunk q + 1
#pragma unroll
            for (int e = 0; e < A_U; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_a[e], il_soff_a, 0);
#pragma unroll
            for (int e = 0; e < B_PER; ++e)
                rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, il_soff_b + (unsigned)(e * HW) * 4u, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = split_products<AR>(a[i], b[j], acc[i][j]);
            // the order wanted: the fragment reads, a first piece of the staging arithmetic while they arrive, then MFMAs with the
            // rest of the staging, the LDS stores and the loads between them
            __builtin_amdgcn_sched_group_barrier(0x100, (MI + NI) * NP, 0);          // DS reads
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                       // VALU
#pragma unroll
            for (int k = 0; k < MI * NI * 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                   // VALU
                if (k < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        // an LDS store
                else if (k < 4 + A_U + B_PER) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a load
            }
            lds_barrier();
            if (q + 3 < nq) advance();                   // offsets of chunk q + 3 (its loads are issued in the next iteration)
        }
        q_start = q;
        // hand over to the plain loop: it expects ld_c / ld_tap to describe the NEXT chunk to load - nothing is left to load
        // (chunks up to nq - 1 have been issued: the registers hold chunk q_start + 1 if it exists)
    }
    for (int q = q_start; q < nq; ++q) {
        const int cur = q & 1;
#ifndef WSDL_EXP_NOSTAGE
        if (q + 1 < nq) {
            store_tiles(cur ^ 1);                        // the registers hold chunk q + 1
        }
        if (q + 2 < nq) load_next();
#endif
#ifdef WSDL_EXP_NOMFMA           // timing-only build: staging, barriers and one LDS read per chunk, no matrix work
        acc[0][0][0] += (float)As[cur][(tid * 16) % (BM * ROW)] + (float)Bs[cur][(tid * 16) % (BN * ROW)];
#else
        mfma_chunk(cur);
#endif
        lds_barrier();
    }
    }   // sources

    if (p.ksplit > 1) {
        // slabs are indexed by the pixel's position in the whole output (b, oh, ow), not inside the block's column band
        float* sl = p.slab + (long long)bz * p.Cout * p.P;
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            const int opix = n0 + wn * (BN / WN) + j * TS + l31;
            if (opix >= W_P) continue;
            const int ob = opix / OHW;
            const int orr = opix - ob * OHW, ooh = orr / w_own;
            const int gpix = ob * OHOW + ooh * p.OW + w_ow0 + (orr - ooh * w_own);
#pragma unroll
            for (int i = 0; i < TMI; ++i)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const int co = m0 + wm * (BM / WM) + i * TS + acc_row(r);
                    float v = acc[i][j][r];
                    if constexpr (LO) v = fmaf(acc_lo[i][j][r], 0x1p-11f, v);
                    if (co < p.Cout) sl[(long long)co * p.P + gpix] = AR >= 1 ? v * out_scale : v;
                }
        }
        return;
    }
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < TNI; ++j) {
        const int opix = n0 + wn * (BN / WN) + j * TS + l31;
        if (opix >= W_P) continue;
        const int ob = opix / OHW;
        const int orr = opix - ob * OHW, ooh = orr / w_own;
        const int orp = ooh * p.OW + w_ow0 + (orr - ooh * w_own);
        float* yb = p.y + (long long)ob * p.y_bs + orp;
        const float* rbp = p.res ? p.res + (long long)ob * p.res_bs + orp : nullptr;
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int co = m0 + wm * (BM / WM) + i * TS + acc_row(r);
                if (co >= p.Cout) continue;
                float v = acc[i][j][r];
                if constexpr (LO) v = fmaf(acc_lo[i][j][r], 0x1p-11f, v);
                if constexpr (AR >= 1) v *= out_scale;
                if (p.scale) v *= p.scale[co];
                if (p.shift) v += p.shift[co];
                const long long off = (long long)co * OHOW;
                if (rbp) v += rbp[off];
                if (p.accumulate) v += acc_prev(p, yb + off);
                if (p.relu) v = fmaxf(v, 0.f);
                yb[off] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    if (p.y_amax) publish_amax(vmax, p.y_amax);
}

template <int BM, int BN, int WM, int BK, int NT = kThreads, int AR = 1, bool MF = false, bool MS = false, bool IL = false>
__global__ __launch_bounds__(NT, NT == kThreads ? 2 : 1) void conv_igemm_split_kernel(ConvP p) {
    conv_igemm_split_body<BM, BN, WM, BK, NT, AR, MF, MS, IL>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// Several convolutions in ONE launch: workgroups [start[g], start[g + 1]) of the 1-D grid work on problem g, as its
// (gx[g], gy[g], K slices) grid would have.  ASPP's branches read the same 134 MB input and execute 1 .. 9 taps per tile
// (padding taps are skipped): launched one after the other each ends with CUs idling behind its longest tiles
// (0.28-0.31 of the MFMA peak against 0.40-0.45 for the balanced layer4 shapes); strung together, heaviest problem first,
// the dispatcher keeps every CU busy until the common end.
struct ConvGroup {
    int n;
    int start[5];
    int gx[4], gy[4];
    // interleaved form (ns > 0): workgroup L works on stream L % ns - a (problem, K slice) pair - and on tile L / ns of it.  With
    // ns = 8 (the XCDs: workgroup L runs on XCD L % 8) every XCD runs ONE stream: its 32 workgroups at a time read the same
    // weights through the same L2.  In problem-major order workgroups of many streams share each L2, nothing is reused and the
    // launch fetches 3.9 GB past L2 for 0.26 GB of operands (profiles/r05_notes.md).
    int ns;
    int s_prob[8], s_slice[8];
    ConvP p[4];
};
template <int BM, int BN, int WM, int BK, int NT = kThreads, int AR = 1, bool MF = false, bool IL = false>
__global__ __launch_bounds__(NT, NT == kThreads ? 2 : 1) void conv_igemm_split_group_kernel(ConvGroup grp) {
    const int L = blockIdx.x;
    if (grp.ns > 0) {
        const int s = L % grp.ns, r = L / grp.ns;
        const int g = grp.s_prob[s], gx = grp.gx[g], gy = grp.gy[g];
        conv_igemm_split_body<BM, BN, WM, BK, NT, AR, MF, false, IL>(grp.p[g], r % gx, r / gx, grp.s_slice[s], gx, gy);
        return;
    }
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < grp.n && L >= grp.start[i]) g = i;
    const int local = L - grp.start[g], gx = grp.gx[g], gy = grp.gy[g];
    const int plane = gx * gy, bz = local / plane, r = local - bz * plane;
    conv_igemm_split_body<BM, BN, WM, BK, NT, AR, MF, false, IL>(grp.p[g], r % gx, r / gx, bz, gx, gy);
}

// ---------------------------------------------------------------------------------------------
// max|x| of a tensor of B images of `per` contiguous floats (batch stride x_bs), into a zero-initialised device scalar
__global__ void amax_kernel(const float* __restrict__ x, long long per, long long x_bs, long long total,
                            float* __restrict__ out) {
    float m = 0.f;
    if (x_bs == per && (per & 3) == 0 && (reinterpret_cast<unsigned long long>(x) & 15) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total / 4; i += (long long)gridDim.x * blockDim.x) {
            const float4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const long long b = i / per;
            m = fmaxf(m, fabsf(x[b * x_bs + (i - b * per)]));
        }
    }
    publish_amax(m, out);
}

// max|.| of n tensors in one launch (every conv weight of the model after the optimiser step): blockIdx.y = tensor,
// 16-byte loads where the tensor allows them
__global__ void multi_amax_kernel(const float* const* __restrict__ ptrs, const long long* __restrict__ counts,
                                  float* __restrict__ out) {
    const float* x = ptrs[blockIdx.y];
    const long long n = counts[blockIdx.y];
    float m = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<unsigned long long>(x) & 15) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n / 4; i += (long long)gridDim.x * blockDim.x)
            m = amax4(m, x4[i]);
    } else {
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
            m = fmaxf(m, fabsf(x[i]));
    }
    publish_amax(m, out + blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// Weight layout for the split kernels: w[co][ci][tap] ->
//   fwd  : [ (tap*Cin + ci) / 16 ][ co ][ piece ][ ci % 16 ]      (rows = co,  Cin  % 16 == 0)
//   dgrad: [ (tap*Cout + co) / 16 ][ ci ][ piece ][ co % 16 ]     (rows = ci,  Cout % 16 == 0)
// One thread converts 8 consecutive k of one row and writes NP 16-byte runs.  AR = 1: `amax` (device scalar, the
// tensor's max|w|, already reduced) gives the scale; it is also copied into both layouts' trailers.
template <int AR>
__device__ __forceinline__ void prep_weights_split_body(const float* __restrict__ w, unsigned char* __restrict__ fwd,
                                                        unsigned char* __restrict__ dg, int Cout, int Cin, int T,
                                                        const float* __restrict__ amax, int bx, int by) {
    constexpr int NP = SplitArith<AR>::NP;
    constexpr int K16B = split_k16_bytes(AR);
    // a block stages w[co0..co0+31][ci0..ci0+31][all taps] like prep_weights_tiled_kernel (T <= 9)
    constexpr int LDT = 32 * 9 + 1;
    __shared__ float tile[32 * LDT];
    const int ci0 = bx * 32, co0 = by * 32;
    const int tid = threadIdx.x;
    const int nci = min(32, Cin - ci0), nco = min(32, Cout - co0);
    const int run = nci * T;
    float ws = 1.f;
    if constexpr (AR >= 1) {
        int e;
        ws = pow2_scale(*amax, e);
        if (bx == 0 && by == 0 && tid == 0) {
            const long long body = split_layout_bytes(AR, (long long)T * Cin, Cout) - 16;     // same for both layouts
            if (fwd) *reinterpret_cast<float*>(fwd + body) = *amax;
            if (dg) *reinterpret_cast<float*>(dg + body) = *amax;
        }
    }
    for (int r = tid >> 5; r < nco; r += 8) {
        const float* src = w + ((long long)(co0 + r) * Cin + ci0) * T;
        for (int j = tid & 31; j < run; j += 32) tile[r * LDT + j] = src[j] * ws;
    }
    __syncthreads();
    auto emit = [&](unsigned char* dst, const float (&v)[8]) {
        unsigned pc[NP][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (AR == 0)
                split3(v[2 * e], v[2 * e + 1], pc[0][e], pc[1][e], pc[2][e]);
            else if constexpr (AR == 1)
                split2h(v[2 * e], v[2 * e + 1], pc[0][e], pc[1][e]);
            else
                split2hs(v[2 * e], v[2 * e + 1], pc[0][e], pc[1][e]);
        }
#pragma unroll
        for (int c = 0; c < NP; ++c) *reinterpret_cast<u32x4*>(dst + c * 32) = u32x4{pc[c][0], pc[c][1], pc[c][2], pc[c][3]};
    };
    if (fwd) {
        // items: (tap, g = 8-run of ci, co) with co fastest
        const int ng = nci / 8, items = T * ng * nco;
        for (int it = tid; it < items; it += blockDim.x) {
            const int col = it % nco, rest = it / nco, g = rest % ng, tap = rest / ng;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[col * LDT + (g * 8 + e) * T + tap];
            const int k = tap * Cin + ci0 + g * 8;
            emit(fwd + ((long long)(k / 16) * Cout + co0 + col) * K16B + (k % 16) * 2, v);
        }
    }
    if (dg) {
        const int ng = nco / 8, items = T * ng * nci;
        for (int it = tid; it < items; it += blockDim.x) {
            const int cil = it % nci, rest = it / nci, g = rest % ng, tap = rest / ng;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[(g * 8 + e) * LDT + cil * T + tap];
            const int k = tap * Cout + co0 + g * 8;
            emit(dg + ((long long)(k / 16) * Cin + ci0 + cil) * K16B + (k % 16) * 2, v);
        }
    }
}

template <int AR>
__global__ void prep_weights_split_kernel(const float* __restrict__ w, unsigned char* __restrict__ fwd,
                                          unsigned char* __restrict__ dg, int Cout, int Cin, int T,
                                          const float* __restrict__ amax) {
    prep_weights_split_body<AR>(w, fwd, dg, Cout, Cin, T, amax, blockIdx.x, blockIdx.y);
}

// All the split layouts of a model in ONE launch (the per-step re-layout after the optimiser step was 66 launches of 7 us on
// the side stream, beside the first layers of the next forward).  Workgroup b belongs to the last entry whose block_begin
// is <= b; the table lives on the device and is built once per set of buffers.
template <int AR>
__global__ void prep_weights_split_multi_kernel(const wsdl_prep_desc* __restrict__ d, int n) {
    const int b = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].block_begin <= b) lo = mid; else hi = mid - 1;
    }
    const wsdl_prep_desc q = d[lo];
    const int lb = b - q.block_begin;
    prep_weights_split_body<AR>(q.w, static_cast<unsigned char*>(q.wt_fwd), static_cast<unsigned char*>(q.wt_dgrad), q.Cout, q.Cin,
                                q.taps, q.w_amax, lb % q.grid_x, lb / q.grid_x);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient on the 16-bit matrix core: D[cout][n] = sum_pixel dY[cout][pixel] * Xg[n][pixel], 32-pixel chunks
// (Cout % 128 == 0, Cin % 128 == 0, at least 6 N tiles; everything else stays on the fp32 kernels of conv_igemm.hip).
//
// A first version kept conv_wgrad_fast_kernel's 16-pixel chunks and split both operands in the kernel: no faster than
// fp32.  What bounds it is the vector-memory pipe, not the matrix core: a 16-pixel chunk reads 64-byte half lines
// (the other half is fetched again by the next chunk after the 32 KiB L1 has turned over), and both operands pay the
// VALU split.  Here
//   * a chunk is 32 consecutive pixels = one full 128-byte line per row, read by 32 adjacent lanes;
//   * dY is split ONCE per launch by dy_split_kernel into the kernel's own LDS row format
//     ([pixel/32][cout][2 x NP pieces x 16 + 16 B pad]): a row tile of a chunk is one contiguous run that every N tile
//     of the launch (36 for a 3x3 on 512 channels) copies with 16-byte units - no arithmetic, no bank conflicts;
//   * x is split in the kernel.  A lane holds ONE pixel of a row, the packed converts want a pixel PAIR per
//     lane: lanes swap one value with their neighbour (DPP quad_perm) so that even lanes own the pair of one
//     row and odd lanes the pair of the row below - 4 VALU ops per pair instead of a second, half-used load.
constexpr int w2row_bytes(int AR) { return 2 * split_k16_bytes(AR) + 16; }     // 32 pixels x NP pieces + padding

template <int AR>
__global__ void dy_split_kernel(const float* __restrict__ dy, unsigned char* __restrict__ out, int B, int Cout,
                                int OHOW, long long dy_bs, int P, const float* __restrict__ amax) {
    constexpr int NP = SplitArith<AR>::NP;
    constexpr int K16B = split_k16_bytes(AR), ROW = w2row_bytes(AR);
    float sc = 1.f;
    if constexpr (AR == 1) {
        int e;
        sc = pow2_scale(*amax, e);
    }
    // one thread: 16 consecutive pixels (linear index over b, oh, ow) of one channel; an even number of groups so that
    // every row is written in full (zeros past the last pixel)
    const int groups = 2 * ((P + 31) / 32);
    const long long total = (long long)groups * Cout;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups), c = (int)(idx / groups);
        float v[16];
        const int p0 = g * 16;
        if (p0 >= P) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
        } else if (OHOW % 16 == 0) {
            const int b = p0 / OHOW, r = p0 - b * OHOW;
            const float4* src = reinterpret_cast<const float4*>(dy + (long long)b * dy_bs + (long long)c * OHOW + r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 q = src[e];
                v[4 * e] = q.x; v[4 * e + 1] = q.y; v[4 * e + 2] = q.z; v[4 * e + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pix = p0 + e;
                float t = 0.f;
                if (pix < P) {
                    const int b = pix / OHOW, r = pix - b * OHOW;
                    t = dy[(long long)b * dy_bs + (long long)c * OHOW + r];
                }
                v[e] = t;
            }
        }
        unsigned pc[NP][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (AR == 0)
                split3(v[2 * e], v[2 * e + 1], pc[0][e], pc[1][e], pc[2][e]);
            else
                split2h(v[2 * e] * sc, v[2 * e + 1] * sc, pc[0][e], pc[1][e]);
        }
        u32x4* dst = reinterpret_cast<u32x4*>(out + ((long long)(g >> 1) * Cout + c) * ROW + (g & 1) * K16B);
#pragma unroll
        for (int c2 = 0; c2 < NP; ++c2) {
            dst[2 * c2] = u32x4{pc[c2][0], pc[c2][1], pc[c2][2], pc[c2][3]};
            dst[2 * c2 + 1] = u32x4{pc[c2][4], pc[c2][5], pc[c2][6], pc[c2][7]};
        }
    }
}

// 256 threads, 64x64 wave tiles, ONE LDS image per workgroup and two barriers per chunk, so two workgroups share a CU:
// while one converts / stores its next chunk, the other one's MFMAs use the matrix cores.  The two rows of a lane pair
// are neighbours (row stride an odd multiple of 16 bytes: disjoint banks).
// Variants measured and dropped (profiles/r01_notes.md): 512 threads with a double-buffered image (both waves of a
// SIMD sit in the same phase), producer / consumer waves with a 3-deep register ring (the pure consumer loop - one
// wave per SIMD, operand reads exposed after every barrier - already runs at half the MFMA rate).
// `dy_amax` / p.x_amax: the scales of the two operands (AR = 1); the slab receives acc / (s_dy * s_x).
//
// (An 8-pixel-run staging of x - two 16-byte loads and one ds_write_b128 per (row, piece) instead of 16 dword loads, 16 DPP
// exchanges and 32 ds_write_b32 - was an option in round 2: bit-identical, 2.5 % slower on the sweep, 8-13 % on the dilated
// ASPP shapes.  Removed in round 3.)
template <int BM, int BN, int AR>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_split32_kernel(WgradP p, const unsigned char* __restrict__ dys,
                                                                        unsigned dys_bytes, const float* __restrict__ dy_amax) {
    static_assert(BM == 128 && BN == 128, "tile");
    using Ar = SplitArith<AR>;
    using frag = typename Ar::frag;
    constexpr int NP = Ar::NP;
    constexpr int K16B = split_k16_bytes(AR);
    constexpr int BK = 32, ROW = w2row_bytes(AR);
    constexpr int WN = 2, MI = 2, NI = 2;
    constexpr int A_UNITS = BM * ROW / 16;                 // the dY image of a chunk is one contiguous run
    constexpr int A_U = (A_UNITS + kThreads - 1) / kThreads;   // per thread, the last one partial (LDS padded)
    constexpr int B_PER = BN / 8;                          // 16 rows per thread: 8 half-waves x 32 pixels per pass
    constexpr unsigned kOOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) unsigned char As[A_U * kThreads * 16];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware tile order (p.xcd_order): workgroups are dealt round-robin to the 8 XCDs in launch order, so in launch
    // order every XCD's L2 sees every (row tile, slab) of dY - 8 x the unique bytes measured at the fabric.  Re-labelled,
    // XCD c owns a contiguous eighth of the (slab, row tile, N tile) sequence: about one and a half (row tile, slab)
    // pairs of dY and the x rows under them.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int L = (bz * gy + by) * gx + bx, eighth = (gx * gy * (int)gridDim.z) >> 3;
        const int Lp = (L & 7) * eighth + (L >> 3);
        bz = Lp / (gx * gy);
        const int r = Lp - bz * (gx * gy);
        if (p.xcd_order == 2 && (gy & 1) == 0 && (p.Cin / BN & 1) == 0) {
            // blocks of (2 row tiles x 2 channel blocks x all taps): the taps of a channel block share x rows
            const int cb = p.Cin / BN, T = gx / cb, blk = 4 * T;
            const int b = r / blk, q = r - b * blk;
            const int mp = b / (cb / 2), cp = b - mp * (cb / 2);
            const int t = q >> 2, ml = (q >> 1) & 1, cl = q & 1;
            by = 2 * mp + ml;
            bx = t * cb + 2 * cp + cl;
        } else {
            by = r / gx;
            bx = r - by * gx;
        }
    }
    const int n0 = bx * BN, m0 = by * BM;
    const int zsplit = bz, w_cps = p.chunks_per_split;
    const int W_P = p.P;
    const int OHOW = p.OH * p.OW, HW = p.H * p.W;
    const int px = tid & 31, hw = tid >> 5;

    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(dys), 0, (int)dys_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    float xs = 1.f, out_scale = 1.f;
    if constexpr (AR == 1) {
        int ex, ed;
        xs = pow2_scale(*p.x_amax, ex);
        (void)pow2_scale(*dy_amax, ed);
        out_scale = pow2(-(ex + ed));
    }

    const int tap0 = n0 / p.Cin, ci0 = n0 - tap0 * p.Cin;
    const int t_dh = (tap0 / p.KW) * p.dil - p.pad, t_dw = (tap0 % p.KW) * p.dil - p.pad;
    // this thread's x rows: 2*hw + (e & 1) + 16*(e >> 1): the two rows of a pair are neighbours
    const unsigned b_row = (unsigned)((ci0 + 2 * hw) * HW);

    // dY arrives in the LDS row format itself: unit u of the row tile goes to byte 16 u of the image - consecutive
    // lanes, consecutive 16 bytes on both sides, no bank conflicts
    unsigned voff_a[A_U];
#pragma unroll
    for (int e = 0; e < A_U; ++e) {
        const int u = tid + e * kThreads;
        voff_a[e] = u < A_UNITS ? (unsigned)(m0 * ROW + u * 16) : kOOB;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[A_U];
    unsigned rb[B_PER];
    const int chunk_begin = zsplit * w_cps;
    const int total_chunks = (W_P + BK - 1) / BK;
    const int chunk_end = min(chunk_begin + w_cps, total_chunks);

    const bool row_chunks = (p.OW % BK) == 0;
    unsigned nvb = kOOB;
    auto decode = [&](int c) {
        nvb = kOOB;
        if (row_chunks) {
            const int first = c * BK;
            const int grow = first / p.OW, ow = first - grow * p.OW + px;
            const int pb = grow / p.OH, oh = grow - pb * p.OH;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                nvb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W) + (unsigned)iw + b_row) * 4u;
            return;
        }
        const int pix = c * BK + px;
        if (pix < W_P) {
            const int pb = pix / OHOW, rr = pix - pb * OHOW;
            const int oh = rr / p.OW, ow = rr - oh * p.OW;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                nvb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W + iw) + b_row) * 4u;
        }
    };
    auto next_valid = [&](int c) {
        for (; c < chunk_end; ++c) {
            decode(c);
            if (__any(nvb != kOOB)) break;
        }
        return c;
    };
    auto load_tiles = [&](int c) {
        const unsigned soff_a = (unsigned)c * (unsigned)(p.Cout * ROW);
#pragma unroll
        for (int e = 0; e < A_U; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rdy, voff_a[e], soff_a, 0);
#pragma unroll
        for (int e = 0; e < B_PER; ++e)
            rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, nvb, (unsigned)(((e & 1) + 16 * (e >> 1)) * HW) * 4u, 0);
    };
    const bool even = (px & 1) == 0;
    const int pair = px >> 1;
    const unsigned st_b = (unsigned)((2 * hw + (even ? 0 : 1)) * ROW + (pair >> 3) * K16B + (pair & 7) * 4);
    auto store_tiles = [&]() {
#pragma unroll
        for (int e = 0; e < A_U; ++e) *reinterpret_cast<u32x4*>(As + (tid + e * kThreads) * 16) = ra[e];
#pragma unroll
        for (int i = 0; i < B_PER / 2; ++i) {
            const unsigned give = even ? rb[2 * i + 1] : rb[2 * i];
            const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);
            const unsigned x0 = even ? rb[2 * i] : recv;
            const unsigned x1 = even ? recv : rb[2 * i + 1];
            unsigned pc[NP];
            if constexpr (AR == 0)
                split3(__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1), pc[0], pc[1], pc[2]);
            else
                split2h(__builtin_bit_cast(float, x0) * xs, __builtin_bit_cast(float, x1) * xs, pc[0], pc[1]);
            unsigned char* d = Bs + st_b + i * 16 * ROW;
#pragma unroll
            for (int c = 0; c < NP; ++c) *reinterpret_cast<unsigned*>(d + c * 32) = pc[c];
        }
    };

    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned char* Ab = As + (wm * 64 + l31) * ROW + lh * 16;
    const unsigned char* Bb = Bs + (wn * 64 + l31) * ROW + lh * 16;
    // a tap that reads padding for EVERY output pixel (dilation >= map size: ASPP d36 on 32x32 maps keeps only the
    // centre tap) contributes exact zeros: its workgroups leave at once
    // (the host leaves those slab columns out of the reduce: live_taps() in conv_igemm.hip applies the same test)
    if (t_dh >= p.H || (p.OH - 1) * p.stride + t_dh < 0 || t_dw >= p.W || (p.OW - 1) * p.stride + t_dw < 0) return;
    int c0 = next_valid(chunk_begin);
    if (c0 < chunk_end) load_tiles(c0);
    while (c0 < chunk_end) {
        store_tiles();                                   // chunk c0: registers -> 16-bit pieces -> LDS
        const int c1 = next_valid(c0 + 1);
        if (c1 < chunk_end) load_tiles(c1);              // in flight during this chunk's MFMAs
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag a[MI][NP], b[NI][NP];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int c = 0; c < NP; ++c)
                    a[i][c] = *reinterpret_cast<const frag*>(Ab + i * 32 * ROW + ks * K16B + c * 32);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int c = 0; c < NP; ++c)
                    b[j][c] = *reinterpret_cast<const frag*>(Bb + j * 32 * ROW + ks * K16B + c * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = split_products<AR>(a[i], b[j], acc[i][j]);
        }
        lds_barrier();
        c0 = c1;
    }

    float* slab = p.slab + (long long)(p.slab0 + bz) * p.Cout * p.N;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)co * p.N + n] = AR == 1 ? acc[i][j][r] * out_scale : acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient on v_mfma_f32_16x16x32_f16 (AR = 1 only): one MFMA covers the whole 32-pixel chunk.  Under
// the power limit the chip sustains a higher clock on this shape than on 32x32x16 for the same FLOPs (MI355X_MICROARCH
// "DVFS give-back" item 7; a timing-only build of the kernel above with its MFMAs swapped showed +10 %).
// LDS rows are 128 bytes without padding - [piece c][k-group g of 8 pixels] = unit u = 4 c + g of 16 bytes, stored at
// unit (u ^ ((row >> 1) & 7)): the ds_read_b128 of a 16x16x32 operand (lane l: row l & 15, k-group l >> 4) then takes 16
// distinct 16-byte slots of the 256-byte bank row in every one of its four lane groups (checked slot by slot against
// the b128 lane groups of the LDS table); the plain padded rows of the 32x32x16 kernel are 2-way conflicted for it.
// dY is pre-split into the same rows (dy_split16_kernel), x is staged as in the kernel above.
constexpr int kW16Row = 128;

__global__ void dy_split16_kernel(const float* __restrict__ dy, unsigned char* __restrict__ out, int B, int Cout,
                                  int OHOW, long long dy_bs, int P, const float* __restrict__ amax, int amax_stride) {
    // amax_stride 0: one scale for the tensor; 1: amax[c] per channel (wsdl_set_option "wgrad_chan_scale")
    const int groups = 2 * ((P + 31) / 32);
    const long long total = (long long)groups * Cout;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups), c = (int)(idx / groups);
        int ex;
        const float sc = pow2_scale(amax[c * amax_stride], ex);
        float v[16];
        const int p0 = g * 16;
        if (p0 >= P) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
        } else if (OHOW % 16 == 0) {
            const int b = p0 / OHOW, r = p0 - b * OHOW;
            const float4* src = reinterpret_cast<const float4*>(dy + (long long)b * dy_bs + (long long)c * OHOW + r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 q = src[e];
                v[4 * e] = q.x; v[4 * e + 1] = q.y; v[4 * e + 2] = q.z; v[4 * e + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pix = p0 + e;
                float t = 0.f;
                if (pix < P) {
                    const int b = pix / OHOW, r = pix - b * OHOW;
                    t = dy[(long long)b * dy_bs + (long long)c * OHOW + r];
                }
                v[e] = t;
            }
        }
        unsigned pc[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2h(v[2 * e] * sc, v[2 * e + 1] * sc, pc[0][e], pc[1][e]);
        unsigned char* row = out + ((long long)(g >> 1) * Cout + c) * kW16Row;
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int u = 4 * c2 + 2 * (g & 1) + h;
                *reinterpret_cast<u32x4*>(row + ((u ^ sw) << 4)) =
                    u32x4{pc[c2][4 * h], pc[c2][4 * h + 1], pc[c2][4 * h + 2], pc[c2][4 * h + 3]};
            }
    }
}

// CS (per-channel scales, the range guard of the weight gradient): p.x_amax / dy_amax are read at [channel * p.xa_stride] /
// [channel * p.da_stride] - stride 1: ARRAYS with one maximum per input / per output channel (published by the channel-resident
// BatchNorm kernels that wrote the tensors, or taken by a pre-pass under "wgrad_chan_scale"), stride 0: one maximum for the tensor.  A channel is a row or a column of this GEMM's OUTPUT (K = pixels), so a scale of its own per channel is
// a power of two on the operand's rows and the inverse on the result's rows / columns: exact, and no second accumulator set.
template <int BM, int BN, bool CS = false>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_split16_kernel(WgradP p, const unsigned char* __restrict__ dys,
                                                                        unsigned dys_bytes, const float* __restrict__ dy_amax) {
    static_assert(BM == 128 && BN == 128, "tile");
    constexpr int BK = 32, ROW = kW16Row;
    constexpr int WN = 2;
    constexpr int TMI = BM / 32, TNI = BN / 32;             // 16 x 16 tiles of a wave (2 x 2 waves)
    constexpr int A_U = BM * ROW / 16 / kThreads;          // units per thread, exact
    constexpr int B_PER = BN / 8;
    constexpr unsigned kOOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) unsigned char As[BM * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int L = (bz * gy + by) * gx + bx, eighth = (gx * gy * (int)gridDim.z) >> 3;
        const int Lp = (L & 7) * eighth + (L >> 3);
        bz = Lp / (gx * gy);
        const int r = Lp - bz * (gx * gy);
        by = r / gx;
        bx = r - by * gx;
    }
    const int n0 = bx * BN, m0 = by * BM;
    const int zsplit = bz, w_cps = p.chunks_per_split;
    const int W_P = p.P;
    const int OHOW = p.OH * p.OW, HW = p.H * p.W;
    const int px = tid & 31, hw = tid >> 5;

    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(dys), 0, (int)dys_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    int ex = 0, ed = 0;
    float xs = 1.f;
    if constexpr (!CS) {
        xs = pow2_scale(*p.x_amax, ex);
        (void)pow2_scale(*dy_amax, ed);
    }
    const float out_scale = pow2(-(ex + ed));

    const int tap0 = n0 / p.Cin, ci0 = n0 - tap0 * p.Cin;
    const int t_dh = (tap0 / p.KW) * p.dil - p.pad, t_dw = (tap0 % p.KW) * p.dil - p.pad;
    const unsigned b_row = (unsigned)((ci0 + 2 * hw) * HW);
    // CS: the scales of the tile's 128 rows of x in LDS (eight registers per thread for them spilled)
    __shared__ float xst[CS ? BN : 1];
    __shared__ int edy_s[CS ? BM : 1];                 // CS: exponent of the scale of each of the tile's rows of dY (the epilogue's)
    if constexpr (CS) {
        if (tid < BN) {
            int e_;
            xst[tid] = pow2_scale(p.x_amax[(ci0 + tid) * p.xa_stride], e_);
            (void)pow2_scale(dy_amax[(m0 + tid) * p.da_stride], e_);        // (BM == BN)
            edy_s[tid] = e_;
        }
        __syncthreads();
    }
    const float* xsr = xst + 2 * hw + (px & 1);

    unsigned voff_a[A_U];
#pragma unroll
    for (int e = 0; e < A_U; ++e) voff_a[e] = (unsigned)(m0 * ROW + (tid + e * kThreads) * 16);

    f32x4 acc[TMI][TNI];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // register ring of TWO chunks (round 3: the loads of chunk k + 2 are issued while chunk k is multiplied - one chunk ahead
    // left load time exposed, see conv_wgrad_split16d_kernel)
    u32x4 ra[2][A_U];
    unsigned rb[2][B_PER];
    const int chunk_begin = zsplit * w_cps;
    const int total_chunks = (W_P + BK - 1) / BK;
    const int chunk_end = min(chunk_begin + w_cps, total_chunks);

    const bool row_chunks = (p.OW % BK) == 0;
    unsigned nvb = kOOB;
    auto decode = [&](int c) {
        nvb = kOOB;
        if (row_chunks) {
            const int first = c * BK;
            const int grow = first / p.OW, ow = first - grow * p.OW + px;
            const int pb = grow / p.OH, oh = grow - pb * p.OH;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                nvb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W) + (unsigned)iw + b_row) * 4u;
            return;
        }
        const int pix = c * BK + px;
        if (pix < W_P) {
            const int pb = pix / OHOW, rr = pix - pb * OHOW;
            const int oh = rr / p.OW, ow = rr - oh * p.OW;
            const int ih = oh * p.stride + t_dh, iw = ow * p.stride + t_dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                nvb = ((unsigned)((long long)pb * p.x_bs) + (unsigned)(ih * p.W + iw) + b_row) * 4u;
        }
    };
    auto next_valid = [&](int c) {
        for (; c < chunk_end; ++c) {
            decode(c);
            if (__any(nvb != kOOB)) break;
        }
        return c;
    };
    auto load_tiles = [&](auto slot_c, int c) {      // decode(c) has just run (nvb)
        constexpr int S = decltype(slot_c)::value;
        const unsigned soff_a = (unsigned)c * (unsigned)(p.Cout * ROW);
#pragma unroll
        for (int e = 0; e < A_U; ++e) ra[S][e] = __builtin_amdgcn_raw_buffer_load_b128(rdy, voff_a[e], soff_a, 0);
#pragma unroll
        for (int e = 0; e < B_PER; ++e)
            rb[S][e] = __builtin_amdgcn_raw_buffer_load_b32(rx, nvb, (unsigned)(((e & 1) + 16 * (e >> 1)) * HW) * 4u, 0);
    };
    const bool even = (px & 1) == 0;
    const int pair = px >> 1;
    // narrow form: rows 2 hw + (odd lane) + 16 i share (row >> 1) & 7 = hw & 7
    const unsigned st_row = (unsigned)((2 * hw + (even ? 0 : 1)) * ROW + (pair & 3) * 4);
    const unsigned st_u0 = (unsigned)((((pair >> 2)) ^ (hw & 7)) << 4), st_u1 = (unsigned)(((4 + (pair >> 2)) ^ (hw & 7)) << 4);
    auto store_tiles = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int e = 0; e < A_U; ++e) *reinterpret_cast<u32x4*>(As + (tid + e * kThreads) * 16) = ra[S][e];
#pragma unroll
        for (int i = 0; i < B_PER / 2; ++i) {
            const unsigned give = even ? rb[S][2 * i + 1] : rb[S][2 * i];
            const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);
            const unsigned x0 = even ? rb[S][2 * i] : recv;
            const unsigned x1 = even ? recv : rb[S][2 * i + 1];
            unsigned ph, pl;
#ifdef WSDL_EXP_W_NOSPLIT        // timing-only build: x as if it arrived pre-split
            ph = x0;
            pl = x1;
#else
            const float xsc = CS ? xsr[CS ? 16 * i : 0] : xs;
            split2h(__builtin_bit_cast(float, x0) * xsc, __builtin_bit_cast(float, x1) * xsc, ph, pl);
#endif
            unsigned char* d = Bs + st_row + i * 16 * ROW;
            *reinterpret_cast<unsigned*>(d + st_u0) = ph;
            *reinterpret_cast<unsigned*>(d + st_u1) = pl;
        }
    };

    const int l15 = lane & 15, lg = lane >> 4;
    const unsigned fr0 = (unsigned)(l15 * ROW + ((lg ^ (l15 >> 1)) << 4)), fr1 = (unsigned)(l15 * ROW + (((4 + lg) ^ (l15 >> 1)) << 4));
    const unsigned char* Ab = As + wm * (BM / 2) * ROW;
    const unsigned char* Bb = Bs + wn * (BN / 2) * ROW;
    if (t_dh >= p.H || (p.OH - 1) * p.stride + t_dh < 0 || t_dw >= p.W || (p.OW - 1) * p.stride + t_dw < 0) return;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int c0 = next_valid(chunk_begin), c1 = chunk_end;
    if (c0 < chunk_end) {
        load_tiles(S0{}, c0);
        c1 = next_valid(c0 + 1);
        if (c1 < chunk_end) load_tiles(S1{}, c1);
    }
#ifdef WSDL_EXP_W_NOSTAGE            // timing-only build: the LDS image filled once (real data), the loop is barrier + ds_read + MFMA
    if (c0 < chunk_end) store_tiles(S0{});
#endif
    // one chunk: its operands sit in register slot S, the next chunk's in slot 1 - S
    auto step = [&](auto slot_c) {
        int c2 = chunk_end;
#ifndef WSDL_EXP_W_NOSTAGE
        store_tiles(slot_c);
        if (c1 < chunk_end) c2 = next_valid(c1 + 1);
        if (c2 < chunk_end) load_tiles(slot_c, c2);            // the slot is free again; in flight during two chunks' MFMAs
#else
        if (c1 < chunk_end) c2 = c1 + 1;
#endif
        lds_barrier();
        half8 a[TMI][2], b[TNI][2];
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
            a[i][0] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr0);
            a[i][1] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr1);
        }
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            b[j][0] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + fr0);
            b[j][1] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + fr1);
        }
#ifdef WSDL_EXP_W_NOMFMA             // timing-only build: staging, barriers and the fragment reads, no matrix work
#pragma unroll
        for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int j = 0; j < TNI; ++j) acc[i][j][0] += (float)a[i][0][0] + (float)a[i][1][1] + (float)b[j][0][2] + (float)b[j][1][3];
#else
#pragma unroll
        for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int j = 0; j < TNI; ++j) {
                f32x4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
#endif
        lds_barrier();
        c0 = c1;
        c1 = c2;
    };
    while (c0 < chunk_end) {
        step(S0{});
        if (c0 >= chunk_end) break;
        step(S1{});
    }

    float* slab = p.slab + (long long)(p.slab0 + bz) * p.Cout * p.N;
    if constexpr (CS) {
        int en[TNI];
#pragma unroll
        for (int j = 0; j < TNI; ++j) (void)pow2_scale(p.x_amax[(ci0 + wn * (BN / 2) + j * 16 + l15) * p.xa_stride], en[j]);
#pragma unroll
        for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / 2) + i * 16 + lg * 4 + r, co = m0 + row;
                const int edr = edy_s[row];
                // (one add and one v_ldexp_f32 per element: exact, and in range whatever the two exponents are; the per-element
                // pow2() products of round 5 were ~1000 instructions of this epilogue - 2-6 % of the kernel)
#pragma unroll
                for (int j = 0; j < TNI; ++j)
                    slab[(long long)co * p.N + n0 + wn * (BN / 2) + j * 16 + l15] = ldexpf(acc[i][j][r], -(en[j] + edr));
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm * (BM / 2) + i * 16 + lg * 4 + r;
                slab[(long long)co * p.N + n] = acc[i][j][r] * out_scale;
            }
        }
}

// The same GEMM with the x operand's fragments taken STRAIGHT FROM GLOBAL MEMORY (round 3).  Timing-only builds of the
// kernel above (profiles/r03_notes.md) showed it bound by its staging, not by the matrix cores: loads, the neighbour
// exchange, the split and the LDS stores of x alone take 79 % of its time, the matrix loop alone 69 %.  Both operands are
// K-major here (K = pixels, contiguous in NCHW), and a lane of v_mfma_f32_16x16x32_f16 holds 8 consecutive k of one row:
// with 32-pixel chunks inside one output row (OW % 32 == 0, stride 1) those are 32 consecutive bytes of x - two 16-byte
// loads per 16 x 32 tile, split in registers, no LDS, no lane exchange.  The four waves sit side by side along N (each:
// all 128 rows of dY x 32 columns), so no x fragment is loaded or split twice; dY (pre-split rows, plain copies) still
// goes through LDS, now double-buffered with ONE barrier per chunk.  Per chunk and thread: 8 vector loads instead of 20,
// 4 LDS stores instead of 20.  Elements whose input column lies outside the image are zeroed by selects (the loads
// themselves cannot fault: buffer descriptor).
// MODE 0 (checked by the host): every tap's column shift is a multiple of 4 elements - the aligned loop, 237 registers.  (A form
// for any shift - a third 16-byte load per tile and one loop instance per shift, 256 registers and spills - was 0-7 % slower than
// the LDS-staged kernel on the dilation-1 / 2 shapes it was for and was removed in round 4; the SH template plumbing of `run`
// is what is left of it.)
// DYRAW: dY is read as fp32 from the tensor itself (p.dy) and split by the staging threads on its way into LDS - no
// dy_split16_kernel pre-pass for the launch (it re-reads and re-writes dY once per layer; here every N tile's workgroup
// splits its 128 x 32 slice again: 8 splits per thread and chunk).
// CS: per-channel scales, as in conv_wgrad_split16_kernel.
template <int MODE, bool DYRAW = false, bool CS = false>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_split16d_kernel(WgradP p, const unsigned char* __restrict__ dys,
                                                                         unsigned dys_bytes, const float* __restrict__ dy_amax) {
    constexpr int BM = 128, BN = 128, BK = 32, ROW = kW16Row;
    constexpr int TMI = BM / 16, TNI = 2;
    constexpr int A_U = BM * ROW / 16 / kThreads;          // 16-byte units per thread, exact
    __shared__ __attribute__((aligned(16))) unsigned char As[2][BM * ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int L = (bz * gy + by) * gx + bx, eighth = (gx * gy * (int)gridDim.z) >> 3;
        const int Lp = (L & 7) * eighth + (L >> 3);
        bz = Lp / (gx * gy);
        const int r = Lp - bz * (gx * gy);
        by = r / gx;
        bx = r - by * gx;
    }
    const int n0 = bx * BN, m0 = by * BM;
    const int HW = p.H * p.W;
    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(dys), 0, (int)dys_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    int ex = 0, ed = 0;
    float xs = 1.f, ds = 1.f;
    if constexpr (!CS) {
        xs = pow2_scale(*p.x_amax, ex);
        ds = pow2_scale(*dy_amax, ed);
    }
    const float out_scale = pow2(-(ex + ed));
    // DYRAW: eight lanes per row of dY (16 bytes = 4 pixels each), rows t / 8 + 32 e: full 128-byte lines per load instruction
    const __amdgpu_buffer_rsrc_t rdr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const int OHOW_ = p.OH * p.OW;
    const int dr_row = tid >> 3, dr_q = tid & 7;                   // row within a group of 32, 4-pixel group of the chunk

    const int tap0 = n0 / p.Cin, ci0 = n0 - tap0 * p.Cin;
    const int t_dh = (tap0 / p.KW) * p.dil - p.pad, t_dw = (tap0 % p.KW) * p.dil - p.pad;
    if (t_dh >= p.H || (p.OH - 1) + t_dh < 0 || t_dw >= p.W || (p.OW - 1) + t_dw < 0) return;     // a dead tap (stride 1)
    const int l15 = lane & 15, lg = lane >> 4;
    // this lane's rows of x for its two tiles (channel ci0 + 32 wid + 16 j + l15), at its 8-pixel group
    unsigned vrow[TNI];
#pragma unroll
    for (int j = 0; j < TNI; ++j) vrow[j] = (unsigned)((ci0 + wid * 32 + j * 16 + l15) * HW + lg * 8);
    float xsr[CS ? TNI : 1], dsr[CS && DYRAW ? A_U : 1];       // CS: the scales of this lane's rows of x / of the rows of dY it stages
    int exr[CS ? TNI : 1];
    __shared__ int edy_s[CS ? BM : 1];                          // CS: exponent of the scale of each of the tile's rows of dY (the epilogue's)
    if constexpr (CS) {
        if (tid < BM) {
            int e_;
            (void)pow2_scale(dy_amax[(m0 + tid) * p.da_stride], e_);
            edy_s[tid] = e_;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TNI; ++j) xsr[j] = pow2_scale(p.x_amax[(ci0 + wid * 32 + j * 16 + l15) * p.xa_stride], exr[j]);
        if constexpr (DYRAW) {
#pragma unroll
            for (int e = 0; e < A_U; ++e) {
                int e_;
                dsr[e] = pow2_scale(dy_amax[(m0 + dr_row + 32 * e) * p.da_stride], e_);
            }
        }
    }

    unsigned voff_a[A_U];
#pragma unroll
    for (int e = 0; e < A_U; ++e) voff_a[e] = (unsigned)(m0 * ROW + (tid + e * kThreads) * 16);

    f32x4 acc[TMI][TNI];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunk_begin = bz * p.chunks_per_split;
    const int total_chunks = (p.P + BK - 1) / BK;
    const int chunk_end = min(chunk_begin + p.chunks_per_split, total_chunks);
    const int cpr = p.OW / BK;                      // chunks per output row

    // chunk c -> element offset of (row ih, column iw0) of image pb (may point in front of the row: masked), iw0, validity
    int c_iw0 = 0;
    long long c_base = 0, c_dy = 0;
    auto decode = [&](int c) {
        const int grow = c / cpr, ow0 = (c - grow * cpr) * BK;
        const int pb = grow / p.OH, oh = grow - pb * p.OH;
        const int ih = oh + t_dh;
        c_iw0 = ow0 + t_dw;
        c_base = (long long)pb * p.x_bs + (long long)ih * p.W + c_iw0;
        c_dy = (long long)pb * p.dy_bs + (long long)oh * p.OW + ow0;       // element offset of the chunk's first pixel in channel 0
        return ih >= 0 && ih < p.H && c_iw0 + BK > 0 && c_iw0 < p.W;
    };
    auto next_valid = [&](int c) {
        for (; c < chunk_end; ++c)
            if (decode(c)) break;
        return c;
    };
    // One instance of the main loop per misalignment (a generic lambda over a compile-time SH): with SH a run-time value the
    // element selects and the third load slowed EVERY shape by 12-35 %.
    auto run = [&](auto sh_c) {
        constexpr int SH = decltype(sh_c)::value;
        // Register ring of TWO chunks: the loads of chunk k + 2 are issued while chunk k is multiplied (one chunk ahead left the
        // loop at load + MFMA time one after the other - timing-only builds: 172 + 182 us alone, 300 together on l4.conv2).
        u32x4 ra[2][A_U];
        u32x4 rbx[2][TNI][3];
        int ld_iw0[2] = {0, 0};                         // iw0 of the chunk whose x values sit in rbx[slot]
        // 16-byte loads need 16-byte aligned addresses (the hardware drops the low address bits).  Rows, channels and chunks are
        // multiples of four elements apart (the host checks it), so the misalignment is the tap's column shift modulo 4 - the
        // same for every chunk of this workgroup: load from the aligned address below (three loads cover the eight elements)
        // and pick element e + SH.
        auto load_tiles = [&](auto slot_c, int c) {     // decode(c) has just run
            constexpr int S = decltype(slot_c)::value;
            if constexpr (DYRAW) {
                const unsigned db = (unsigned)((c_dy + (long long)(m0 + dr_row) * OHOW_ + dr_q * 4) * 4ll);
#pragma unroll
                for (int e = 0; e < A_U; ++e)
                    ra[S][e] = __builtin_amdgcn_raw_buffer_load_b128(rdr, db + (unsigned)(e * 32 * OHOW_) * 4u, 0, 0);
            } else {
                const unsigned soff_a = (unsigned)c * (unsigned)(p.Cout * ROW);
#pragma unroll
                for (int e = 0; e < A_U; ++e) ra[S][e] = __builtin_amdgcn_raw_buffer_load_b128(rdy, voff_a[e], soff_a, 0);
            }
            // a chunk that starts in front of the tensor's first row wraps to a huge offset: out of range, zeros
            const unsigned cb = (unsigned)(c_base * 4ll);
#pragma unroll
            for (int j = 0; j < TNI; ++j) {
                const unsigned vo = vrow[j] * 4u + cb - (unsigned)(SH * 4);
                rbx[S][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
                rbx[S][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo + 16u, 0, 0);
                if constexpr (SH != 0) rbx[S][j][2] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo + 32u, 0, 0);
            }
            ld_iw0[S] = c_iw0;
        };
        auto store_a = [&](auto slot_c, int buf) {
            constexpr int S = decltype(slot_c)::value;
            if constexpr (DYRAW) {
                // row r = dr_row + 32 e, pixels 4 dr_q .. + 3: unit dr_q / 2 of piece h (and 4 + that of piece l), its half dr_q & 1
#pragma unroll
                for (int e = 0; e < A_U; ++e) {
                    const int r = dr_row + 32 * e;
                    const unsigned u0 = ra[S][e][0], u1 = ra[S][e][1], u2 = ra[S][e][2], u3 = ra[S][e][3];
                    unsigned h0, l0, h1, l1;
                    const float dsc = CS ? dsr[CS ? e : 0] : ds;
                    split2h(__builtin_bit_cast(float, u0) * dsc, __builtin_bit_cast(float, u1) * dsc, h0, l0);
                    split2h(__builtin_bit_cast(float, u2) * dsc, __builtin_bit_cast(float, u3) * dsc, h1, l1);
                    const unsigned sw = (unsigned)((r >> 1) & 7);
                    unsigned char* row = As[buf] + r * ROW + (dr_q & 1) * 8;
                    *reinterpret_cast<u32x2*>(row + ((((unsigned)(dr_q >> 1)) ^ sw) << 4)) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(row + (((4u + (unsigned)(dr_q >> 1)) ^ sw) << 4)) = u32x2{l0, l1};
                }
            } else {
#pragma unroll
                for (int e = 0; e < A_U; ++e) *reinterpret_cast<u32x4*>(As[buf] + (tid + e * kThreads) * 16) = ra[S][e];
            }
        };
        const unsigned fr0 = (unsigned)(l15 * ROW + ((lg ^ (l15 >> 1)) << 4)), fr1 = (unsigned)(l15 * ROW + (((4 + lg) ^ (l15 >> 1)) << 4));
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;

        int c0 = next_valid(chunk_begin), c1 = chunk_end;
        int buf = 0;
        if (c0 < chunk_end) {
            load_tiles(S0{}, c0);
            c1 = next_valid(c0 + 1);
            if (c1 < chunk_end) load_tiles(S1{}, c1);
            store_a(S0{}, 0);
        }
        // one chunk: its operands sit in slot S (dY already in As[buf]), the next chunk's in slot 1 - S
        auto step = [&](auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
            // x of this chunk: registers -> masked, scaled, split fragments (before the loads below reuse the slot)
            half8 b[TNI][2];
            {
                const int first = ld_iw0[S] + lg * 8;              // input column of this lane's element 0
                const int lo = -first, hi = p.W - first;           // valid elements: lo <= e < hi
#pragma unroll
                for (int j = 0; j < TNI; ++j) {
                    // (through scalars: __builtin_bit_cast applied to a vector ELEMENT read element 0 for every e - hipcc 7.2)
                    unsigned d[12];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d[e] = rbx[S][j][0][e];
                        d[4 + e] = rbx[S][j][1][e];
                        d[8 + e] = SH != 0 ? rbx[S][j][2][e] : 0u;
                    }
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned u = d[e + SH];
                        v[e] = __builtin_bit_cast(float, u);
                    }
                    u32x4 ph, pl;
                    const float xsc = CS ? xsr[CS ? j : 0] : xs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = (2 * e >= lo && 2 * e < hi) ? v[2 * e] * xsc : 0.f;
                        const float x1 = (2 * e + 1 >= lo && 2 * e + 1 < hi) ? v[2 * e + 1] * xsc : 0.f;
                        unsigned h, l;
                        split2h(x0, x1, h, l);
                        ph[e] = h;
                        pl[e] = l;
                    }
                    b[j][0] = __builtin_bit_cast(half8, ph);
                    b[j][1] = __builtin_bit_cast(half8, pl);
                }
            }
            int c2 = chunk_end;
            if (c1 < chunk_end) c2 = next_valid(c1 + 1);
#ifndef WSDL_EXP_D_NOLOAD            // timing-only build: operands loaded once, the loop is convert + barrier + LDS reads + MFMA
            if (c2 < chunk_end) load_tiles(slot_c, c2);            // slot S is free: x converted, dY stored one step ago
#endif
            lds_barrier();                                         // As[buf] is complete; nobody still reads As[buf ^ 1]
            const unsigned char* Ab = As[buf];
#pragma unroll
            for (int i = 0; i < TMI; ++i) {
                const half8 a0 = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr0);
                const half8 a1 = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr1);
#pragma unroll
                for (int j = 0; j < TNI; ++j) {
                    f32x4 c = acc[i][j];
#ifdef WSDL_EXP_D_NOMFMA             // timing-only build: everything but the matrix work
                    c[0] += (float)a0[0] + (float)a1[1] + (float)b[j][0][2] + (float)b[j][1][3];
#else
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[j][0], c, 0, 0, 0);
#endif
                    acc[i][j] = c;
                }
            }
#ifndef WSDL_EXP_D_NOLOAD
            if (c1 < chunk_end) store_a(std::integral_constant<int, 1 - S>{}, buf ^ 1);
            buf ^= 1;
#endif
            c0 = c1;
            c1 = c2;
        };
        while (c0 < chunk_end) {
            step(S0{});
            if (c0 >= chunk_end) break;
            step(S1{});
        }
    };
    static_assert(MODE == 0, "only the aligned loop is instantiated (the misaligned-tap loops were removed in round 4)");
    run(std::integral_constant<int, 0>{});

    float* slab = p.slab + (long long)(p.slab0 + bz) * p.Cout * p.N;
    if constexpr (CS) {
#pragma unroll
        for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + lg * 4 + r, co = m0 + row;
                const int edr = edy_s[row];
#pragma unroll
                for (int j = 0; j < TNI; ++j)
                    slab[(long long)co * p.N + n0 + wid * 32 + j * 16 + l15] = ldexpf(acc[i][j][r], -(exr[j] + edr));
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            const int n = n0 + wid * 32 + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + i * 16 + lg * 4 + r;
                slab[(long long)co * p.N + n] = acc[i][j][r] * out_scale;
            }
        }
}
