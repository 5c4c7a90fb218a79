// conv_w4.h - the fp16x2 implicit-GEMM convolution with ONE wave per SIMD (round 3; included by conv_igemm.hip after
// conv_split.h, same ConvP, same weight layout, same arithmetic: three v_mfma_f32_16x16x32_f16 per fp32 product).
//
// Why a second kernel for the same tiles.  Timing-only builds of conv_igemm_split_kernel<256,128,4,16,512> on the
// 512 -> 512 3x3 of layer4 (profiles/r03_notes.md): the whole kernel 238 us; staging alone (loads, split, LDS stores,
// barriers, no MFMA) 142 us; MFMAs + fragment reads + barriers alone (both LDS images filled once) 181 us = 47 % of the
// 16-bit peak.  So the matrix loop itself - eight waves, 64 x 64 per wave, a barrier and a cold ds_read restart after
// every 12 MFMAs per wave, both waves of a SIMD in the same phase - runs at under half the pipe's rate before any
// staging is added, and the two halves overlap only partly.  This kernel is built the other way round
// (cdna_hip_programming.md section 5; MI355X_MICROARCH.md "one wave per SIMD" rows):
//   * 256 x 128 x 32 tile, FOUR waves (2 x 2), 128 x 64 per wave: 96 MFMAs per wave between two barriers, 24 fragment
//     reads for them (0.25 ds_read_b128 per MFMA against 0.67), accumulators in the 512-register budget one wave per
//     SIMD has (128 accumulator + ~100 fragment registers);
//   * every MFMA gap has issue room for ~5 other instructions: the activation split (VALU), its LDS stores and the
//     next chunk's global loads are placed behind the MFMAs of the current chunk by the scheduler - the one wave of a
//     SIMD is never waiting for a partner's phase;
//   * weights (pre-split by the layout kernel) are copied with 16-byte loads and ds_write_b128 (LDS-DMA, tried first,
//     feeds a CU only ~30 GB/s: the 32 KB of weights per chunk took longer than the chunk's MFMAs);
//   * LDS rows of 128 bytes, 16-byte unit u = 4 piece + (k / 8) stored at slot u ^ ((row >> 1) & 7): the operand read of
//     v_mfma_f32_16x16x32_f16 (lane l: row l & 15, k-group l >> 4) takes 16 distinct 16-byte slots of the 256-byte bank
//     row in each of the four ds_read_b128 lane groups (the pattern conv_wgrad_split16_kernel uses);
//   * one raw s_barrier per chunk; the global loads of chunk q + 2 stay in flight across it.
// Tap skipping, column bands, split-K slabs, the XCD-aware tile order and the epilogue are those of
// conv_igemm_split_kernel (bit-compatible slabs; the accumulation ORDER inside a chunk differs, so results agree to
// fp32 rounding, not bit for bit).
#pragma once

template <int AR>
__global__ __launch_bounds__(256, 1) void conv_igemm_w4_kernel(ConvP p) {
    static_assert(AR == 1, "fp16x2 arithmetic only");
    constexpr int BM = 256, BN = 128, BK = 32, NT = 256;
    constexpr int ROW = 128;                             // LDS row: 2 pieces x 32 k x 2 bytes
    constexpr int K16B = 64;                             // the layout's bytes per row and k16 slab (2 pieces x 16 x 2)
    constexpr int TMI = 8, TNI = 4;                      // 16 x 16 tiles of a wave (128 x 64)
    constexpr int A_REG = BM * ROW / 16 / NT;            // 16-byte weight units per thread and chunk (8)
    constexpr int B_PER = 16;                            // channels of one pixel per thread and chunk
    constexpr unsigned kOOB = 0x80000000u;

    // one LDS object (cdna_hip_programming.md section 5 item 4a): [buffer][A rows | B rows], then the tap table
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (BM + BN) * ROW + 64 * 4 + 10 * NT * 4];
    unsigned char* const As0 = lds;
    unsigned char* const Bs0 = lds + BM * ROW;
    constexpr int BUF = (BM + BN) * ROW;
    int* const vtaps = reinterpret_cast<int*>(lds + 2 * BUF);
    // per (valid tap, thread): byte offset of the thread's pixel under that tap (kOOB = padding): the K loop advances from
    // tap to tap with one LDS read instead of a branch around the gather arithmetic - its body stays ONE basic block,
    // which is what lets the scheduler place the staging instructions between the MFMAs
    unsigned* const tapoff = reinterpret_cast<unsigned*>(lds + 2 * BUF + 64 * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_py > 0) {
        const int gx = gridDim.x;
        const int L = by * gx + bx;
        const int xcd = L & 7, idx = L >> 3;
        const int py = p.xcd_py, px = 8 / py;
        const int lx = gx / px, ly = (int)gridDim.y / py;
        bx = (xcd / py) * lx + idx % lx;
        by = (xcd % py) * ly + idx / lx;
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
    const int n0 = (bx - w_tile0) * BN;
    const int OHOW = p.OH * p.OW;
    const int HW = p.H * p.W;

    const int wbytes = (p.K / 16) * p.Cout * K16B;       // the layout without its trailer
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    int ex, ew;
    const float xs = pow2_scale(*p.x_amax, ex);
    (void)pow2_scale(*reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.wt) + wbytes), ew);
    const float out_scale = pow2(-(ex + ew));

    // activations: thread = (pixel pl, half kr of the chunk's 32 channels)
    const int pl = tid & (BN - 1), kr = tid >> 7;
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
    const unsigned img_off = (unsigned)((long long)pb * p.x_bs) + (unsigned)(kr * B_PER * HW);   // elements

    auto tap_src = [&](int t, int& sp) {
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

    const int T = p.KH * p.KW;                            // <= 9 (split_eligible)
    const int cpt = p.Cin / BK;
    int nv = 0;
    for (int t = 0; t < T; ++t) {
        int sp;
        const bool ok = tap_src(t, sp);
        if (__syncthreads_or(ok)) {
            if (tid == 0) vtaps[nv] = t;
            tapoff[nv * NT + tid] = ok ? (img_off + (unsigned)sp) * 4u : kOOB;
            ++nv;
        }
    }
    const int nq_all = nv * cpt;
    const int q0 = p.ksplit > 1 ? (int)((long long)nq_all * blockIdx.z / p.ksplit) : 0;
    const int q1 = p.ksplit > 1 ? (int)((long long)nq_all * (blockIdx.z + 1) / p.ksplit) : nq_all;
    const int nq = q1 - q0;
    __syncthreads();

    // ---- weights: register staged (16-byte loads of the pre-split layout, ds_write_b128 into the swizzled rows).  LDS-DMA was
    // built first and measured: a CU takes in ~30 GB/s that way (32 KB of weights per chunk = 1.06 us, longer than the chunk's
    // MFMAs), plain 16-byte loads from L2 more than twice that (profiles/r03_notes.md).
    // Unit u of the chunk's A image = (row r, 16-byte unit v of its 128 bytes): thread t handles units t + 256 e, e < 8 ->
    // row = (t + 256 e) / 8 = t / 8 + 32 e, v = t & 7 = 4 piece + kg: source = slab (kg >> 1), row's 64 bytes, part
    // 2 piece + (kg & 1); destination slot v ^ ((row >> 1) & 7) (8 consecutive lanes: one row's 8 units -> 8 slots of one
    // 128-byte row: conflict-free ds_write_b128 groups).
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, wbytes, 0x00020000);
    unsigned voff_w[A_REG], lds_w[A_REG];
#pragma unroll
    for (int e = 0; e < A_REG; ++e) {
        const int r = (tid >> 3) + 32 * e, v = tid & 7;
        const int piece = v >> 2, kg = v & 3;
        int grow = m0 + r;
        grow = grow < p.Cout ? grow : p.Cout - 1;        // rows past Cout: any valid bytes (their outputs are never stored)
        voff_w[e] = (unsigned)((kg >> 1) * p.Cout * K16B + grow * K16B + (2 * piece + (kg & 1)) * 16);
        lds_w[e] = (unsigned)(r * ROW + ((v ^ ((r >> 1) & 7)) << 4));
    }

    f32x4 acc[TMI][TNI];
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // cursors (wave-uniform, advanced without branches): activation loads run two chunks ahead, the weight DMA one
    unsigned rb[B_PER];
    int ld_vi = q0 / cpt, ld_c = q0 - (q0 / cpt) * cpt;
    // The K loop has ONE body for every chunk (no peeled tail: two copies of the body made the register allocator shuffle the
    // 128 accumulators between them on every iteration): the staging of the last two iterations runs past the end - its
    // loads are parked out of range (no traffic, zeros), its DMA re-reads the last valid chunk, its LDS writes land in the
    // buffer nobody reads any more.
    int ld_q = q0;
    auto load_acts = [&]() {
        const unsigned voff_b = tapoff[ld_vi * NT + tid] | (ld_q < q1 ? 0u : kOOB);     // (row nv of the table exists: 10 rows)
        ++ld_q;
        const unsigned soff_b = (unsigned)(ld_c * BK * HW) * 4u;
#pragma unroll
        for (int e = 0; e < B_PER; ++e)
            rb[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, soff_b + (unsigned)(e * HW) * 4u, 0);
        ++ld_c;
        const int adv = ld_c == cpt;
        ld_vi += adv;
        ld_c = adv ? 0 : ld_c;
    };
    u32x4 ra[A_REG];
    int wq_vi = q0 / cpt, wq_c = q0 - (q0 / cpt) * cpt;
    auto load_w = [&]() {
        const int tap = __builtin_amdgcn_readfirstlane(vtaps[wq_vi < nv ? wq_vi : nv - 1]);
        const int c16 = (tap * p.Cin + wq_c * BK) / 16;
        const unsigned soff_a = (unsigned)(c16 * p.Cout * K16B);
#pragma unroll
        for (int e = 0; e < A_REG; ++e) ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w[e], soff_a, 0);
        ++wq_c;
        const int adv = wq_c == cpt;
        wq_vi += adv;
        wq_c = adv ? 0 : wq_c;
    };
    auto write_w = [&](int buf, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e) *reinterpret_cast<u32x4*>(As0 + buf * BUF + lds_w[e]) = ra[e];
    };
    // the thread's 16 values -> two pieces x two 16-byte units (k-groups 2 kr, 2 kr + 1) of pixel row pl; in slices, so
    // that the K loop can place them between its MFMA groups
    const unsigned st_row = (unsigned)(pl * ROW);
    const int st_sw = (pl >> 1) & 7;
    unsigned pc[2][B_PER / 2];
    auto split_acts = [&](int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e)
            split2h(__builtin_bit_cast(float, rb[2 * e]) * xs, __builtin_bit_cast(float, rb[2 * e + 1]) * xs, pc[0][e], pc[1][e]);
    };
    auto write_acts = [&](int buf) {
        unsigned char* rowp = Bs0 + buf * BUF + st_row;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                *reinterpret_cast<u32x4*>(rowp + (((4 * c + 2 * kr + g) ^ st_sw) << 4)) =
                    u32x4{pc[c][4 * g], pc[c][4 * g + 1], pc[c][4 * g + 2], pc[c][4 * g + 3]};
    };
    auto store_acts = [&](int buf) {
        split_acts(0, B_PER / 2);
        write_acts(buf);
    };

    // ---- prologue: chunk 0 into buffer 0 (DMA + activations), chunk 1's activations in flight
    if (nq > 0) {
        load_w();
        load_acts();
        write_w(0, 0, A_REG);
        store_acts(0);
        load_w();                                       // chunk 1 (or, past the end, a parked repeat: see load_acts)
        load_acts();
    }
    lds_barrier();

#ifdef WSDL_EXP_NOSTAGE          // timing-only build: both LDS images filled once (real data), the loop is barrier + ds_read + MFMA
    if (nq > 1) {
        store_acts(1);
        write_w(1, 0, A_REG);
        lds_barrier();
    }
#endif
    const int l15 = lane & 15, lg = lane >> 4;
    const unsigned fr0 = (unsigned)(l15 * ROW + ((lg ^ (l15 >> 1)) << 4)), fr1 = (unsigned)(l15 * ROW + (((4 + lg) ^ (l15 >> 1)) << 4));
    for (int q = 0; q < nq; ++q) {
        const int cur = q & 1;
        const unsigned char* Ab = As0 + cur * BUF + wm * 128 * ROW;
        const unsigned char* Bb = Bs0 + cur * BUF + wn * 64 * ROW;
        // fragments of chunk q (rows 16 apart keep (row >> 1) & 7 and row & 15 of the lane: one offset pair serves every tile)
        half8 b[TNI][2], a[TMI][2];
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            b[j][0] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + fr0);
            b[j][1] = *reinterpret_cast<const half8*>(Bb + j * 16 * ROW + fr1);
        }
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
            a[i][0] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr0);
            a[i][1] = *reinterpret_cast<const half8*>(Ab + i * 16 * ROW + fr1);
        }
        // Eight MFMA groups (one row of tiles each: 12 MFMAs, 192 cycles of the matrix pipe), each with a slice of the staging
        // of the chunks ahead in front of it; the fences keep the compiler from gathering the MFMAs at the end of the block
        // (it did: 96 MFMAs behind all the staging, nothing overlapped).  Slices: registers (chunk q + 1) -> split -> LDS;
        // loads of chunk q + 2 (after the last split has read the registers); weights of chunk q + 1 by DMA.
        auto mfma_row = [&](int i) {
#ifdef WSDL_EXP_NOMFMA
            if (i == 0) acc[0][0][0] += (float)a[0][0][0] + (float)b[0][0][0];
            return;
#endif
#pragma unroll
            for (int j = 0; j < TNI; ++j) {
                f32x4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        };
#ifndef WSDL_EXP_NOSTAGE
#define WSDL_W4_STAGE(x) x
#else
#define WSDL_W4_STAGE(x)
#endif
#ifdef WSDL_EXP_NODMA
#define WSDL_W4_DMA(x)
#else
#define WSDL_W4_DMA(x) WSDL_W4_STAGE(x)
#endif
#ifdef WSDL_EXP_NOACT
#define WSDL_W4_ACT(x)
#else
#define WSDL_W4_ACT(x) WSDL_W4_STAGE(x)
#endif
        WSDL_W4_DMA(write_w(cur ^ 1, 0, 4));
        WSDL_W4_ACT(split_acts(0, 2));
        mfma_row(0);
        __builtin_amdgcn_sched_barrier(0);
        WSDL_W4_DMA(write_w(cur ^ 1, 4, 8));
        WSDL_W4_ACT(split_acts(2, 4));
        mfma_row(1);
        __builtin_amdgcn_sched_barrier(0);
        WSDL_W4_ACT(split_acts(4, 6));
        mfma_row(2);
        __builtin_amdgcn_sched_barrier(0);
        WSDL_W4_ACT(split_acts(6, 8));
        mfma_row(3);
        __builtin_amdgcn_sched_barrier(0);
        WSDL_W4_ACT(write_acts(cur ^ 1));
        WSDL_W4_DMA(load_w());
        WSDL_W4_ACT(load_acts());
        mfma_row(4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(5);
        mfma_row(6);
        mfma_row(7);
#undef WSDL_W4_STAGE
#undef WSDL_W4_DMA
#undef WSDL_W4_ACT
        // (the loads of chunk q + 2 stay in flight across the barrier: they are consumed in rows 0-3 of the next iteration)
        lds_barrier();
    }

    // ---- epilogue (as conv_igemm_split_kernel's MF form): D[row = 16 i + 4 lg + r][col = 16 j + l15]
    if (p.ksplit > 1) {
        float* sl = p.slab + (long long)blockIdx.z * p.Cout * p.P;
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            const int opix = n0 + wn * 64 + j * 16 + l15;
            if (opix >= W_P) continue;
            const int ob = opix / OHW;
            const int orr = opix - ob * OHW, ooh = orr / w_own;
            const int gpix = ob * OHOW + ooh * p.OW + w_ow0 + (orr - ooh * w_own);
#pragma unroll
            for (int i = 0; i < TMI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m0 + wm * 128 + i * 16 + lg * 4 + r;
                    if (co < p.Cout) sl[(long long)co * p.P + gpix] = acc[i][j][r] * out_scale;
                }
        }
        return;
    }
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < TNI; ++j) {
        const int opix = n0 + wn * 64 + j * 16 + l15;
        if (opix >= W_P) continue;
        const int ob = opix / OHW;
        const int orr = opix - ob * OHW, ooh = orr / w_own;
        const int orp = ooh * p.OW + w_ow0 + (orr - ooh * w_own);
        float* yb = p.y + (long long)ob * p.y_bs + orp;
        const float* rbp = p.res ? p.res + (long long)ob * p.res_bs + orp : nullptr;
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm * 128 + i * 16 + lg * 4 + r;
                if (co >= p.Cout) continue;
                float v = acc[i][j][r] * out_scale;
                if (p.scale) v *= p.scale[co];
                if (p.shift) v += p.shift[co];
                const long long off = (long long)co * OHOW;
                if (rbp) v += rbp[off];
                if (p.accumulate) v += yb[off];
                if (p.relu) v = fmaxf(v, 0.f);
                yb[off] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    if (p.y_amax) publish_amax(vmax, p.y_amax);
}
