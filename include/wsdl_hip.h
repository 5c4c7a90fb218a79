/*
 * wsdl_hip.h - C ABI of libwsdl_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * weakly-supervised segmentation hot path of alexncoleman/WeaklySupervisedDL.
 *
 * The reference has no FFI of its own: every operator below is a chain of PyTorch ATen ops reached
 * from Python (reference file:line cited per entry point, relative to the reference root).  These
 * entry points are what a binding for that path binds instead; INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every tensor is fp32, contiguous NCHW unless a *_bs (batch stride, in elements) says otherwise;
 *     labels are int64 (the host API's dtype), masks uint8;
 *   - all pointers are caller-owned DEVICE pointers; the library allocates nothing persistent,
 *     scratch is an explicit caller-sized workspace (see the *_workspace functions);
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant across
 *     streams, and performs no host synchronisation: launches can be captured into a hipGraph;
 *   - return value 0 = OK, negative = WSDL_E* ; wsdl_last_error() returns a thread-local message.
 *     Geometry is validated on the host before any launch; a rejected call launches nothing.
 */
#ifndef WSDL_HIP_H
#define WSDL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wsdl_stream_t; /* hipStream_t */

enum {
    WSDL_OK = 0,
    WSDL_EINVAL = -1,      /* bad geometry / null pointer / unsupported size */
    WSDL_EHIP = -2,        /* a HIP runtime call failed (message holds hipGetErrorString) */
    WSDL_EWORKSPACE = -3   /* workspace too small */
};

const char* wsdl_last_error(void);
/* Launch trace (diagnostic): wsdl_launch_trace(1) makes the convolution entry points describe every launch they choose -
 * kernel form and tile, arithmetic, K slices, XCD order, column bands, pixel splits, grid - into a per-thread string;
 * wsdl_last_launches() returns this thread's descriptions since its previous call ("; "-separated, valid until the next
 * call) and clears them.  Off (default): one relaxed load per launch site.  The full-size parity tests use it to name
 * the configuration each comparison against the float64 oracle went through. */
int wsdl_launch_trace(int on);
const char* wsdl_last_launches(void);
int wsdl_version(void);               /* 10000*major + 100*minor + patch */
const char* wsdl_target_arch(void);   /* "gfx950" */

/* Process-wide options.  Each is an int read with relaxed atomic loads by every entry point: a call that runs while another
 * host thread sets an option sees the old or the new value, never a torn one.  wsdl_set_option is serialised and REFUSES
 * (WSDL_EINVAL) while any host thread is recording a launch plan - a plan freezes the tile choices of the option set it was
 * recorded under.  Changing "conv_split" / "conv_arith" changes what the weight layout buffers hold: re-run
 * wsdl_conv2d_prep_weights afterwards (callers that cache layouts key them on the option set: ops.LAYOUT_EPOCH).
 *   conv_arith     1*  arithmetic of the split kernels: 1 = fp16x2 (three 16-bit MFMAs per fp32 product, per-tensor
 *                      power-of-two scales from the amax arguments), 0 = bf16x3 (six MFMAs, no scales),
 *                      2 = fp16x2 with the low piece carried at 2^11 and the cross products in a second accumulator
 *                      (forward / input gradient; the RANGE GUARD: a region of a tensor keeps 22 bits down to 2^-29 of the
 *                      tensor's maximum instead of 2^-18 and is exact to 2^-50 of it instead of 2^-39 - same three MFMAs, 64
 *                      more registers per lane, which ends the co-residency with the weight-gradient workgroups: -6.6 % img/s
 *                      on the training step, profiles/r04_notes.md; use it when gradients span more than 2^25 inside one tensor)
 *   conv_split     1*  0 = forward / input-gradient convolutions on the exact-fp32 MFMA kernels everywhere
 *   wgrad_split    1*  0 = weight gradients on the exact-fp32 MFMA kernels everywhere
 *   tile256        1*  256x128 workgroup tiles (512 threads) where they still give >= 256 workgroups
 *   tile64         1*  64x64 tiles for layers that give fewer than `tile_threshold` tiles of 128x64 (128 output rows at 32 x 32 x 16):
 *                      two four-wave workgroups per CU instead of one (round 6: 3-9 % per launch on layer2's convolutions)
 *   split_bk32     1*  K chunks of 32 in the small-tile split forms;   bk32  1*  same for the fp32 kernels
 *   ksplit_big     1*  128x128 tiles + 2 K slices for grids of 200..399 tiles with K >= 2048
 *   tile_threshold 400* workgroups below which the half-size pixel tile is used
 *   col_bands      1*  dilated convolutions: one pixel-tile range per output-column band (exact padding-tap skipping)
 *   xcd_map        1*  XCD-aware tile order of the split kernels (0 off, 1 auto, 10 + py forces py row groups)
 *   wgrad_min_tiles 1*  which shapes the fp16x2 weight-gradient kernel takes: from this many 128-wide N tiles on (rounds 2-5: 6 - from
 *                      one tile on the layer2 1x1 kernels were 20-30 % faster and the step 1 % slower: pre-split, reduce and amax
 *                      launches; round 6, with the pre-split and the maxima coming from the producers: +0.5-0.9 % on the step)
 *   wgrad_xcd      1*  split weight-gradient kernel: XCD-aware tile order (1 contiguous eighths, 2 in 2x2 blocks); bit-identical,
 *                      -1.4 % on the kernel sweep, +0.4 % on the step (half the traffic past L2 for the kernels beside it)
 *   group_tps10   45*  wsdl_conv2d_fwd_group: taps per K slice in tenths (45: a 9-tap problem in 2 slices, the others whole; 20 -
 *                      round 5's default: 4 and 2 slices, eight streams; same box 575.7 us at 45 against 638.6 at 20)
 *   tile_img_major 1*  pixel tiles taken image-fastest inside a column band: 1 = in the grouped forward launch only, 2 = in every
 *                      split launch with taps, 0 = off.  Tiles at the same place of different images run the same tap list;
 *                      bit-identical; grouped forward 601.0 -> 575.7 us, multi-source input gradient 693 -> 709 (hence 1, not 2)
 *   group_interleave 1* ... with the workgroups of its (problem, slice) streams interleaved, one stream per XCD when there are 8
 *   ms_rowfast     1*  wsdl_conv2d_dgrad_multi: XCD-aware tile order taken row tile fastest;  ms_py 0* = 4 row groups (1 / 2 / 4 / 8)
 *   bn_coop        0*  (64 = on for the 64-channel layers) channel-resident BatchNorm kernels with 4 / 2 workgroups per channel for layers of up to this many channels
 *                      (when the caller passes the `coop` counters): 64 channels at B=16, 64x64 forward 17.2 -> 13.8 us, backward
 *                      21.7 -> 16.2 us; two per channel at 128 / 256 channels LOSE 1-9 us to the hand-over (bn_coop_wide 0*)
 *   range_sentinel 0*  1 = the amax arguments of wsdl_bn_train_fwd / _bwd are (max, ~min channel maximum) pairs (wsdl_range_check)
 *   bn_resident    1*  channel-resident fused BatchNorm kernels where a channel fits one workgroup's registers (0 off, 1 = from
 *                      64 channels, n > 1 = from n channels; measured: resident wins at every channel count of the networks)
 *   bn_wide_c    512*  resident BatchNorm kernels as 1024 threads x 4 float4 (instead of 256 x 16) up to this channel count
 *                      (half of it for the backward): 16 waves per CU loading at once where one workgroup per channel would
 *                      leave a CU with 4 - forward -20..23 %, backward -7 % at 256 channels (profiles/r03_bn_kernels.txt)
 *   conv_il        1*  256x128 forward / input-gradient form (all three entry points): steady-state K loop with every wave's MFMAs
 *                      and staging instructions interleaved (branch-free body + scheduling directives) - both waves of a SIMD run
 *                      the same phase, so staging otherwise never overlaps the other wave's MFMAs; bit-identical, 2-8 % faster
 *                      per launch in isolation, +0.5 % on the (power-bound) step
 *   wgrad_blocks 768*  target workgroups of a weight-gradient launch;  wgrad_force_s 0*  fixed number of pixel splits
 *   wgrad_imbalance_split 1*  one more pixel split for the fp16x2 weight gradient of a dilated convolution whose outer taps do less
 *                      than 70 % of the centre tap's work (padding-only chunks are skipped): ASPP d12 463 -> 442 us, d24 357 -> 287
 *   wgrad_bk      16*  pixel chunk of the fp32 weight-gradient kernel (16 | 32)
 *   wgrad_direct   1*  fp16x2 weight-gradient kernel with the x operand's MFMA fragments loaded straight from global memory (no LDS, no
 *                      lane exchange for x; dY double-buffered in LDS, one barrier per chunk) where OW % 32 == 0, stride 1 and every
 *                      tap's column shift is a multiple of 4 elements (1x1, dilation 4 / 12 / 24 / 36): 7-10 % faster there, bit-identical;
 *                      0 = the LDS-staged kernel everywhere
 *   wgrad_dyraw    1*  the direct-fragment kernel reads dY as fp32 and splits it while staging (no dy_split16 pre-pass) for 1x1 convolutions
 *                      with at most 10 N tiles (7-11 % faster there); 2 = for every launch of that kernel (3x3: 7-21 % slower), 0 = never
 *   wgrad_chan_scale 0* the range guard of the weight gradient (conv_arith = 2 is the forward / input gradient's): the fp16x2 weight-gradient
 *                      kernels scale x and dY by one power of two per CHANNEL instead of per tensor - a channel is a row / a column
 *                      of that GEMM's output (K = pixels), so the inverse scales go onto the result's rows and columns: exact, no
 *                      second accumulator set.  A pre-pass takes the per-channel maxima (one more read of x and dY per layer)
 *   stem_kernel    1*  7x7 stride-2 convolution of 3 -> 64 channels (ResNet's conv1) on its own kernel: input patch and all weights
 *                      in LDS, fp32 MFMA (0: the generic fp32 implicit-GEMM kernel, the A/B partner)
 *   ksplit_target 512* / ksplit_max 8* / ksplit_min_chunks 4*  small grids (CAM path at B=8): workgroups aimed at by the K split,
 *                      most K slices, fewest 32-deep chunks per slice
 *   layercam_tail_mod 32*  LayerCAM epilogue: the last hw % n pixels of a map are summed over the channels the way ATen's CPU sum
 *                      handles its scalar columns (four interleaved streams), all others by its four-level cascade - the epilogue
 *                      is bit-identical to the reference's torch-CPU arithmetic on identical inputs.  32 = torch built for AVX-512
 *                      run on 1-12, 24 or 32 threads (the fixtures: torch.set_num_threads(4)), 16 = an AVX2 torch, 0 = the
 *                      cascade for every pixel.  torch-CPU's own result depends on its THREAD COUNT: ATen splits the
 *                      columns of the sum over its threads and the thread left with fewer than a vector's worth takes the
 *                      scalar path, so no single setting reproduces every reference run - with 16 threads (the GPU boxes'
 *                      default) 32 leaves 2 of 196 values of a 1024 x 14 x 14 sum one ulp off and 0 leaves 6 of 784 (28 x 28) and 9
 *                      of 49 (7 x 7); pin the reference's thread count (1-12) when bit-exact CAM values are compared
 * (Options measured slower and removed in round 3: conv_glds - weights by LDS-DMA; wgrad_wide - 8-pixel-run staging of x;
 *  wgrad_tile64 - 64-row weight-gradient tiles; occupancy_cap.  Figures: profiles/r02_notes.md.  Round 4: t256_bk32 - K chunks of 32
 *  in the 256x128 form (1-2.6 % slower per step); wgrad_direct = 2 - the direct-fragment weight gradient for misaligned taps (0-7 %
 *  slower, spills); conv_mfma16 / wgrad_mfma16 = 0 - the 32x32x16 MFMA shape in the K-chunk-32 forward forms and in the fp16x2 weight
 *  gradient (2.2 % / 2 % slower on the step; v_mfma_f32_16x16x32_f16 is what those kernels run on).)
 * (* = default). */
int wsdl_set_option(const char* name, int value);

/* ---- per-kernel-class timing (bench.py roofline leg) --------------------------------------
 * When enabled every launch of the instrumented classes is bracketed by hipEventRecord on the
 * launch stream.  wsdl_prof_collect synchronises the events and returns, per class, the number of
 * launches, summed milliseconds and summed algorithmic work (flops for conv classes, bytes else).
 * Classes are kernel instantiations, so a class lines up with one row of `rocprofv3 --stats`. */
enum { WSDL_PROF_IGEMM_128x128_A = 0,  /* conv_igemm_fast_kernel<128,128,2,16> (forward + dgrad launches) */
       WSDL_PROF_IGEMM_128x128_U = 1,  /* conv_igemm_kernel<128,128,2,false> (K not a multiple of 16)     */
       WSDL_PROF_IGEMM_128x64_A = 2, WSDL_PROF_IGEMM_128x64_U = 3,
       WSDL_PROF_IGEMM_64x256_A = 4, WSDL_PROF_IGEMM_64x256_U = 5,
       WSDL_PROF_IGEMM_64x128_A = 6, WSDL_PROF_IGEMM_64x128_U = 7,
       WSDL_PROF_WGRAD_128x128 = 8,    /* conv_wgrad_kernel<128,128,2> */
       WSDL_PROF_WGRAD_64x128 = 9,     /* conv_wgrad_kernel<64,128,1>  */
       WSDL_PROF_WGRAD_FAST_128x128 = 10,  /* conv_wgrad_fast_kernel<128,128,2,16> (and its 128x64 / 64x128 / 64x64 forms) */
       WSDL_PROF_PAIRWISE = 11, WSDL_PROF_LAYERCAM = 12,
       /* split kernels (three fp16 / six bf16 MFMAs per fp32 product; work counted in fp32-equivalent FLOPs);
        * the last template argument is the arithmetic (1 = fp16x2, 0 = bf16x3) */
       WSDL_PROF_SPLIT_128x128 = 13,   /* conv_igemm_split_kernel<128,128,2,16,256,AR> (forward + dgrad launches) */
       WSDL_PROF_SPLIT_128x64 = 14, WSDL_PROF_SPLIT_64x256 = 15, WSDL_PROF_SPLIT_64x128 = 16,
       WSDL_PROF_WGRAD_SPLIT32 = 17,   /* conv_wgrad_split32_kernel<128,128,AR> */
       WSDL_PROF_SPLIT_256x128 = 18,   /* conv_igemm_split_kernel<256,128,4,32|16,512,AR> */
       WSDL_PROF_STEM = 19,            /* stem_conv7x7s2_kernel */
       WSDL_PROF_WGRAD_SPLIT16D = 20,  /* conv_wgrad_split16d_kernel<MODE, DYRAW> (x fragments straight from global memory); class 17 is then
                                          the LDS-staged conv_wgrad_split16_kernel / conv_wgrad_split32_kernel.  Both brackets include the
                                          launch's dy_split16 pre-pass where there is one */
       WSDL_PROF_SPLIT_GROUP = 21,     /* conv_igemm_split_group_kernel<256,128,4,16,512,AR,false>: several forward convolutions, one launch */
       WSDL_PROF_SPLIT_MULTI = 22,     /* conv_igemm_split_kernel<256,128,4,16,512,AR,false,true>: several input gradients, one accumulator */
       WSDL_PROF_NCLASSES = 23 };
const char* wsdl_prof_class_name(int cls);
int wsdl_prof_enable(int on);
int wsdl_prof_collect(int cls, long long* launches, double* total_ms, double* total_work,
                      double* total_work_executed /* work minus the skipped all-padding K-chunks */,
                      double* total_bytes /* algorithmic HBM bytes: every operand read once, result written once */);
int wsdl_prof_reset(void);

/* ---- profiler ranges (roctx) ----------------------------------------------------------------
 * Named host-side ranges for `rocprofv3 --marker-trace`: wsdl_range_enable(1) loads librocprofiler-sdk-roctx.so (dlopen; the
 * library does not link against it) and from then on wsdl_range_push / _pop forward to roctxRangePushA / roctxRangePop, and
 * every launch of an instrumented kernel class (the classes above) sits inside a range carrying the class name.  Disabled
 * (the default) all three cost one branch.  Returns 0, or WSDL_EINVAL when the roctx library cannot be loaded.
 * The reference has no tracing of any kind: this is the build's addition. */
int wsdl_range_enable(int on);
int wsdl_range_push(const char* name);
int wsdl_range_pop(void);

/* ---- convolution: implicit GEMM on the matrix cores -------------------------------------------
 * Arithmetic paths, all at fp32-level accuracy (tools/conv_accuracy.py, tests/test_hip_ops.py measure them against fp64):
 *   fp32 : v_mfma_f32_32x32x2_f32, exact fp32 fma chains;
 *   split: every fp32 operand is split exactly into 16-bit pieces and a product is a few 16-bit MFMA partial products
 *          accumulated in fp32 - used whenever the contracted channel count is a multiple of 16 and kh*kw <= 9
 *          (wsdl_set_option("conv_split", 0) / ("wgrad_split", 0) select the fp32 kernels everywhere):
 *            fp16x2 (default): x*s = h + l in fp16 (22 bits), three v_mfma_f32_32x32x16_f16 per product; s is a power of
 *                    two per tensor, derived in the kernel from the `*_amax` device scalars (>= max|tensor|, e.g. from
 *                    wsdl_bn_train_fwd / wsdl_amax), which are therefore REQUIRED (non-NULL) for these launches;
 *            bf16x3 ("conv_arith" 0): x = h + m + l in bf16 (24 bits), six v_mfma_f32_32x32x16_bf16, amax unused.
 * Replaces the ATen conv2d forward / input-gradient / weight-gradient reached from
 *   torchvision ResNet-50 and DeepLabV3 convs called at TraditionalModel/ClassificationModel.py:29-33,
 *   TraditionalModel/SegmentationModel.py:102,110, TraditionalModel/AlternatingDirectionCutLoss.py:697-703,
 *   and the class-logit backward at TraditionalModel/LayerCAM.py:48.
 * Square kernels, one stride / padding / dilation for both spatial dims, groups = 1.
 *   OH = (H + 2*pad - dil*(kh-1) - 1)/stride + 1 (same for OW).                                   */

/* Re-layout w[Cout][Cin][kh][kw] for the kernels.  The layout buffers are opaque to the caller; their size
 * comes from wsdl_conv2d_weight_layout_bytes (dgrad = 0: forward layout, 1: input-gradient layout):
 *   plain  (fp32 MFMA kernels)      : wt_fwd[(tap*Cin+ci)][Cout], wt_dgrad[(tap*Cout+co)][Cin]  fp32;
 *   split  (used when the contracted channel count % 16 == 0 and kh*kw <= 9):
 *            [(k/16)][row][piece][k%16] 16-bit pieces (fp16x2: w * 2^e = h + l, 4 bytes per weight; bf16x3: h + m + l,
 *            6 bytes per weight) + a 16-byte trailer holding max|w| (the scale the kernels derive 2^e from).
 * *is_plain (optional) tells which one the current options select; for a plain 1x1 kernel the dgrad layout
 * equals w itself.  Either destination of prep_weights may be NULL. */
size_t wsdl_conv2d_weight_layout_bytes(int Cout, int Cin, int kh, int kw, int dgrad, int* is_plain);
int wsdl_conv2d_prep_weights(const float* w, void* wt_fwd, void* wt_dgrad,
                             int Cout, int Cin, int kh, int kw,
                             const float* w_amax /* optional device scalar max|w| (wsdl_multi_amax); NULL: reduced here */,
                             wsdl_stream_t stream);

/* y = act( scale[co]*conv(x) + shift[co] + residual ), any of scale/shift/residual may be NULL
 * (scale NULL = 1, shift NULL = 0).  relu != 0 applies max(.,0).  x_bs / y_bs / res_bs: batch strides
 * in elements (0 = dense).  Folded eval-mode BatchNorm, conv bias and the Linear layer (a 1x1 conv on
 * a 1x1 map) all go through scale/shift. */
int wsdl_conv2d_fwd(const float* x, const void* wt_fwd, float* y,
                    int B, int Cin, int H, int W, int Cout, int kh, int kw,
                    int stride, int pad, int dil,
                    const float* scale, const float* shift, const float* residual, int relu,
                    long long x_bs, long long y_bs, long long res_bs,
                    const float* x_amax /* device scalar >= max|x| (fp16x2 split launches; else may be NULL) */,
                    float* y_amax /* optional: atomicMax of max|y| into a ZEROED device scalar */,
                    void* ws, size_t ws_bytes, wsdl_stream_t stream);
/* The split layouts of MANY convolutions in one launch (the re-layout after an optimiser step).  Entries must be convolutions
 * whose forward AND dgrad layouts are split layouts (wsdl_conv2d_weight_layout_bytes: plain = 0 for both; kh*kw <= 9) under the
 * fp16x2 arithmetic; w_amax as in wsdl_conv2d_prep_weights but REQUIRED.  `desc` is a DEVICE array of n entries, block_begin =
 * the running sum of grid_x * grid_y with grid_x = ceil(Cin / 32), grid_y = ceil(Cout / 32); total_blocks = its final value. */
typedef struct wsdl_prep_desc {
    const float* w;          /* [Cout][Cin][taps] */
    void* wt_fwd;            /* may be NULL */
    void* wt_dgrad;          /* may be NULL */
    const float* w_amax;     /* device scalar >= max|w| */
    int Cout, Cin, taps, grid_x;
    int block_begin, reserved;
} wsdl_prep_desc;
int wsdl_conv2d_prep_weights_multi(const wsdl_prep_desc* desc, int n, int total_blocks, wsdl_stream_t stream);

/* Optional scratch for forward / dgrad: grids too small to fill 256 CUs (small batches of small maps) are split
 * along K into slabs summed in fixed order.  Returns 0 when the geometry does not benefit; ws may be NULL. */
size_t wsdl_conv2d_igemm_workspace(int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                   int stride, int pad, int dil, int dgrad);

/* dx = conv_transpose(dy, w)  (+ dx if accumulate).  dy is (B,Cout,OH,OW) with batch stride dy_bs.
 * acc_mask (optional, with accumulate): bit e % 8 of byte e / 8 says whether element e of the value already in dx counts -
 * dx then holds a bottleneck block's output gradient and the mask is the block's final ReLU as written by
 * wsdl_bn_train_fwd(relu_mask): dx = dgrad + [y > 0] * dx, the identity branch's gradient without a tensor of its own. */
int wsdl_conv2d_dgrad(const float* dy, const void* wt_dgrad, float* dx,
                      int B, int Cin, int H, int W, int Cout, int kh, int kw,
                      int stride, int pad, int dil, int accumulate, const uint8_t* acc_mask,
                      long long dy_bs, const float* dy_amax, void* ws, size_t ws_bytes, wsdl_stream_t stream);
/* The input gradient of n (2..4) convolutions that read the SAME input, in one launch:
 *   dx = [accumulate: dx +] sum_i dgrad(dy[i], wt_dgrad[i])      (1x1 / 3x3, stride 1, padding dil*(k-1)/2: 'same')
 * - the ASPP head of DeepLabV3, whose 2048-channel input feeds a 1x1 and three dilated 3x3 branches (torchvision's
 * ASPP.forward; reference TraditionalModel/SegmentationModel.py:85 builds it) and whose input gradient the reference
 * obtains as four conv-backward calls and three tensor adds.  The output tile accumulates over all sources' taps in
 * registers (per-source power-of-two scales, the accumulators are moved between them exactly) and is stored once.
 * dy / wt_dgrad / dy_amax / k / dil / dy_bs are HOST arrays of n entries (dy_bs may be NULL: dense).  Source 0 decides the
 * output-column bands of the tap skipping: pass the smallest dilation > 1 first.  wsdl_conv2d_dgrad_multi_ok says whether a
 * geometry is served (else: chain wsdl_conv2d_dgrad calls with accumulate = 1). */
/* n (2..4) forward convolutions of ONE input in one launch - ASPP's 1x1 and three dilated 3x3 branches (torchvision's
 * ASPP.forward, built by reference TraditionalModel/SegmentationModel.py:85): y[i] = conv(x, w[i]), 1x1 / 3x3, stride 1, 'same'
 * padding, raw outputs (the BatchNorm kernels follow).  The branches execute 1..9 taps per pixel tile (padding taps are
 * skipped): launched one by one each ends with idle CUs behind its longest tiles; here the workgroups of all problems form
 * one grid, heaviest problem first, tiles of more than two taps cut into K slices (slabs in the workspace + the fixed-order
 * reduce).  wt_fwd / y / k / dil / y_bs: HOST arrays of n entries (y_bs may be NULL: dense). */
int wsdl_conv2d_fwd_group_ok(int n, int B, int Cin, int H, int W, int Cout);
size_t wsdl_conv2d_fwd_group_workspace(int n, const int* k, const int* dil, int B, int Cin, int H, int W, int Cout);
int wsdl_conv2d_fwd_group(int n, const float* x, const void* const* wt_fwd, float* const* y, const int* k, const int* dil,
                          int B, int Cin, int H, int W, int Cout, long long x_bs, const long long* y_bs,
                          const float* x_amax, void* ws, size_t ws_bytes, wsdl_stream_t stream);
int wsdl_conv2d_dgrad_multi_ok(int n, int B, int Cin, int H, int W, int Cout);
int wsdl_conv2d_dgrad_multi(int n, const float* const* dy, const void* const* wt_dgrad, const float* const* dy_amax,
                            const int* k, const int* dil, const long long* dy_bs, float* dx, int B, int Cin, int H, int W,
                            int Cout, int accumulate, wsdl_stream_t stream);

/* dw[Cout][Cin][kh][kw] = sum_{b,oh,ow} dy * x_shifted  (+ dw if accumulate).  Split over pixel
 * ranges into fp32 slabs in `ws`, summed in fixed order by a second kernel (bitwise reproducible). */
size_t wsdl_conv2d_wgrad_workspace(int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                   int stride, int pad, int dil);
int wsdl_conv2d_wgrad(const float* x, const float* dy, float* dw,
                      int B, int Cin, int H, int W, int Cout, int kh, int kw,
                      int stride, int pad, int dil, int accumulate,
                      long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax,
                      void* ws, size_t ws_bytes, wsdl_stream_t stream);
/* Weight gradient with the slab reduction DEFERRED (round 6).  wsdl_conv2d_wgrad splits the pixel dimension over S workgroup
 * slabs and ends with a small launch that adds them in fixed order into dw - ~60 such launches per training step, 9-10 us each
 * for a few MB (launch and tail latency, not bandwidth).  The deferred form runs everything but that launch and fills *desc (a
 * HOST struct) with what is left to do; the caller collects the descriptors of many layers, uploads them as ONE device table
 * (block_begin = the running sum of nblocks) and runs wsdl_wgrad_reduce_multi when the gradients are needed - before the
 * optimiser step or a gradient bucket's all-reduce (reference: the loss.backward() / optimizer.step() pair of
 * TraditionalModel/SegmentationModel.py:110-111).  Same sums in the same order: bit-identical to wsdl_conv2d_wgrad.
 *   - the slabs live in `ws`: the caller keeps each layer's workspace untouched until the multi launch has run;
 *   - desc->kind < 0: nothing is pending (the call reduced by itself: a batch processed in slices);
 *   - two deferred gradients into the same dw must not share a multi launch (flush in between). */
enum { WSDL_WGRAD_REDUCE_PLAIN = 0, WSDL_WGRAD_REDUCE_TILED = 1, WSDL_WGRAD_REDUCE_VEC4 = 2, WSDL_WGRAD_REDUCE_MANY = 3,
       WSDL_WGRAD_REDUCE_TRANSPOSED = 4, WSDL_WGRAD_REDUCE_MANY16 = 5 };
typedef struct wsdl_wgrad_reduce_desc {
    const float* slab;       /* [S][Cout][taps*Cin]   (TRANSPOSED: [S][Cin][Cout]) */
    float* dw;               /* [Cout][Cin][taps] */
    unsigned long long live; /* bit t: tap t has slab data */
    int S, Cout, Cin, T;
    int accumulate, kind;
    int grid_x, nblocks;     /* blocks of 256 threads this reduction takes (grid_x: its inner extent where it is 2-D) */
    int block_begin, reserved;
} wsdl_wgrad_reduce_desc;
int wsdl_conv2d_wgrad_deferred(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                               int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax, void* ws,
                               size_t ws_bytes, wsdl_wgrad_reduce_desc* desc, wsdl_stream_t stream);
/* The weight gradient with PER-CHANNEL operands (round 6; both optional, NULL = as wsdl_conv2d_wgrad):
 *   x_chan_amax[Cin] / dy_chan_amax[Cout]: one maximum per channel of x / dY, as the channel-resident BatchNorm kernels publish
 *     them (wsdl_bn_train_fwd / _bwd chan_amax).  The fp16x2 kernels then scale each channel by its OWN power of two - exact (a
 *     channel is a row / column of this GEMM's output) and the range guard of the weight gradient for nothing: a channel 2^30
 *     below the tensor's maximum keeps its 22 bits.  "wgrad_chan_scale" = 1 takes missing maxima with a pre-pass instead.
 *   dy_presplit: dY already as the kernel's fp16 (high, low) rows [ceil(P / 32)][Cout][128 B], written by the BatchNorm backward that
 *     produced dY (wsdl_bn_train_bwd dy_presplit) with the scales of dy_chan_amax: dy_split16_kernel (one more read and write of
 *     dY per layer, 24 launches per training step) does not run.  wsdl_conv2d_wgrad_presplit_bytes: the buffer's size for a
 *     geometry, 0 where the weight gradient would not use it.
 *   desc_or_null: non-NULL = the deferred form (wsdl_conv2d_wgrad_deferred). */
size_t wsdl_conv2d_wgrad_presplit_bytes(int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil);
int wsdl_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W,
                         int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                         long long x_bs, long long dy_bs, const float* x_amax, const float* dy_amax,
                         const float* x_chan_amax, const float* dy_chan_amax, const void* dy_presplit, void* ws,
                         size_t ws_bytes, wsdl_wgrad_reduce_desc* desc_or_null, wsdl_stream_t stream);
/* desc: DEVICE array of n entries; total_blocks = the sum of their nblocks */
int wsdl_wgrad_reduce_multi(const wsdl_wgrad_reduce_desc* desc, int n, int total_blocks, wsdl_stream_t stream);

/* out = max|x| over B images of per_image contiguous floats (batch stride x_bs elements, 0 = dense); zero_first != 0
 * zeroes `out` first (else the caller passes a zeroed scalar).  For tensors whose producer did not publish an amax
 * (network input, concatenations, dropout outputs).  wsdl_multi_amax: n tensors in one launch - ptrs / counts are
 * DEVICE arrays of n pointers / element counts, out[n] is zeroed first (every conv weight after the optimiser step). */
int wsdl_amax(const float* x, int B, long long per_image, long long x_bs, float* out, int zero_first,
              wsdl_stream_t stream);
int wsdl_multi_amax(const float* const* ptrs, const long long* counts, int n, float* out, wsdl_stream_t stream);

/* dbias[co] = sum_{b,hw} dy  (classifier[4] / fc bias gradient). */
int wsdl_bias_grad(const float* dy, float* dbias, int B, int C, int HW, long long dy_bs,
                   int accumulate, wsdl_stream_t stream);

/* ---- BatchNorm2d (train-mode batch statistics; torchvision BN inside the models above) ------ */
size_t wsdl_bn_workspace(int C);
/* y = act( (x-mean)*invstd*gamma + beta + residual ); saves mean / invstd (biased var), updates
 * running_mean / running_var (unbiased var, momentum) when they are non-NULL. */
int wsdl_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y,
                      float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                      float momentum, float eps, int B, int C, int HW,
                      const float* residual, int relu, long long y_bs,
                      float* y_amax /* optional: atomicMax of max|y| into a ZEROED device scalar */,
                      uint8_t* relu_mask /* optional (relu, HW % 8 == 0, y_bs % 4 == 0): B*C*HW/8 bytes, bit e%8 of byte
                                            e/8 = [y > 0] for the dense element index e - for relu = 3 of the backward */,
                      void* ws, size_t ws_bytes,
                      int* coop /* optional: 2*C ZERO-INITIALISED ints that persist between calls, one region per stream.  Given
                                   them, layers of few channels (option bn_coop: 0* = off, 64) run several workgroups per channel, which
                                   hand their partial sums over through the workspace (64 channels at B=16, 64x64: 17.2 -> 13.8 us;
                                   profiles/r05_notes.md); the kernel leaves the counters zeroed.  NULL: one workgroup per channel */,
                      float* chan_amax /* optional [C]: max|y| per CHANNEL (a plain store by the channel's workgroup; wants y_amax and
                                          wsdl_bn_channel_resident(B, C, HW, 0)) - the per-channel scales of the weight gradient that
                                          reads y as its x operand (wsdl_conv2d_wgrad_ex) */,
                      wsdl_stream_t stream);
/* 1 when (B, C, HW) runs the channel-resident kernel (one workgroup holds a whole channel in registers: what chan_amax /
 * dy_presplit need), forward (backward = 0) or backward (1), under the current options. */
int wsdl_bn_channel_resident(int B, int C, int HW, int backward);
/* Backward of the above.  relu = 1: the ReLU mask is read from the forward output y (needed when a residual was
 * added); relu = 2: the mask is recomputed from x - y = fma(x - mean, invstd*gamma, beta), the forward's own pinned
 * expression - so y is neither read nor needs keeping (y may be NULL, beta is required); relu = 3: the mask is read from
 * the bits the forward wrote (relu_mask: 1/32 of y's bytes - the residual layers' backward reads four tensor streams
 * instead of five; same result bit for bit as relu = 1); relu = 0: no activation.
 * dres (optional) receives the masked upstream gradient (the residual branch's gradient). */
int wsdl_bn_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd,
                      float* dx, float* dgamma, float* dbeta, float* dres,
                      int B, int C, int HW, int relu, int accumulate_param_grads,
                      long long dy_bs, long long y_bs,
                      float* dx_amax /* optional: atomicMax of max|dx| into a ZEROED device scalar */,
                      const uint8_t* relu_mask /* relu = 3 */,
                      void* ws, size_t ws_bytes, int* coop /* as for the forward */,
                      float* chan_amax /* optional [C]: max|dx| per channel (as for the forward; wsdl_bn_channel_resident(.., 1)) */,
                      void* dy_presplit /* optional: dx ALSO as the fp16 (high, low) rows the producing convolution's weight gradient
                                           reads ([B*HW / 32][C][128 B], each channel scaled by the power of two of its chan_amax;
                                           size: wsdl_conv2d_wgrad_presplit_bytes) - the channel's workgroup holds it in registers
                                           anyway.  Wants chan_amax and B*HW % 32 == 0 (dx itself is always dense) */,
                      wsdl_stream_t stream);
/* eval-mode fold: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale (fed to wsdl_conv2d_fwd). */
int wsdl_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, float* scale, float* shift, int C,
                 wsdl_stream_t stream);
/* y = act(scale[c]*x + shift[c]) - a stand-alone eval-mode BatchNorm2d (scale / shift from wsdl_bn_fold) or a
 * stand-alone ReLU (scale = shift = NULL); the models run both fused behind the convolution instead. */
int wsdl_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                        int relu, wsdl_stream_t stream);
/* backward of y = act(scale*conv + shift + res) wrt conv: dconv = dy*[y>0]*scale ; dres = dy*[y>0] */
int wsdl_affine_act_bwd(const float* dy, const float* y, const float* scale, float* dconv, float* dres,
                        int B, int C, int HW, int relu,
                        float* dconv_amax /* optional: atomicMax of max|dconv| into a ZEROED device scalar */,
                        wsdl_stream_t stream);

/* ---- pooling / resampling / elementwise ----------------------------------------------------- */
/* MaxPool2d(3, stride 2, pad 1) as in ResNet's stem; argmax (0..8, first max wins) kept as uint8 */
int wsdl_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int BC, int H, int W,
                          wsdl_stream_t stream);
int wsdl_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int BC, int H, int W,
                          wsdl_stream_t stream);
/* AdaptiveAvgPool2d(1) */
int wsdl_global_avgpool_fwd(const float* x, float* y, int BC, int HW, wsdl_stream_t stream);
int wsdl_global_avgpool_bwd(const float* dy, float* dx, int BC, int HW, int accumulate,
                            wsdl_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=False): (BC,h,w) -> (BC,H,W).  y_bs/C: output may be a
 * channel slice of a wider tensor (C planes per image, batch stride y_bs elements; 0 = dense). */
int wsdl_bilinear_fwd(const float* x, float* y, int B, int C, int h, int w, int H, int W,
                      long long y_bs, wsdl_stream_t stream);
int wsdl_bilinear_bwd(const float* dy, float* dx, int B, int C, int h, int w, int H, int W,
                      long long dy_bs, wsdl_stream_t stream);
/* Dropout: mask is uint8 0/1.  If gen_mask != 0 the mask is drawn (counter hash of seed, element
 * index) and written; else it is read (injected mask, parity tests).  y = x*mask/(1-p).
 * seed_dev (optional): a device-resident call counter mixed into the seed - a captured (hipGraph) launch replays with a
 * frozen host `seed`, the counter (bumped on the stream by the caller) keeps the masks changing. */
int wsdl_dropout_fwd(const float* x, float* y, uint8_t* mask, size_t n, float p,
                     unsigned long long seed, int gen_mask, const unsigned long long* seed_dev,
                     wsdl_stream_t stream);
int wsdl_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, size_t n, float p,
                     wsdl_stream_t stream);
/* y = a + b (optionally relu), y = alpha*x, strided channel-slice copy */
int wsdl_add(const float* a, const float* b, float* y, size_t n, int relu, wsdl_stream_t stream);
int wsdl_scale_by_device_scalar(const float* x, const float* s, float* y, size_t n, wsdl_stream_t stream);
int wsdl_copy_planes(const float* src, float* dst, int B, int C, int HW, long long src_bs,
                     long long dst_bs, wsdl_stream_t stream);

/* ---- losses --------------------------------------------------------------------------------- */
size_t wsdl_reduce_workspace(void);
/* lovasz_softmax(probas, labels, classes, per_image=False, ignore) - the optional loss of train_segmentation_model
 * (TraditionalModel/SegmentationModel.py:103-105; LossFunctions/Lovasz-Softmax_Loss.py: lovasz_grad :11-23,
 * lovasz_softmax_flat :164-192, flatten_probas :195-211).  probas (B,C,H,W) class probabilities, labels int64 (B,H,W).
 * *loss = mean over the classes that occur (classes_all = 0, 'present') or over all C (classes_all = 1, 'all') of
 * <errors sorted descending, Jaccard-gradient of the sorted foreground>; dprobas (optional, (B,C,H,W)) = d loss / d probas
 * (the Jaccard gradient is a constant of the sort order, as in the reference).  ignore_label: pixels with that label take
 * no part (pass a value no label has, e.g. -1, for None).  One stable radix sort (rocPRIM) + one scan + one pass per class;
 * bitwise reproducible; ties in the errors are ranked by pixel index. */
size_t wsdl_lovasz_softmax_workspace(int B, int C, int H, int W);
int wsdl_lovasz_softmax_fwd_bwd(const float* probas, const int64_t* labels, float* loss, float* dprobas, int B, int C,
                                int H, int W, int classes_all, long long ignore_label, void* ws, size_t ws_bytes,
                                wsdl_stream_t stream);

/* nn.CrossEntropyLoss() on (B,C,H,W) logits and int64 (B,H,W) labels
 * (TraditionalModel/SegmentationModel.py:90,107; AlternatingDirectionCutLoss.py:789,699): mean over the pixels whose
 * label != ignore_index (PyTorch's default is -100; such pixels get zero loss and zero gradient).  Any other label
 * outside [0,C) - PyTorch raises - makes the loss NaN: a kernel cannot raise, and it must not train the pixel as a
 * real class.  dlogits (optional) = grad_scale * (softmax - onehot), NOT yet divided by the valid-pixel count:
 * *inv_count (device scalar, required with dlogits) receives 1/count and is applied together with the upstream
 * gradient (wsdl_scale_by_device_scalar). */
int wsdl_softmax_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                            float* inv_count, int B, int C, int H, int W, float grad_scale,
                            long long ignore_index, void* ws, size_t ws_bytes, wsdl_stream_t stream);
/* Pairwise-affinity loss over a reflect-padded window x window neighbourhood:
 *   apply_softmax=1, normalise=0, sigma_space<=0 : LocalNormalizedCutLoss.forward
 *                                  (TraditionalModel/AlternatingDirectionCutLoss.py:65-105)
 *   apply_softmax=0, normalise=1                 : ConstrainToBoundaryLossSingle.forward
 *                                  (TraditionalModel/AlternatingDirectionBoundaryLoss.py:12-70)
 * loss: 1 float (normalise=0) or B floats (normalise=1).  dpreds (optional) = d loss / d preds for
 * an upstream gradient of 1 (per image for normalise=1).
 * cache (optional): the image's colour affinities from wsdl_pairwise_cache (same B, H, W, window, sigma_color); the
 * kernel then reads them instead of evaluating 24 exponentials per pixel - refine_pseudo_mask evaluates the loss 10 x 5
 * times on one image (AlternatingDirectionCutLoss.py:736-757, :803-810).  Bit-identical results either way. */
int wsdl_pairwise_affinity_loss_fwd_bwd(const float* preds, const float* image, float* loss,
                                        float* dpreds, int B, int C, int H, int W, int window,
                                        float sigma_color, float sigma_space, int apply_softmax,
                                        int normalise, const float* cache, void* ws, size_t ws_bytes,
                                        wsdl_stream_t stream);
size_t wsdl_pairwise_workspace(int B, int H, int W);
/* exp(-|I_q - I_p|^2 / (2 sigma_color^2)) for the (window^2 - 1) / 2 "forward" offsets of every pixel (the affinity
 * is symmetric): cache[(window^2-1)/2][B][H][W] floats, wsdl_pairwise_cache_bytes of them. */
size_t wsdl_pairwise_cache_bytes(int B, int H, int W, int window);
int wsdl_pairwise_cache(const float* image, float* cache, int B, int H, int W, int window, float sigma_color,
                        wsdl_stream_t stream);
/* compute_affinities (TraditionalModel/AlternatingDirectionCutLoss.py:612-637): K=(w*w-1) maps,
 * out[(k*B + b)*H*W + p]. */
int wsdl_compute_affinities(const float* image, float* out, int B, int H, int W, int window,
                            float sigma_color, float sigma_space, wsdl_stream_t stream);

/* ---- LayerCAM epilogue (TraditionalModel/LayerCAM.py:52-76; variant 1 = notebook arithmetic,
 * AlternatingDirectionCutLoss.py:261-284) + threshold of PsuedoMasks.py:59-62 ------------------
 * act[l], grad[l]: (B,C[l],h[l],w[l]) device pointers, given as HOST arrays of n_layers entries.
 * cam: (B,outH,outW).  mask (optional, thresh >= 0): uint8 (cam >= thresh && cam > 0).
 * Every operation and the ORDER of the channel sum follow the reference's PyTorch-CPU path (ATen cascade_sum; bilinear
 * weights and blends as UpSampleKernel.cpp evaluates them): on identical act / grad the CAM and the mask are bit-identical
 * to the reference's for alpha in {1, 0.5, 2, 3} (csrc/layercam_optim.hip; option layercam_tail_mod). */
size_t wsdl_layercam_workspace(int n_layers, int B, const int* C, const int* h, const int* w);
int wsdl_layercam_epilogue(const float* const* act, const float* const* grad, const int* C,
                           const int* h, const int* w, int n_layers, int B, int outH, int outW,
                           float alpha, int variant, float* cam, float thresh, uint8_t* mask,
                           void* ws, size_t ws_bytes, wsdl_stream_t stream);

/* The class-logit head of the LayerCAM pass in one call (TraditionalModel/LayerCAM.py:41-48 - `logits, _ = model(images)`,
 * `class_idx = argmax(logits)` when none is given, `logits.gather(1, class_idx).backward(ones)` - through
 * ClassificationModel.py:35-37, `fc(avgpool(f4).view(B, -1))`): pooled (B,C) = wsdl_global_avgpool_fwd of layer4's output,
 * weight (K,C), bias (K) or NULL, class_idx (B) int64 or NULL -> logits (B,K), cls (B) int32 (the class each image was
 * differentiated for; -1 and a NaN gradient for an index outside [0,K)), dx (B,C,HW) = weight[cls[b]][c] / HW: the gradient of
 * the chosen logit with respect to layer4's output (fc's backward followed by the average pool's, exactly: both are linear).
 * The parameters' own gradients (fc.weight.grad, fc.bias.grad - a side effect of the reference's backward nothing reads)
 * are not formed. */
int wsdl_class_logit_head(const float* pooled, const float* weight, const float* bias, const long long* class_idx,
                          float* logits, int* cls, float* dx, int B, int C, int K, int HW, wsdl_stream_t stream);

/* keep_largest (TraditionalModel/PsuedoMasks.py:15-21: skimage label + regionprops, the largest area wins, the first
 * label on ties, an empty mask stays empty) for n masks (n,h,w) of uint8 (non-zero = foreground) -> out (n,h,w) in {0,1};
 * 8-connectivity, labels in raster order of a component's first pixel as skimage numbers them.  One workgroup per mask,
 * labels in LDS up to 65535 pixels (224 x 224), in the workspace beyond.  out may alias mask.  Bit-exact. */
size_t wsdl_keep_largest_workspace(int n, int h, int w);
int wsdl_keep_largest(const uint8_t* mask, uint8_t* out, int n, int h, int w, void* ws, size_t ws_bytes,
                      wsdl_stream_t stream);

/* classic CAM normalisation (CAMGenerator.generate_all_cams, TraditionalModel/AlternatingDirectionCutLoss.py:343-372):
 * y = (relu(x) - min) / (max + 1e-8) per plane; the class-weighted channel sum itself is wsdl_conv2d_fwd with
 * fc.weight as a 1x1 kernel. */
int wsdl_plane_relu_minmax(const float* x, float* y, int planes, int hw, wsdl_stream_t stream);

/* ---- optimiser: torch.optim.Adam defaults (TraditionalModel/SegmentationModel.py:91,109-111) -
 * one launch over a flat parameter / gradient buffer; grad_scale folds the 1/world_size of DP.
 * step_dev (optional): the step number as a device int (bias corrections computed in the kernel) - for captured
 * (hipGraph) launches, where a host `step` would be frozen at capture time. */
int wsdl_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                   float beta2, float eps, int step, const int* step_dev, float grad_scale, wsdl_stream_t stream);
/* The same step with EVERY per-step quantity on the device: hyper_dev = {lr, beta1, beta2, eps, grad_scale} (five floats) and
 * the step number step_dev - a learning-rate schedule then changes five floats in device memory, not a kernel argument, so a
 * recorded launch plan (below) stays valid. */
int wsdl_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper_dev, const int* step_dev,
                       wsdl_stream_t stream);

/* ---- refine_pseudo_mask inner step (TraditionalModel/AlternatingDirectionCutLoss.py:736-757) -
 * KL(softmax(X) || S) with log(X+1e-8), reduction 'batchmean', and its gradient wrt softmax(X). */
int wsdl_kl_div_fwd_bwd(const float* xn, const float* s, float* loss, float* dxn, size_t n, int batch,
                        void* ws, size_t ws_bytes, wsdl_stream_t stream);
/* Batched refinement: the same step for N images at once, every per-image scalar kept on the
 * device.  kl_div_per_image: loss[i] = sum_i target*(log target - log(xn+1e-8)) over image i (reduction
 * 'batchmean' of the reference's (1,C,H,W) call), dxn = -target/(xn+1e-8).  refine_combine:
 * out = dkl + lambda*kl_i/(nc_scale*nc_i + 1e-6) * nc_scale * dnc  - the reference's dynamic weight
 * (AlternatingDirectionCutLoss.py:748) without its two host synchronisations per step. */
int wsdl_kl_div_per_image_fwd_bwd(const float* xn, const float* s, float* loss, float* dxn, int N,
                                  size_t per_image, void* ws, size_t ws_bytes, wsdl_stream_t stream);
int wsdl_refine_combine(const float* dkl, const float* dnc, const float* kl, const float* nc, float lambda,
                        float nc_scale, float* out, int N, size_t per_image, wsdl_stream_t stream);
/* softmax over C of (B,C,HW) and its backward */
int wsdl_softmax_fwd(const float* x, float* y, int B, int C, int HW, wsdl_stream_t stream);
int wsdl_softmax_bwd(const float* y, const float* dy, float* dx, int B, int C, int HW, wsdl_stream_t stream);

/* ---- launch plans: the reference's training / CAM loops as one host call per iteration ------------------------------
 * The reference runs its loops statement by statement from Python (one training iteration:
 * TraditionalModel/AlternatingDirectionCutLoss.py:693-703 = SegmentationModel.py:96-113; one CAM batch:
 * TraditionalModel/PsuedoMasks.py:47-62); on this path an iteration is ~520 kernel launches on three streams, 10-14 ms of
 * host time through Python + ctypes against 18.7 ms on the GPU.  A plan records the launches of ANY sequence of the calls of
 * this header once - function, grid, block, LDS bytes, stream and a private copy of every argument value, plus the
 * cross-stream dependencies made through the wsdl_event_* / wsdl_stream_wait_* calls below - while the sequence runs in the
 * ordinary way, and wsdl_plan_replay issues them again from one C loop (csrc/plan.hip).  Same kernels, same arguments,
 * same order per stream: the results are the recorded sequence's, bit for bit.
 *   - one recording at a time PER HOST THREAD (the recording state is thread-local): other threads may call into the
 *     library meanwhile - their launches run normally and are not part of this thread's plan - and may record plans of
 *     their own; wsdl_set_option is refused while any thread records;
 *   - the caller keeps every buffer the sequence touched alive and at its address for the plan's lifetime, and anything
 *     that varies from replay to replay on the device (wsdl_adam_step's step_dev, wsdl_dropout_fwd's counter);
 *   - wsdl_lovasz_softmax_fwd_bwd (rocPRIM launches kernels of its own) poisons a recording: wsdl_plan_end fails;
 *   - wsdl_plan_mark(tag) cuts the plan into segments: wsdl_plan_replay_segment(plan, k) replays segment k (0 .. marks),
 *     so the host can do its own work (a gradient collective) at the places it did while recording. */
int wsdl_plan_begin(void);
int wsdl_plan_recording(void);                       /* 1 between begin and end */
int wsdl_plan_end(void** plan_out);                  /* error (and no plan) if the sequence cannot be replayed */
int wsdl_plan_abort(void);                           /* drop the recording in progress */
int wsdl_plan_mark(long long tag);
int wsdl_plan_pause(void);                           /* host section: what the library is asked to do until wsdl_plan_resume is NOT */
int wsdl_plan_resume(void);                          /* recorded (the host repeats it itself between the segments of a replay)      */
int wsdl_plan_poison(const char* why);               /* the caller did something between begin and end that a replay would miss */
int wsdl_plan_replay(void* plan);
int wsdl_plan_replay_segment(void* plan, int segment);
/* diagnostic twin of wsdl_plan_replay: host microseconds inside the runtime and the count per kind of operation, six
 * entries each (0 kernel launch, 1 memset, 2 stream-waits-for-stream, 3 event record, 4 event wait, 5 mark) */
int wsdl_plan_replay_timed(void* plan, double* us_by_kind, long long* n_by_kind);
int wsdl_plan_stats(void* plan, long long* kernels, long long* memsets, long long* stream_waits, long long* events,
                    long long* marks);
long long wsdl_plan_mark_tag(void* plan, int i);
int wsdl_plan_destroy(void* plan);
/* Stream ordering through the library, so that a plan sees it: events (no timing), "stream waits for event", and
 * "waiter waits for everything enqueued on waited so far".  Outside a recording they are the plain runtime calls. */
int wsdl_event_create(void** ev);
int wsdl_event_destroy(void* ev);
int wsdl_event_record(void* ev, wsdl_stream_t stream);
int wsdl_stream_wait_event(wsdl_stream_t stream, void* ev);
int wsdl_stream_wait_stream(wsdl_stream_t waiter, wsdl_stream_t waited);
/* The few stream-ordered helpers a training step used the tensor library's own kernels for (a plan records launches of
 * THIS library only): memset, *p += delta for a device int32 / int64 (Adam's step number, the dropout call counters),
 * out[i] = a[i] * b[i] (the loss scale of the cross-entropy backward), y = min(x, hi) for int64 labels
 * (torch.clamp(masks, max=1), reference SegmentationModel.py:100). */
int wsdl_memset_async(void* dst, int value, size_t bytes, wsdl_stream_t stream);
int wsdl_add_int(void* p, int is64, long long delta, wsdl_stream_t stream);
int wsdl_mul(const float* a, const float* b, float* out, int n, wsdl_stream_t stream);
int wsdl_clamp_max_i64(const long long* x, long long* y, long long n, long long hi, wsdl_stream_t stream);
/* Weighted loss terms without the tensor library: out[0] = w * mean(x[0..n)) (the "0.1 * ncut" / "0.1 * boundary.mean()" of the
 * reference's combined losses, AlternatingDirectionBoundaryLoss.py:196-200; fixed summation order) and its gradient
 * out[i] = g[0] * c (c = w / n).  The sum of terms is wsdl_add. */
/* Range sentinel of the fp16x2 arithmetic.  wsdl_set_option("range_sentinel", 1) declares that every y_amax / dx_amax handed
 * to wsdl_bn_train_fwd / wsdl_bn_train_bwd points at TWO floats (8-byte aligned): [0] receives max|tensor| as before, [1] the smallest non-zero
 * maximum of any CHANNEL of the tensor (the channel-resident kernels run one workgroup per channel),
 * kept as the bitwise complement of its float bits (a zeroed pair = nothing published).  The convolutions scale a tensor by
 * ONE power of two, so a region 2^E below the tensor's maximum is computed to 2^-(38-E) of its own maximum - past the 1e-3
 * of the parity bar from E ~ 29.  wsdl_range_check reduces npairs such pairs: out[0] = the largest log2(max / min channel
 * maximum), out[1] = the number of pairs beyond limit_log2 (25: the safe range), out[2] = pairs looked at.  `out` may be
 * pinned host memory: the host reads it a step later, without a synchronisation, and selects conv_arith = 2 (the guard). */
int wsdl_range_check(const float* pairs, int npairs, int limit_log2, float* out, wsdl_stream_t stream);
int wsdl_scale_mean(const float* x, int n, float w, float* out, wsdl_stream_t stream);
int wsdl_scale_fill(const float* g, float c, float* out, int n, wsdl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WSDL_HIP_H */
