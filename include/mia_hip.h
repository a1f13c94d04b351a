/* libmia_hip.so -- C ABI of the MI355X (gfx950) UNet-2D training hot path.
 *
 * The reference (trnKhanh/medical-image-analysis) is pure Python on top of torch / torchvision library
 * kernels: it has NO FFI layer of its own for this path.  Each entry point below therefore cites the
 * reference call site whose library kernel(s) it replaces; the Python classes that bind them
 * (medical-image-analysis_amd/{models,losses,transforms,...}) keep the reference's class / argument
 * surface, and INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer unless named host_*; the caller owns all memory
 *     (inputs, outputs, saved statistics, workspaces); the library never allocates or frees
 *   - `stream` is a hipStream_t (0 = null stream); all work is enqueued, nothing synchronises
 *   - activations are NHWC; `dtype` is MIA_F32 or MIA_BF16 and applies to activations and packed
 *     weights; statistics, parameters, gradients of parameters and logits are always fp32
 *   - return 0 on success, negative for argument errors (MIA_EARG / MIA_EUNSUPPORTED), positive =
 *     hipError_t; mia_last_error() returns a thread-local message
 */
#ifndef MIA_HIP_H
#define MIA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIA_F32 0
#define MIA_BF16 1

int mia_version(void);
const char* mia_last_error(void);

/* ------------------------------------------------------------------ weights / layout */
/* fp32 parameter [D0][D1][taps] -> packed [taps][npad][kpad] (zero padded) in `dtype`.
 * Replaces the cuDNN/MIOpen filter transform behind nn.Conv2d / nn.ConvTranspose2d
 * (src/models/unet/blocks.py:83-90, unet.py:142).  n_from_d0: n indexes D0 (else D1). */
int mia_pack_weight(const float* src, void* dst, int dtype, int d0, int d1, int taps, int npad, int kpad,
                    int n_from_d0, void* stream);
/* The same for every weight of a model in ONE launch (the optimizer rewrites all parameters each step).  descs_dev: device
 * array of `count` records of mia_pack_desc_bytes() bytes each:
 *   { const float* src; void* dst; int d0, d1, taps, npad, kpad, n_from_d0; int brick_begin, bricks_x; }
 * a brick is 16 x 64 (n_from_d0) or 64 x 16 (D0 x D1) source elements; brick_begin = running brick count, bricks_x =
 * bricks along D1; total_bricks = sum over records; bricks cover the PADDED extents. */
int mia_pack_desc_bytes(void);
int mia_pack_weight_batch(const void* descs_dev, int count, int total_bricks, int max_taps, int dtype, void* stream);
/* generic strided (n, c, p) copy with dtype conversion: NCHW <-> NHWC, fp32 <-> bf16
 * (replaces .to(dtype)/.contiguous() at al_trainer.py:1366-1368). */
int mia_relayout(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int64_t hw,
                 int64_t ssn, int64_t ssc, int64_t ssp, int64_t dsn, int64_t dsc, int64_t dsp, void* stream);
/* out = a + b: the residual connection of ResidualBlock (blocks.py:164) */
int mia_add(const void* a, const void* b, void* out, int dtype, int64_t n, void* stream);
/* out[c] (+)= sum over p rows of x[p][c]: bias gradients of Conv2d / ConvTranspose2d. */
/* Dropout2d channel masks (blocks.py:92-96): out[i] = 1/keep with probability keep, else 0 (Philox4x32-10 keyed by seed, offset) */
int mia_dropout_mask(float* out, int64_t n, float keep, uint64_t seed, uint64_t offset, void* stream);
/* the same with seed and base offset read from device memory (seed_base[0], seed_base[1]; the launch uses offset seed_base[1] +
 * rel_offset): for a train step replayed from a captured hipGraph, whose kernel arguments are frozen (training/engine.py graph mode) */
int mia_dropout_mask_dyn(float* out, int64_t n, float keep, const uint64_t* seed_base, uint64_t rel_offset, void* stream);
/* optimizer.zero_grad() (al_trainer.py:1375) on a flat buffer: bytes % 16 == 0, 16-byte aligned */
int mia_zero(void* p, int64_t bytes, void* stream);
/* dst[i] = src[i * stride], fp32 (per-channel sums out of interleaved statistics) */
int mia_gather_f32(const float* src, int64_t stride, float* dst, int n, void* stream);
int mia_colsum_workspace(int64_t p, int c); /* floats */
int mia_colsum(const void* x, int dtype, int64_t p, int c, float* workspace, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------ dense conv products (MFMA) */
#define MIA_CONV_G3S1 0 /* gather 3x3 s1 p1: Conv2d fwd (blocks.py:83-90) and its input gradient (flip_taps=1) */
#define MIA_CONV_G3S2 1 /* gather 3x3 s2 p1: strided Conv2d fwd (unet.py:58-66, first block of each level) */
#define MIA_CONV_G2S2 2 /* gather 2x2 s2 p0: ConvTranspose2d input gradient (unet.py:142) */
#define MIA_CONV_T3S2 3 /* transposed 3x3 s2: strided Conv2d input gradient */
#define MIA_CONV_T2S2 4 /* transposed 2x2 s2: ConvTranspose2d fwd (unet.py:142, :212) */
#define MIA_CONV_G1 5   /* 1x1: ResidualBlock skip conv (blocks.py:147-153) */
/* Library options: kernel-selection / launch-shape knobs for A/B measurements (process-wide, not part of any reference
 * interface; results never depend on one beyond fp32 summation order).  ONE table (csrc/options.h), each entry read from its
 * environment variable once at first use, stored in an atomic, changed only by mia_set_option; every entry point takes one
 * snapshot per call.  Thread-safe.  Names (env var, default):
 *   conv_bt (MIA_CONV_BT, 1)       wide stride-1 3x3 bf16 convs on the 512-thread big-tile LDS-DMA kernel (0: tile kernel)
 *   conv_bt_order (MIA_CONV_BT_ORDER, 1)   its work-item order: 1 = the channel blocks of a pixel tile run together on one XCD
 *   conv64 (MIA_CONV64, 1)         64 -> 64 channel bf16 3x3 stride-1 launches on the persistent register-weight kernel
 *   conv64_dma (MIA_CONV64_DMA, 1) the 512-thread LDS-DMA 64-channel kernel: 1 = the 64 -> (64 | 64) two-destination input
 *                                  gradient in one pass (measured 7 % faster than two conv64 passes), 2 = also plain
 *                                  64 -> 64 launches (measured 6-11 % slower than conv64_persist_kernel), 0 = never
 *   conv_s2_wide (MIA_CONV_S2_WIDE, 1)   stride-2 3x3 bf16 forward with 128-multiples of output channels on 512-thread
 *                                  workgroups / 16-row tiles: 1 = from 128 input channels on, 2 = always, 0 = never
 *   conv_pw (MIA_CONV_PW, 1)       ConvTranspose 2x2 / stride 2 forward and input gradient (bf16) as one pointwise GEMM on the
 *                                  512-thread LDS-DMA ring (0: the tile kernel, one parity class per workgroup)
 *   conv_pw_s2 (MIA_CONV_PW_S2, 1) stride-2 3x3 bf16 forward as a tap-gathered GEMM on the same ring (needs conv_pw; 1: up to 256
 *                                  input channels, 2: always, 0: tile kernels -- default: -0.16 ms per step in isolation, none inside the step)
 *   conv_xcd / wgrad_xcd (MIA_CONV_XCD / MIA_WGRAD_XCD, 1)   blocks sharing an input tile run on one XCD (0: plain grid order)
 *   wgrad_bt (MIA_WGRAD_BT, 1)     bf16 3x3 stride-1 weight gradients with >= 128 output channels on the 512-thread 128 x 64 block kernel
 *   wgrad_t2 (MIA_WGRAD_T2, 1)     ConvTranspose 2x2 weight gradient (bf16, >= 128 coarse channels) on the 512-thread three-stage ring
 *   wgrad_dma (MIA_WGRAD_DMA, 1)   bf16 3x3 stride-1 weight gradients on the LDS-DMA ring kernel; 0: the register-staged
 *                                  two-workgroups-per-CU kernel
 *   stream_blocks (MIA_STREAM_BLOCKS, 32768)   target block count of the norm / activation streaming passes
 *   stem_mfma (MIA_STEM_MFMA, 1)   matrix-core stem kernel for fp32 images
 *   f32_split (MIA_F32_SPLIT, 2)      fp32 convs / weight gradients of the branch-free tile kernels on the f16 matrix cores: every
 *                                  operand element, scaled by a per-tensor power of two taken from the tensor's max |x| (the amax_*
 *                                  arguments below), enters as h + l (two fp16: 22-23 significand bits), four exact products (value 1) or, in
 *                                  the convs with >= 32 output channels under value 2, the three that matter (h H + h L + l H on
 *                                  32 x 32 tiles; l L <= 2^-24 of the product), fp32 accumulation, exact rescaling.  Accuracy of the exact fp32 MFMA kernels (not bit-identical to
 *                                  them), ~2x on the fp32 training step.  Tensors stay fp32.  A call without the maxima, or value 0,
 *                                  runs the exact fp32 MFMA kernels.
 *   reserve_cus (MIA_RESERVE_CUS, 0)   CUs the persistent kernels leave free (0..64, rounded so that the grids stay
 *                                  multiples of 8): the grids of conv_bt / conv_pw / conv64 / conv64_dma shrink to CUs - k
 *                                  workgroups (same work items: bit-identical results) and mia_wgrad_target_blocks follows
 *                                  (different split-K count: deterministic, not bit-identical across settings).  Set by
 *                                  the data-parallel trainer (training/engine.py) so that RCCL's ring kernels find CUs
 *                                  while a persistent kernel runs (all-reduce overlapped with backward)
 * (16 options.  The measured-and-rejected experiments of rounds 3-4 -- Winograd conv64, conv_pw T3S2, conv_t3_wide, wgrad_narrow, conv_mt8, the
 * column-reduce epilogue -- are no longer in the shipping library: records under profiles/, entry points of the ones a probe still builds in
 * include/mia_hip_experiments.h, compiled with -DMIA_EXPERIMENTS.)
 * Do not change wgrad_* between mia_wgrad_geometry and the mia_conv_wgrad it sizes.  The Python loader applies
 * MIA_OPTIONS="name=value,..." through mia_set_option.  Unknown name: MIA_EARG. */
int mia_set_option(const char* name, int value);
int mia_get_option(const char* name, int* value);
/* out[p][n] = bias[n] + sum_taps sum_k in[p*s + tap - pad][k] * wpack[tap][n][k].
 * in1|in2 are concatenated along channels (c1 + c2) -- this is how torch.cat([skip, up], 1)
 * (unet.py:213) is eliminated; out1|out2 are split along channels (o1 + o2) for its gradient.
 * stat_partials (optional): [N][tiles][o1+o2][2] per-tile sum / sum-of-squares of the output
 * (feeds mia_norm_finalize; replaces the statistics pass of InstanceNorm2d/BatchNorm2d, blocks.py:98).
 * tiles = tiles_y * tiles_x from mia_conv_mma_tiles(). */
int mia_conv_mma_tiles(int mode, int hout, int wout, int* tiles_y, int* tiles_x, int* tile_h);
/* amax_in1 / amax_in2 / amax_w (optional, fp32 tensors only): device pointers to max |x| of in1 / in2 / the weight tensor as fp32
 * bit patterns (mia_amax / mia_amax_batch).  All present (amax_in2 only when c2 > 0) and option f32_split on: the products run on the
 * f16 matrix cores from two-part split operands (see f32_split above); any of them NULL: exact fp32 MFMAs.
 * amax_out1 / amax_out2 (optional, fp32): ZEROED 4-byte slots that receive max |out1| / max |out2| as a by-product (epilogue of the
 * tile kernel, a separate pass for generic shapes) -- for outputs another conv consumes directly (unet.py:142 -> :213 and back). */
int mia_conv_mma(int mode, int dtype, const void* in1, int c1, const void* in2, int c2, const void* wpack, int npad,
                 int kpad, int flip_taps, const float* bias, void* out1, int o1, void* out2, int o2,
                 float* stat_partials, int n, int hin, int win, int hout, int wout, const void* amax_in1, const void* amax_in2,
                 const void* amax_w, const void* wpack_split, void* amax_out1, void* amax_out2, void* stream);
/* wpack_split (optional, with the maxima): the packed weights already split into (h | l << 16) fp16 words of w * 2^e (e from amax_w's
 * slot) by mia_split_f16_batch -- same [tap][npad][kpad] layout, 4 bytes per weight; the split kernels then stage them as they are instead
 * of splitting them once per tile.  Device table of `count` {const float* src; void* dst; int64_t n; const void* amax;} records
 * (mia_split_desc_bytes() each; n a multiple of 4, 16-byte aligned buffers). */
int mia_split_desc_bytes(void);
int mia_split_f16_batch(const void* descs_dev, int count, void* stream);
/* max |x| of an fp32 tensor of n elements, folded (atomic unsigned maximum of the fp32 bit pattern: order-independent, deterministic)
 * into *slot; reset != 0 zeroes the slot first (stream-ordered, by a kernel: hipMemsetAsync nodes on graph-pool memory were seen to
 * replay wrongly inside a captured step).  The batched form takes a device table of `count`
 * {const float* src; int64_t n;} records (mia_amax_desc_bytes() each) and writes slots[0 .. count): every weight of a model in one
 * launch after the optimizer step.  These maxima are the scale source of the split-f16 products (no reference counterpart: the
 * reference multiplies in fp32, blocks.py:83-90). */
int mia_amax(const float* x, int64_t n, void* slot, int reset, void* stream);
/* dst[i] = src[i], uint8 -> int64: label maps travel host -> device as bytes (1/8 of the int64 tensor `label.to(device)` moves,
 * al_trainer.py:1368) and are widened here for the loss kernels; src 8-byte, dst 16-byte aligned (training/feed.py). */
int mia_widen_u8_i64(const void* src, void* dst, int64_t n, void* stream);
/* HOST helper of the same feed (no device work): dst[i] = (uint8) src[i] over host memory in one multi-threaded pass; returns 1 when
 * every label lies in 0 .. 255 (dst valid), 0 when some label does not fit (ship the int64 tensor instead), < 0 on bad arguments.
 * threads <= 0: 8. */
int mia_host_narrow_labels(const int64_t* src, uint8_t* dst, int64_t n, int threads);
int mia_host_copy(void* dst, const void* src, int64_t bytes, int threads); /* pageable -> pinned memcpy on `threads` (<= 0: 4) plain threads */
int mia_amax_desc_bytes(void);
int mia_amax_batch(const void* descs_dev, int count, void* slots, void* stream);

/* Fused PlainBlock, consumer side ("normalise-on-load"; SURVEY 8b export list: conv3x3 with "optional fused normalise +
 * LeakyReLU on load taking per-(n,c) scale/shift").  Replaces, for a PlainBlock whose only consumer is the next block's conv
 * (the two blocks of one encoder / decoder level, unet.py:54-76, 157-173), the producer's InstanceNorm/BatchNorm + LeakyReLU
 * pass (blocks.py:98-102): y_in is the producer's RAW conv output [N][H][W][c1] and the conv stages
 * x = lrelu(in_scale[n][c] * y + in_shift[n][c]) (fp32 fma, select, round to nearest even: bit for bit what
 * mia_norm_act_fwd would have written), zero padding applied to x.  in_scale / in_shift = the `scale` / `shift` rows that
 * mia_norm_finalize wrote ([N][c1], Dropout2d multipliers folded in).  One source, one destination, forward taps.
 * mia_conv_nl_supported: 1 if a kernel serves the shape (today: 3x3 stride 1, bf16, 64 -> 64 channels, H > 8), else 0 --
 * the caller then materialises the activation with mia_norm_act_fwd and uses mia_conv_mma. */
int mia_conv_nl_supported(int mode, int dtype, int c1, int nout, int hout, int wout);
int mia_conv_mma_nl(int mode, int dtype, const void* y_in, int c1, const float* in_scale, const float* in_shift, float slope,
                    const void* wpack, int npad, int kpad, const float* bias, void* out, int nout, float* stat_partials,
                    int n, int hin, int win, int hout, int wout, void* stream);

/* Fused PlainBlock, backward: the input-gradient conv of the CONSUMING block with the producing block's norm-backward reduction
 * in its epilogue.  out = dz, the gradient w.r.t. the producing block's activated output (what mia_conv_mma computes with
 * flip_taps = 1); y_prod / scale / shift / xa / xb = that block's raw conv output and coefficient rows (mia_norm_finalize).
 * While a tile's accumulators are in registers the kernel adds up g = dz * lrelu'(scale * y + shift) and g * (xa * y + xb) from
 * the bf16-rounded dz it stores, and writes partials [N][tiles][nout][2] (tiles as mia_conv_mma_tiles) -- the input of
 * mia_norm_act_bwd_pre, which then skips its own reduction pass over dz and y (blocks.py:98-102 backward).  3x3 stride 1, bf16,
 * 64 -> 64 channels (mia_conv_cr_supported). */
/* mia_conv_mma with out += result (no bias, one source, one destination; the tile kernel: returns MIA_EUNSUPPORTED outside its
 * contract).  Use: a skip tensor has two consumers (unet.py:213 and the next encoder level), so its gradient arrives in two pieces;
 * the piece computed second -- the stride-2 conv's input gradient, mode CONV_T3S2 -- is added into the first in this launch's
 * epilogue (read-modify-write), and the skip block's norm backward reads one gradient tensor instead of two. */
int mia_conv_acc_supported(int mode, int dtype, int c1, int nout);
int mia_conv_mma_acc(int mode, int dtype, const void* in1, int c1, const void* wpack, int npad, int kpad, int flip_taps,
                     void* out_inout, int nout, int n, int hin, int win, int hout, int wout, const void* amax_in, const void* amax_w,
                     const void* wpack_split, void* stream);


/* Stem: Conv2d(1, C0, 3, padding=1) (first encoder block, unet.py:54-66 with input_channels=1): HBM-streaming VALU
 * kernels (9 FMAs per output; MFMA would idle 31/32 of its K).  x is the [N][H][W] image in x_dtype (fp32 or bf16),
 * y / dy are NHWC in `dtype`; stat_partials [N][mia_stem_slabs()][C0][2]; grad is the [C0][1][3][3] parameter layout. */
int mia_stem_slabs(void);
int mia_stem_wgrad_workspace(int c0); /* floats */
int mia_stem_fwd(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype, float* stat_partials, int n,
                 int h, int wd, int c0, void* stream);
int mia_stem_wgrad(const void* x, int x_dtype, const void* dy, int dtype, float* workspace, float* grad, int n, int h, int wd,
                   int c0, int accumulate, void* stream);
/* mia_stem_wgrad with the block's norm + LeakyReLU backward folded in: dz is the gradient w.r.t. the stem block's ACTIVATED
 * output, y its raw conv output, and the kernel forms dy = scale * (g - c1 - xhat * c2), g = dz * lrelu'(scale * y + shift), on
 * load (rounded to `dtype` like the stored dy of mia_norm_act_bwd: same bits).  The stem has no input gradient, so this weight
 * gradient is the only consumer of dy: pair it with mia_norm_bwd_sums and the backward apply pass (read dz + y, write dy) and the
 * read of dy disappear.  scale / shift / xa / xb: mia_norm_finalize's rows; c1 / c2: mia_norm_bwd_sums' outputs. */
int mia_stem_wgrad_fused(const void* x, int x_dtype, const void* dz, const void* y, int dtype, const float* scale,
                         const float* shift, const float* xa, const float* xb, const float* c1, const float* c2, float slope,
                         float* workspace, float* grad, int n, int h, int wd, int c0, int accumulate, void* stream);

#define MIA_WGRAD_3S1 0
#define MIA_WGRAD_3S2 1
#define MIA_WGRAD_2S2 2 /* ConvTranspose2d: x := grad_output (fine grid), dy := layer input (coarse grid) */
/* slabs[z][tap][npad][kpad] = sum over the z-th share of output pixels of dy[p][n] * x[p*s+tap-pad][k]
 * (autograd weight gradient of the layers above); mia_wgrad_reduce sums the ksplit slabs in a fixed
 * order into grad[nn][kk][taps] (= OIHW for Conv2d, [Cin][Cout][2][2] for ConvTranspose2d). */
int mia_wgrad_target_blocks(int mode, int dtype); /* split-K workgroups to aim for (ksplit = target / (npad/64 * kpad/64)) */
int mia_wgrad_geometry(int mode, int dtype, int hy, int wy, int* tiles_y, int* tiles_x);
/* What the caller needs to size ksplit for THIS shape: the column blocks of one split-K slice, the workgroup count to aim for and the
 * tile height (rows of 16 output pixels per split-K step) of the kernel mia_conv_wgrad will pick -- 64-wide channel blocks cut per
 * source, or the 96-wide blocks of 3x3 stride-1 bf16 layers whose channel counts are multiples of 96 and not of 64 (768 threads, one
 * block where the 64-wide form needs 2 x 2).  Supersedes the two queries above for callers that know the channel counts. */
int mia_wgrad_plan(int mode, int dtype, int c1, int c2, int cdy, int npad, int hy, int* column_blocks, int* target_blocks, int* tile_h);
int mia_conv_wgrad(int mode, int dtype, const void* x1, int c1, const void* x2, int c2, const void* dy, int cdy,
                   float* slabs, int ksplit, int npad, int kpad, int n, int hx, int wx, int hy, int wy, const void* amax_x1,
                   const void* amax_x2, const void* amax_dy, void* stream); /* amax_*: as for mia_conv_mma */
/* mia_conv_wgrad with normalise-on-load of x (backward half of the fused PlainBlock, see mia_conv_mma_nl): y_in is the raw
 * conv output of the PRODUCING block, x = lrelu(in_scale[n][k] * y + in_shift[n][k]) is formed while the tile is staged (zero
 * outside the image), so the activation the weight gradient of blocks.py:83-90 contracts with is never read from memory.
 * 3x3 stride 1, bf16, one source; slabs / ksplit / reduce exactly as for mia_conv_wgrad. */
int mia_wgrad_nl_supported(int mode, int dtype, int c1, int cdy);
int mia_conv_wgrad_nl(int mode, int dtype, const void* y_in, int c1, const float* in_scale, const float* in_shift, float slope,
                      const void* dy, int cdy, float* slabs, int ksplit, int npad, int kpad, int n, int hx, int wx, int hy,
                      int wy, void* stream);
int mia_wgrad_reduce(const float* slabs, int ksplit, int taps, int npad, int kpad, float* grad, int nn, int kk,
                     int accumulate, void* stream);

/* ------------------------------------------------------------------ Dropout2d + norm + LeakyReLU */
#define MIA_NORM_INSTANCE 0
#define MIA_NORM_BATCH 1
/* per-(n,c) coefficients from conv-epilogue partials: xhat = xa*y + xb, z = lrelu(scale*y + shift).
 * drop_scale [N][C] (0 or 1/(1-p)) folds nn.Dropout2d (blocks.py:92-96); batch mode updates
 * running_mean/var (momentum, unbiased var) and num_batches_tracked like nn.BatchNorm2d. */
int mia_norm_finalize(const float* partials, int n, int tiles, int c, int64_t hw, int mode, int training,
                      const float* drop_scale, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, long long* num_batches, float* xa, float* xb,
                      float* scale, float* shift, float* ysum, void* stream);
/* stand-alone statistics partials [N][slabs][C][2] when no conv epilogue produced them */
int mia_norm_stats(const void* y, int dtype, int n, int64_t hw, int c, int slabs, float* partials, void* stream);
/* amax_out (nullable; here and in the backward forms below): 4-byte device slot, ZEROED by the caller, into which max |output| is
 * folded as an fp32 bit pattern during the same pass (atomic unsigned maximum; fp32 tensors, ignored for bf16) -- the scale source
 * of the split-f16 convs that consume the output (mia_conv_mma amax_*), so no separate mia_amax pass over it is needed. */
int mia_norm_act_fwd(const void* y, void* z, int dtype, const float* scale, const float* shift, int n, int64_t hw, int c,
                     float slope, void* amax_out, void* stream);
/* dz2 (nullable, here and in the _reduce / _apply_sync forms): a second piece of the output gradient, summed on load --
 * the two consumers of a skip tensor (unet.py:213 and the next encoder level) each deliver one; needs c % 32 == 0
 * (mia_norm_two_piece_ok). */
int mia_norm_two_piece_ok(int dtype, int c);
/* backward of (dropout . norm . lrelu): dy, dgamma, dbeta.  partials: [N][slabs][C][2]; c1, c2: [N][C].
 * dbias (optional, with ysum [N][C] from mia_norm_finalize) = gradient of the Conv2d bias in front of the norm
 * (= sum over pixels of dy) in closed form from the reduction sums: no extra pass over dy. */
int mia_norm_act_bwd(const void* dz, const void* dz2, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                     const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                     int fixed_stats, float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma,
                     float* dbeta, float* dbias, int accumulate, void* amax_out, void* stream);
/* mia_norm_act_bwd without its apply pass (no dy is written): reduction + finalize only -- c1 / c2 (group means of g and
 * g * xhat), dgamma, dbeta, dbias.  For a block whose only consumer of dy forms it on load (mia_stem_wgrad_fused). */
int mia_norm_bwd_sums(const void* dz, const void* dz2, const void* y, int dtype, const float* scale, const float* shift,
                      const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                      int fixed_stats, float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma,
                      float* dbeta, float* dbias, int accumulate, void* stream);
/* mia_norm_act_bwd with the reduction already done: partials [n][parts][c][2] come from mia_conv_mma_cr.  Sums + finalize; the
 * apply pass (dy = ...) runs only when dy != NULL (NULL: the consumer forms dy on load, mia_stem_wgrad_fused). */

/* Synchronised batch norm for data-parallel runs (build-side addition; SURVEY.md 8e: "a second, small collective"):
 * the caller moves 3*C floats (forward, all-gather) and 2*C floats (backward, all-reduce sum) per layer over RCCL and
 * these entry points do the device work on either side of it, so N ranks x bs reproduce one process at N*bs
 * (BatchNorm2d batch statistics, blocks.py:98).
 *   mia_bn_sync_local_stats : conv-epilogue partials -> per-(n,c) sums parked in xa/xb + local[3][C] = mean, M2, count
 *   mia_norm_finalize_sync  : gathered[world][3][C] -> coefficients / running statistics as mia_norm_finalize(training)
 *   mia_norm_act_bwd_reduce : per-(n,c) sums in c1/c2 + tot[3][C] = local (sum g, sum g*xhat, pixel count)
 *   mia_norm_act_bwd_apply_sync : group_tot[3][C] = the all-reduced (summed) totals; dgamma, dbeta and dbias stay
 *                             LOCAL sums (the gradient all-reduce adds them up). */
int mia_bn_sync_local_stats(const float* partials, int n, int tiles, int c, int64_t hw, const float* drop_scale, float* xa,
                            float* xb, float* local, void* stream);
int mia_norm_finalize_sync(const float* gathered, int world, int n, int c, int64_t hw, const float* drop_scale,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, long long* num_batches, float* xa, float* xb, float* scale, float* shift,
                           float* ysum, void* stream);
int mia_norm_act_bwd_reduce(const void* dz, const void* dz2, const void* y, int dtype, const float* scale, const float* shift, const float* xa,
                            const float* xb, int n, int64_t hw, int c, float slope, int slabs, float* partials, float* c1,
                            float* c2, float* tot, void* stream);
int mia_norm_act_bwd_apply_sync(const void* dz, const void* dz2, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                                const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, float slope,
                                float* c1, float* c2, const float* group_tot, float* dgamma, float* dbeta, float* dbias,
                                int accumulate, void* amax_out, void* stream);

/* ------------------------------------------------------------------ 1x1 head + Dice/CE loss */
/* seg_output = Conv2d(c0, K1, 1) (unet.py:176), K1 <= 8.  Logits are fp32 with element strides (osn, osk, osp). */
int mia_head_fwd(const void* x, int dtype, const float* w, const float* b, float* logits, int n, int64_t hw, int c0, int k1,
                 int64_t osn, int64_t osk, int64_t osp, void* stream);
int mia_head_bwd_workspace(int c0, int k1); /* floats */
int mia_head_bwd(const float* dlogits, const void* x, int dtype, const float* w, void* dx, float* dw, float* db,
                 float* workspace, int n, int64_t hw, int c0, int k1, int64_t gsn, int64_t gsk, int64_t gsp, int accumulate,
                 void* stream);
#define MIA_LOSS_SOFTMAX 1
#define MIA_LOSS_DO_BG 2
#define MIA_LOSS_BATCH 4
#define MIA_LOSS_SQUARED 8
#define MIA_LOSS_DENSE 16 /* `labels` is a dense fp32 target [B][K1][HW] (already one-hot / class probabilities) instead of int64
                           * indices: DiceLoss skips its one-hot encoder when shapes match (dice_loss.py:40-41) and
                           * torch.nn.CrossEntropyLoss treats such a target as probabilities.  Pass the float pointer cast to
                           * `const long long*`; same strides as the logits (sn, sk, sp). */
/* out[0] = ce_w*CE + dice_w*Dice, out[1] = CE (mean over pixels), out[2] = Dice
 * (DiceLoss.forward dice_loss.py:32-76, DiceAndCELoss.forward compound_losses.py:33-49).
 * sums [B][K1][3] = (I, sum p, sum t); coef [B][K1][2] feeds the backward; bad_label = int[2], zero before the first call: [1] is SET (never
 * cleared by the library: sticky until the caller zeroes it) by a call that met a label outside [0,K1) (loss and coef of THAT call are
 * NaN), [0] = scratch re-armed by every call. */
int mia_dice_ce_workspace(int nb, int k1, int slabs); /* floats */
int mia_dice_ce_fwd(const float* logits, const long long* labels, int nb, int64_t hw, int k1, int64_t sn, int64_t sk,
                    int64_t sp, int flags, float smooth, float dice_w, float ce_w, int slabs, float* workspace, float* sums,
                    float* coef, float* out, int* bad_label, void* stream);
int mia_dice_ce_bwd(const float* logits, const long long* labels, const float* coef, const float* grad_out, float* dlogits,
                    int nb, int64_t hw, int k1, int64_t sn, int64_t sk, int64_t sp, int64_t gsn, int64_t gsk, int64_t gsp,
                    int flags, float dice_w, float ce_w, void* stream);

/* Head fused with the last decoder block (unet.py:176 behind blocks.py:98-100): the block's activation
 * z = lrelu(scale*y + shift) has one consumer, the 1x1 head, so it is never materialised -- the head recomputes it from the
 * raw conv output y on load, and the block's norm backward recomputes dz = W^T dlogits instead of reading it.
 * mia_head_norm_eligible: 0 or the units per pixel; contract c0 in {4,8,12,16} 16-byte units (12 = 96 bf16 channels, on sixteen-lane groups), 2 <= k1 <= 4. */
int mia_head_norm_eligible(int dtype, int n, int64_t hw, int c0, int k1);
int mia_head_norm_fwd(const void* y, int dtype, const float* scale, const float* shift, float slope, const float* w,
                      const float* b, float* logits, int n, int64_t hw, int c0, int k1, int64_t osn, int64_t osk, int64_t osp,
                      void* stream);
int mia_head_norm_wgrad(const float* dlogits, const void* y, int dtype, const float* scale, const float* shift, float slope,
                        float* dw, float* db, float* workspace, int n, int64_t hw, int c0, int k1, int64_t gsn, int64_t gsk,
                        int64_t gsp, int accumulate, void* stream);
int mia_norm_act_bwd_head(const float* dlogits, const float* w, int k1, int64_t gsn, int64_t gsk, int64_t gsp, const void* y,
                          void* dy, int dtype, const float* scale, const float* shift, const float* xa, const float* xb,
                          const float* ysum, int n, int64_t hw, int c, int mode, int fixed_stats, float slope, int slabs,
                          float* partials, float* c1, float* c2, float* dgamma, float* dbeta, float* dbias, int accumulate,
                          void* amax_out, void* stream);
/* mia_norm_act_bwd_head + mia_head_norm_wgrad in ONE reduction pass (both read exactly dlogits and y): the kernel that adds up the
 * block's norm-backward sums also accumulates the head's dW[k][c] = sum_p dl[p][k] * lrelu(scale * y + shift) and db[k] (unet.py:176
 * backward).  c == 64 (mia_head_w_supported); head_workspace: n * slabs * k1 * (c + 1) floats. */
int mia_head_w_supported(int dtype, int c, int k1);
int mia_norm_act_bwd_head_w(const float* dlogits, const float* w, int k1, int64_t gsn, int64_t gsk, int64_t gsp,
                            const void* y, void* dy, int dtype, const float* scale, const float* shift, const float* xa,
                            const float* xb, const float* ysum, int n, int64_t hw, int c, int mode, int fixed_stats,
                            float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma, float* dbeta,
                            float* dbias, int accumulate, float* head_workspace, float* dw_head, float* db_head,
                            int accumulate_head, void* amax_out, void* stream);

/* ------------------------------------------------------------------ optimizer (al_trainer.py:1374-1379) */
#define MIA_OPT_ADAM 0
#define MIA_OPT_ADAMW 1
#define MIA_OPT_SGD 2
int mia_grad_norm_workspace(void); /* floats */
/* out[0] = L2 norm of grad_scale*grad, out[1] = min(1, max_norm/(norm+1e-6)) (torch.nn.utils.clip_grad_norm_);
 * grad_scale = 1/world_size when `grad` holds the all-reduced SUM of per-rank gradients */
int mia_grad_norm(const float* grad, int64_t n, float max_norm, float grad_scale, float* workspace, float* out, void* stream);
int mia_scale_by_clip(float* x, int64_t n, const float* clip, void* stream);
int mia_optim_step(float* param, const float* grad, float* m, float* v, int64_t n, int kind, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, int first_step,
                   const float* clip, float grad_scale, void* stream);
/* mia_optim_step with the per-step scalars read from device memory: dyn = {lr, bias_corr1, bias_corr2, first_step != 0} (fp32[4]),
 * rewritten by the host before every replay of a captured step (PolyLRScheduler, al_trainer.py:1366-1379) */
/* writes dyn (32 bytes, 16-byte aligned: fp32 {lr, bias_corr1, bias_corr2, first_step}, u64 {Philox seed, base offset}) from kernel
 * ARGUMENTS -- safe however far the host runs ahead of the device; the u64 half is what mia_dropout_mask_dyn reads */
int mia_step_dyn_set(void* dyn32, float lr, float bias_corr1, float bias_corr2, int first_step, uint64_t seed, uint64_t offset, void* stream);
int mia_optim_step_dyn(float* param, const float* grad, float* m, float* v, int64_t n, int kind, float beta1, float beta2, float eps,
                       float weight_decay, const float* dyn, const float* clip, float grad_scale, void* stream);

/* ------------------------------------------------------------------ augmentation / resize / normalisation (src/transforms) */
/* images [B][C][H][W] fp32, labels [B][H][W] int64, per-sample parameter arrays on the device;
 * apply[b] == 0 (apply may be NULL = all) copies sample b through unchanged. */
/* RandomAffine / RandomRotation (joint_transform.py:158-206, :100-127): torchvision F.affine / F.rotate tensor path
 * = inverse matrix mats[b][6] -> base grid -> grid_sample(nearest, zeros, align_corners=False); image and label in one launch. */
int mia_affine_nearest(const float* img_in, float* img_out, const long long* lab_in, long long* lab_out, int nb, int c, int h,
                       int w, const float* mats, const int* apply, void* stream);
/* Elastic deformation (north_star; NO reference counterpart -- SURVEY 0 row 2 -- own spec, see csrc/augment.hip): displacement
 * vectors disp[B][2][gh][gw] (pixels; component 0 = x, 1 = y) on a coarse control grid spanning the image corner to corner,
 * bilinearly interpolated per pixel; image sampled bilinearly with zero padding, label at the nearest source pixel. */
int mia_elastic_warp(const float* img_in, float* img_out, const long long* lab_in, long long* lab_out, int nb, int c, int h,
                     int w, const float* disp, int gh, int gw, const int* apply, void* stream);
/* RandomRotation90 / MirrorTransform (joint_transform.py:40-97): torch.rot90(k) then optional flips; 4- or 8-byte elements */
int mia_rot90_flip(const void* in, void* out, int elem_bytes, int nb, int c, int h, int w, int k, int flip_h, int flip_w,
                   void* stream);
/* RandomCrop2D (joint_transform.py:130-155): F.crop with per-sample window origins top[b], left[b] (device int arrays);
 * the window lies inside the image (T.RandomCrop.get_params draws it so); 4- or 8-byte elements */
int mia_crop(const void* in, void* out, int elem_bytes, int nb, int c, int h, int w, int oh, int ow, const int* top,
             const int* left, void* stream);
/* RandomGaussianBlur (image_transform.py:145-193): F.gaussian_blur = k x k outer-product kernel, reflect pad */
int mia_gaussian_blur(const float* in, float* out, int nb, int c, int h, int w, const float* sigma, const int* ksize,
                      int max_ksize, const int* apply, void* stream);
/* per-sample (mean, unbiased std) over C*H*W (gray=1, C=3: of the luma image): feeds contrast and z-score */
int mia_sample_stats_workspace(int nb); /* floats */
int mia_sample_stats(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std, void* stream);
/* Selected-sample forms for the batched pipeline (al_trainer.py:670-697: every stage is drawn per sample with p = 0.1 .. 0.2).
 * In every augmentation entry point `apply[b]` > 0 transforms sample b, 0 copies it through, < 0 SKIPS it (not read, not
 * written).  mia_sample_stats_sel: statistics of the samples with apply[b] >= 0 only.  mia_copy_selected: out[b] = in[b] for
 * apply[b] > 0 (bytes_per_sample % 16 == 0) -- the copy-back of a neighbourhood stage run into a scratch buffer. */
int mia_sample_stats_sel(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std,
                         const int* apply, void* stream);
int mia_copy_selected(const void* in, void* out, int64_t bytes_per_sample, int nb, const int* apply, void* stream);
#define MIA_EW_GAMMA 0    /* RandomGamma            image_transform.py:31 */
#define MIA_EW_CONTRAST 1 /* RandomContrast / RandomBrightness (both ColorJitter(contrast=)) image_transform.py:62,:93 */
#define MIA_EW_NOISE 2    /* RandomGaussianNoise with an explicit noise tensor image_transform.py:130-132 */
#define MIA_EW_ZSCORE 3   /* ZScoreNormalize        normalization.py:17-21 */
int mia_elementwise(const float* in, float* out, int64_t per_sample, int nb, int op, const float* p0, const float* mean_std,
                    const float* aux, const int* apply, void* stream);
/* RandomGaussianNoise with on-device Philox4x32-10 normal noise (sigma[b]) */
int mia_noise_clip(const float* in, float* out, int64_t per_sample, int nb, const float* sigma, uint64_t seed, uint64_t offset,
                   const int* apply, void* stream);
/* JointResize image path / UnetProcessor.preprocess / deep-supervision Upsample: interpolate(bilinear, align_corners=False);
 * lowres_hw[b][2] != NULL = SimulateLowRes (image_transform.py:218-225: nearest-exact down, bilinear up) fused in one pass */
int mia_resize_bilinear(const float* in, float* out, int nb, int c, int h, int w, int oh, int ow, const int* lowres_hw,
                        const int* apply, void* stream);
int mia_resize_bilinear_bwd(const float* dout, float* din_zeroed, int nb, int c, int h, int w, int oh, int ow, void* stream);
/* torchvision >= 0.17 default antialias=True on the tensor path: separable triangle filter (two passes, tmp = [B][C][H][OW]) */
int mia_resize_bilinear_aa(const float* in, float* tmp, float* out, int nb, int c, int h, int w, int oh, int ow, void* stream);
/* JointResize label path / UnetProcessor.postprocess: interpolate(nearest) */
int mia_resize_nearest(const void* in, void* out, int elem_bytes, int64_t planes, int h, int w, int oh, int ow, void* stream);

/* ------------------------------------------------------------------ validation / selection reductions */
/* pred = output.softmax(1).argmax(1) (al_trainer.py:1430-1431) + per-(image, class) hard Dice of calculate_metric_percase
 * (:1539-1556, medpy.metric.dc: 2|A&B|/(|A|+|B|), 0 for an empty prediction).  pred / (labels, workspace, counts[B][K1][3],
 * dice[B][K1]) are each optional.  logits == NULL = label-map mode: `pred` is an INPUT label map (the post-processed
 * prediction of valid_slices, al_trainer.py:1442-1446) and only the counts / Dice against `labels` are computed. */
int mia_argmax_dice_workspace(int nb, int k1, int slabs); /* floats */
int mia_argmax_dice(const float* logits, const long long* labels, long long* pred, int nb, int64_t hw, int k1, int64_t sn, int64_t sk,
                    int64_t sp, int slabs, float* workspace, float* counts, float* dice, void* stream);
/* scores[B][3] = (entropy, least-confidence, margin) acquisition scores of the active-learning selectors
 * (entropy_selector.py:42-49, confidence_selector.py:42-47, margin_selector.py:42-48); smooth = the entropy selector's
 * log2(p + smooth) guard (1e-8 in al_train) */
int mia_selector_scores_workspace(int nb, int slabs); /* floats */
int mia_selector_scores(const float* logits, int nb, int64_t hw, int k1, int64_t sn, int64_t sk, int64_t sp, float smooth, int slabs,
                        float* workspace, float* scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif
