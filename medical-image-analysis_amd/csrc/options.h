// Library options (kernel-selection / launch-shape knobs for A/B measurements; never numerics contracts).  16 of them since round 5: the
// measured-and-rejected experiments of rounds 3-4 (Winograd conv64, conv_pw T3S2, conv_t3_wide, wgrad_narrow, conv_mt8, the column-reduce
// epilogue, the register-staged two-workgroup weight gradient as an alternative to the LDS-DMA ring) left the shipping library; their
// records stay under profiles/, their kernels are compiled only with -DMIA_EXPERIMENTS (tools/probe/).
//
// ONE table, read from the environment ONCE (first use, std::call_once), changed only through mia_set_option; every C-ABI
// entry point takes ONE snapshot (`const MiaOptions o = mia_options();`) and passes it down, so a call never sees two values
// of an option.  The fields are atomics: set / get / snapshot are safe from any thread.  Nothing else in the library reads
// getenv or keeps a lazily-initialised knob.
//
// Geometry queries (mia_conv_mma_tiles, mia_wgrad_geometry, mia_wgrad_target_blocks) size the caller's statistics / slab
// buffers in a SEPARATE call from the launch that fills them.  The options that change that geometry (wgrad_dma, wgrad_bt,
// wgrad_t2, reserve_cus) are A/B knobs: mia_set_option on them must not race with compute calls of another thread
// (set them between steps; the Python side does -- tools/ab_option.py).  Everything else may change at any time.
#pragma once

struct MiaOptions {
  int conv_xcd;       // XCD-aware block order of conv_mma_fast_kernel                         env MIA_CONV_XCD      default 1
  int conv64;         // persistent register-weight kernel for 64 -> 64 3x3 bf16               env MIA_CONV64        default 1
  int conv64_dma;     // 512-thread LDS-DMA 64-channel kernel: 1 two-destination 64 -> (64|64) launches only, 2 every 64 -> 64 launch, 0 never   env MIA_CONV64_DMA   default 1
  int conv_bt;        // big-tile LDS-DMA kernel for the wide stride-1 3x3 bf16 convs          env MIA_CONV_BT       default 1
  int conv_bt_order;  // item order of conv_bt_kernel: 1 = a tile's channel blocks together on one XCD, 0 = channel block slow   env MIA_CONV_BT_ORDER default 1
  int conv_s2_wide;   // stride-2 3x3 bf16 forward with 128-multiples of output channels on 512-thread 16-row tiles (1: from 128 input channels on, 2: always)   env MIA_CONV_S2_WIDE  default 1
  int conv_pw;        // ConvTranspose 2x2 stride 2 forward / input gradient (bf16) as one pointwise GEMM on the LDS-DMA ring   env MIA_CONV_PW       default 1
  int conv_pw_s2;     // stride-2 3x3 bf16 forward as a tap-gathered GEMM on the conv_pw ring (needs conv_pw; 1: up to 256 input channels, 2: always; -0.13 ms of kernel time per cfg3 step measured INSIDE the step with rocprofv3, two interleaved pairs)   env MIA_CONV_PW_S2    default 1
  int wgrad_xcd;      // XCD-aware block order of the bf16 weight-gradient kernels             env MIA_WGRAD_XCD     default 1
  int wgrad_dma;      // LDS-DMA ring weight-gradient kernel (3x3 stride 1 bf16)               env MIA_WGRAD_DMA     default 1
  int wgrad_bt;       // 512-thread 128 n x 64 k weight-gradient kernel (3x3 stride 1 bf16, >= 128 output channels)   env MIA_WGRAD_BT      default 1
  int wgrad_t2;       // ConvTranspose 2x2 weight gradient (bf16, >= 128 coarse channels) on the 512-thread three-stage ring kernel   env MIA_WGRAD_T2      default 1
  int stream_blocks;  // target block count of the norm / activation streaming passes          env MIA_STREAM_BLOCKS default 32768
  int stem_mfma;      // matrix-core stem kernel for fp32 images                               env MIA_STEM_MFMA     default 1
  int f32_split;      // fp32 convs / weight gradients of the branch-free tile kernels on the f16 matrix cores from two-part split operands scaled per tensor (common.h SplitF16: x * 2^e = h + l in fp16, 22-23 significand bits, fp32 accumulate) whenever the caller passes the operands' maxima; 2 = the convs with 32 x 32 tiles (>= 32 output channels) take THREE products on separate h / l planes (h H + h L + l H; the dropped l L is below the parts' own rounding), 1 = four products on interleaved words everywhere, 0 = always the exact fp32 MFMA kernels   env MIA_F32_SPLIT   default 2
  int reserve_cus;    // CUs the persistent kernels leave free (grids of conv_bt / conv_pw / conv64 / conv64_dma, split-K target of the weight gradients): room for RCCL's ring kernels under data parallelism   env MIA_RESERVE_CUS   default 0
};

MiaOptions mia_options();  // consistent snapshot, by value
