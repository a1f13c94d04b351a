// Shared geometry / MFMA helpers of the implicit-GEMM conv kernels (conv_mma.hip, conv_mma_fast.hip).
#pragma once
#include "common.h"

enum { MODE_G3S1 = 0, MODE_G3S2 = 1, MODE_G2S2 = 2, MODE_T3S2 = 3, MODE_T2S2 = 4, MODE_G1 = 5 };

struct ConvArgs {
  const void* in1; const void* in2; int c1; int c2;
  const void* wp; const float* bias;
  void* out1; void* out2; int o1; int o2;
  float* stats;
  int N, Hin, Win, Hout, Wout;
  int npad, kpad, flip;
  int tiles_x, tiles_y, nblk_n;
  int st_tiles_y;  // rows of the statistics tile grid (8-row tiles under the 512-thread stride-2 kernel's 16-row tiles; else = tiles_y)
  int vec_in, vec_out;
  int xcd;  // fast kernel: XCD-aware block order (the blocks that share an input tile run on one XCD, back to back)
  // normalise-on-load (mia_conv_mma_nl): in1 is the RAW conv output y of the producing PlainBlock and the kernel forms
  // lrelu(nl_scale[n][c] * y + nl_shift[n][c]) while staging it; nullptr = in1 is an ordinary activation
  const float* nl_scale = nullptr; const float* nl_shift = nullptr; float nl_slope = 0.f;
  // column-reduce epilogue (mia_conv_mma_cr): this launch is an INPUT GRADIENT whose output dz is the gradient w.r.t. the
  // activated output of the PRODUCING PlainBlock; cr_y is that block's raw conv output (same geometry as out1) and the
  // epilogue adds up, per tile, sum g and sum g*xhat with g = dz * lrelu'(scale*y + shift), xhat = xa*y + xb -- the
  // reduction pass of that block's norm backward -- into `stats` ([N][tiles][o1][2])
  // accumulate mode (mia_conv_mma_acc): out1 += result instead of out1 = result (the second gradient piece of a skip tensor is
  // added into the first in the epilogue of the kernel that produces it: one read-modify-write instead of a second tensor)
  int acc_out = 0;
  const void* cr_y = nullptr; const float* cr_scale = nullptr; const float* cr_shift = nullptr;
  const float* cr_xa = nullptr; const float* cr_xb = nullptr; float cr_slope = 0.f;
  // fp32 tensors, products on the f16 matrix cores from two-part split operands (option f32_split; common.h SplitF16): max |x| of
  // each operand tensor as an fp32 bit pattern in device memory (mia_amax); split = all three known and the option on
  int split = 0;
  const unsigned* amax_in1 = nullptr; const unsigned* amax_in2 = nullptr; const unsigned* amax_w = nullptr;
  int wsplit = 0;  // split mode: wp already holds (h | l) words of w * 2^eb (mia_split_f16_batch; eb from amax_w) -- staged as they are
  // optional by-product (fp32 tensors): max |out1| / max |out2| folded into these ZEROED slots by the epilogue, for the convs that
  // consume the outputs directly (ConvTranspose output -> decoder conv; a decoder conv's second input gradient -> ConvTranspose backward)
  unsigned* amax_out1 = nullptr; unsigned* amax_out2 = nullptr;
};

__device__ __forceinline__ int pi16(int r) {
  // rows 4..11 <-> even pixels, rows 0..3 / 12..15 <-> odd pixels (see header comment)
  return (r >= 4 && r < 12) ? 2 * (r - 4) : (r < 4 ? 2 * r + 1 : 2 * (r - 8) + 1);
}

template <int MODE, int MT> struct Geo {
  static constexpr int TH = 4 * MT;
  static constexpr int S = (MODE == MODE_G3S2 || MODE == MODE_G2S2) ? 2 : 1;
  static constexpr int IH = MODE == MODE_G3S1 ? TH + 2 : MODE == MODE_G3S2 ? 2 * TH + 1 : MODE == MODE_G2S2 ? 2 * TH
                          : MODE == MODE_T3S2 ? TH + 1 : TH;
  static constexpr int IW = MODE == MODE_G3S1 ? 18 : MODE == MODE_G3S2 ? 33 : MODE == MODE_G2S2 ? 32
                          : MODE == MODE_T3S2 ? 17 : 16;
  static constexpr int IWH = (S == 2) ? (IW + 1) / 2 : 0;
  static constexpr int PITCH = (S == 2) ? 2 * IWH : IW;
  static constexpr int NPIX = IH * PITCH;
  static constexpr int NPA = ((NPIX + 13) / 16) * 16 + 2;  // >= NPIX, == 2 (mod 16)
  static constexpr int MAXTAPS = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 9 : (MODE == MODE_G2S2 || MODE == MODE_T3S2) ? 4 : 1;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], c, 0, 0, 0);
    return c;
  }
};

// `reserve_cus` (option of the same name): the persistent kernels below size their grids to the CU count; under data parallelism
// RCCL's ring kernels need somewhere to run WHILE a persistent kernel owns the chip, and because the work lists are static
// (workgroup b takes items b, b + G, ...) a workgroup that cannot be placed would not lose a share of the work but run AFTER
// the others: one occupied CU doubles the launch.  With k CUs reserved the grids use (CUs - k) workgroups (a multiple of 8: one
// free CU per XCD for k = 8); the items are the same, so results are bit-identical for every k.
inline int persistent_cus(int ncu, int reserve) { int v = (ncu - reserve) & ~7; return v < 8 ? 8 : v; }

// conv_mma_fast.hip: branch-free variant (buffer loads/stores with hardware OOB zero-fill); returns false when the
// shape does not meet its alignment contract and the generic kernel must be used.
bool conv_mma_fast_eligible(int dtype, const ConvArgs& a, int nt);
int conv_mma_fast_launch(int mode, int dtype, const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st);

// conv64.hip: persistent 64 -> 64 channel 3x3 / stride-1 bf16 kernel with register-resident weights (the canonical block
// launch of the benchmark and its input gradient); false = shape outside its contract.
bool conv64_eligible(int mode, int dtype, const ConvArgs& a);
int conv64_launch(const ConvArgs& a, int blocks_override, int reserve_cus, hipStream_t st);

// conv_bt.hip: 512-thread "big tile" LDS-DMA kernel for the stride-1 3x3 bf16 convs with >= 64-channel blocks (levels >= 1
// of the network, the decoder's two-source convs, cfg5's 96-multiples); false = shape outside its contract.
bool conv_bt_eligible(int mode, int dtype, const ConvArgs& a);
int conv_bt_launch(const ConvArgs& a, int order /* option conv_bt_order: 1 = tile-major item order */, int reserve_cus, hipStream_t st);

// conv64_dma.hip: second generation of the 64 -> 64 (| 64) kernel: 512 threads per CU, LDS-DMA ring of three input tiles, one
// barrier per tile; both destinations of a two-destination input gradient in one pass.
bool conv64_dma_eligible(int mode, int dtype, const ConvArgs& a);
int conv64_dma_launch(const ConvArgs& a, int reserve_cus, hipStream_t st);

// conv_pw.hip: ConvTranspose2d 2x2 / stride 2 forward (MODE_T2S2) and input gradient (MODE_G2S2), bf16, as one pointwise GEMM
// per launch (512 threads per CU, LDS-DMA ring of three 64-wide K stages); false = shape outside its contract.
bool conv_pw_eligible(int mode, int dtype, const ConvArgs& a);
int conv_pw_launch(int mode, const ConvArgs& a, int reserve_cus, hipStream_t st);
