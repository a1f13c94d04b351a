// Shared helpers for libmia_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MIA_F32 0
#define MIA_BF16 1

#define MIA_OK 0
#define MIA_EARG (-1)
#define MIA_EUNSUPPORTED (-2)

extern "C" const char* mia_last_error(void);
void mia_set_error(const char* fmt, ...);

#define MIA_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      mia_set_error(__VA_ARGS__);           \
      return MIA_EARG;                      \
    }                                       \
  } while (0)

// Returns positive hipError_t on launch failure (C-ABI convention, include/mia_hip.h).
#define MIA_LAUNCH_CHECK()                                                  \
  do {                                                                      \
    hipError_t _e = hipGetLastError();                                      \
    if (_e != hipSuccess) {                                                 \
      mia_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return (int)_e;                                                       \
    }                                                                       \
  } while (0)

typedef unsigned short bf16_t;  // raw storage

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, ((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPU = 4;  // elements per 16-byte unit
  static constexpr int DT = MIA_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float cvt(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPU = 8;
  static constexpr int DT = MIA_BF16;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  __device__ static __forceinline__ bf16_t cvt(float v) { return f2bf(v); }
};

// fp32 tensors on the f16 matrix cores ("split" mode of the fp32 tile kernels, option f32_split, ON by default).  An fp32 value has 24
// significand bits and an fp16 value 11, so x * 2^e = h + l with h = f16_rne(x * 2^e), l = f16_rne(x * 2^e - h) keeps 22-23 of them
// (bf16 parts keep 8 each: the same accuracy takes three parts and six products).  Every fp32 operand element becomes ONE 32-bit
// word (h | l << 16): a 16-byte unit still holds four channels, so LDS layout, staging and fragment reads are those of the exact fp32
// kernel.  An A fragment (8 f16 per lane) is (h0, l0, h1, l1, h2, l2, h3, l3) and two v_mfma_f32_16x16x32_f16 against the B fragment's
// (H0, H0, H1, H1, ..) and (L0, L0, L1, L1, ..) forms add all four products (h + l)(H + L) of 16 channels: 2 x 16 matrix cycles instead
// of 4 x 32 for the four v_mfma_f32_16x16x4_f32.  f16 x f16 products are exact in fp32 and accumulation is fp32.
// RANGE: fp16 spans 2^-24 .. 65504, so each operand tensor is scaled by a power of two (exact) chosen from its max |x| -- the caller
// hands the kernels a device pointer to that maximum as an fp32 bit pattern (mia_amax, or a producer kernel's by-product) -- such that
// the maximum lands in [2^14, 2^15); the accumulators are scaled back by 2^-(ea + eb) (v_ldexp_f32, exact) before bias / statistics /
// store.  Elements down to 2^-18 of the tensor maximum keep all 22 bits (l stays a normal fp16; the f16 MFMA keeps denormal inputs --
// tools/probe/mfma_f16_denorm.hip), smaller ones keep an ABSOLUTE error <= 2^-40 of the maximum.  Measured on the full-width nets:
// logits within ~3e-6 of the fp32 CPU oracle (exact fp32 MFMA kernels: 2.4e-6 .. 5.9e-6; the two-part bf16 split of round 4: 3e-5).
struct SplitF16 {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  // power-of-two exponent e with amax * 2^e in [2^14, 2^15) from the fp32 bit pattern of amax (0 / denormal amax: e = 120; inf / NaN:
  // e = -114, the non-finite values then propagate as they would through fp32 products)
  static __device__ __forceinline__ int exp_of(unsigned amax_bits) {
    const int e = 141 - (int)((amax_bits >> 23) & 0xFFu);
    return e > 120 ? 120 : e;
  }
  static __device__ __forceinline__ float pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }  // -126 <= e <= 127
  static __device__ __forceinline__ u32x4 unit(const u32x4& raw, float s) {  // 4 fp32 -> 4 words (h | l << 16) of x * s
    const f32x4 x = __builtin_bit_cast(f32x4, raw) * s;
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 v = {x[2 * p], x[2 * p + 1]};
      const h2 hh = __builtin_convertvector(v, h2);  // RNE
      const h2 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f2), h2);  // the residual is exact in fp32
      const unsigned hp = __builtin_bit_cast(unsigned, hh), lp = __builtin_bit_cast(unsigned, ll);
      o[2 * p] = __builtin_amdgcn_perm(lp, hp, 0x05040100u);      // h0 | l0 << 16
      o[2 * p + 1] = __builtin_amdgcn_perm(lp, hp, 0x07060302u);  // h1 | l1 << 16
    }
    return o;
  }
  // planar form of the same split: 4 fp32 -> (h0 | h1 << 16, h2 | h3 << 16) and (l0 | l1 << 16, l2 | l3 << 16) -- the converts deliver exactly
  // these pairs, the interleaved form spends four v_perm on top.  planes_of: interleaved words -> the two pairs.
  static __device__ __forceinline__ void unit_planar(const u32x4& raw, float s, u32x2& hp, u32x2& lp) {
    const f32x4 x = __builtin_bit_cast(f32x4, raw) * s;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 v = {x[2 * p], x[2 * p + 1]};
      const h2 hh = __builtin_convertvector(v, h2);  // RNE
      const h2 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f2), h2);
      hp[p] = __builtin_bit_cast(unsigned, hh); lp[p] = __builtin_bit_cast(unsigned, ll);
    }
  }
  static __device__ __forceinline__ void planes_of(const u32x4& w, u32x2& hp, u32x2& lp) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      hp[p] = __builtin_amdgcn_perm(w[2 * p + 1], w[2 * p], 0x05040100u);
      lp[p] = __builtin_amdgcn_perm(w[2 * p + 1], w[2 * p], 0x07060302u);
    }
  }
  static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ u32x4 dup_hi(const u32x4& w) {  // (h, l) words -> (h, h)
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_perm(w[d], w[d], 0x01000100u);
    return o;
  }
  static __device__ __forceinline__ u32x4 dup_lo(const u32x4& w) {  // (h, l) words -> (l, l)
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_perm(w[d], w[d], 0x03020302u);
    return o;
  }
  // (h, l) words -> (l, h): with the interleaved layout on BOTH sides, (h, l) x (H, L) gives h H + l L and (h, l) x (L, H) gives h L + l H --
  // all four products from the fragment as it is plus ONE rotated copy (4 VALU per fragment; the (H, H) / (L, L) forms above cost 8)
  static __device__ __forceinline__ u32x4 swap_hl(const u32x4& w) {
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(w[d], w[d], 16);
    return o;
  }
  static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  // (h, l) A fragment x expanded B fragment, and expanded A fragment x (h, l) B fragment
  static __device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& bh, const u32x4& bl, f32x4 c) { return mfma(a, bl, mfma(a, bh, c)); }
  static __device__ __forceinline__ f32x4 mma_a(const u32x4& ah, const u32x4& al, const u32x4& b, f32x4 c) { return mfma(al, b, mfma(ah, b, c)); }
  static __device__ __forceinline__ float unscale(float acc, int eo) { return __builtin_ldexpf(acc, eo); }  // eo = -(ea + eb)
};

// Norm + LeakyReLU backward for one element (reference autograd of blocks.py:98-102): g = dz * lrelu'(scale*y + shift),
// dy = scale*(g - c1 - xhat*c2) = scale*g + ka*y + kb with ka = -scale*c2*xa, kb = -scale*(c1 + c2*xb).  ONE definition with explicit
// fmas, shared by the streaming apply pass (norm.hip) and the kernels that form dy on load (stem.hip: mia_stem_wgrad_fused), so
// both produce the same bits.
__device__ __forceinline__ float norm_bwd_dy(float g, float yv, float sc, float sf, float ka, float kb, float slope) {
  if (!(__builtin_fmaf(sc, yv, sf) > 0.f)) g *= slope;
  return __builtin_fmaf(sc, g, __builtin_fmaf(ka, yv, kb));
}

// Two extra wait states behind a 12- / 16-byte vector store whose data registers the compiler may recycle at once (store-data hazard:
// tools/check_store_hazard.py, profiles/r05_store_hazard.txt -- hipcc's own 2 wait states were not enough next to a dozen in-flight
// loads; 2 more always were).  VALU instructions may not cross (mask 0x3AC lets SALU, MFMA, vector loads and LDS instructions through, so
// loads and LDS reads of later iterations still overlap; vector STORES may not cross either, or the store itself would sink past the pad): put it right behind the store.
// (store_data_fence() in FRONT of the store keeps the VALU work that precedes it in the source from being scheduled behind it.)
__device__ __forceinline__ void store_data_fence() { __builtin_amdgcn_sched_barrier(0x3AC); }
__device__ __forceinline__ void store_data_pad() {
  __builtin_amdgcn_sched_barrier(0x3AC);
  asm volatile("s_nop 1" ::: "memory");
  __builtin_amdgcn_sched_barrier(0x3AC);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* red /*>= 16 floats LDS*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    r = (l < nw) ? red[l] : 0.f;
    r = wave_sum(r);
  }
  return r;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
