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

// fp32 tensors on the bf16 matrix cores ("split" mode of the fp32 tile kernels, option f32_split).  Every fp32 operand element
// becomes ONE 32-bit word (hi | lo << 16): hi = bf16_rne(x), lo = bf16_rne(x - hi), i.e. x to 16-17 significant bits; a 16-byte
// unit still holds four channels, so LDS layout, staging and fragment reads are those of the exact fp32 kernel.  An A fragment
// (8 bf16 per lane) is then (h0, l0, h1, l1, h2, l2, h3, l3) and two MFMAs against the B fragment's (H0, H0, H1, H1, ..) and
// (L0, L0, L1, L1, ..) forms add all four products (h + l)(H + L) of 16 channels: 2 x 16 cycles instead of 4 x 32 for the four
// v_mfma_f32_16x16x4_f32, fp32 accumulation.  bf16 x bf16 products are exact in fp32, so the only error is the 2^-17 operand
// representation (measured on the full-width nets: logits within 3e-5 of the exact fp32 path).
struct SplitBf16 {
  static __device__ __forceinline__ u32x4 unit(const u32x4& raw) {  // 4 fp32 -> 4 words (hi | lo << 16)
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const f32x4 x = __builtin_bit_cast(f32x4, raw);
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 v = {x[2 * p], x[2 * p + 1]};
      const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));  // h0 | h1 << 16
      const f2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xFFFF0000u)};
      const unsigned lp = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, b2));  // l0 | l1 << 16
      o[2 * p] = __builtin_amdgcn_perm(lp, hp, 0x05040100u);      // h0 | l0 << 16
      o[2 * p + 1] = __builtin_amdgcn_perm(lp, hp, 0x07060302u);  // h1 | l1 << 16
    }
    return o;
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
  static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // (h, l) A fragment x expanded B fragment, and expanded A fragment x (h, l) B fragment
  static __device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& bh, const u32x4& bl, f32x4 c) { return mfma(a, bl, mfma(a, bh, c)); }
  static __device__ __forceinline__ f32x4 mma_a(const u32x4& ah, const u32x4& al, const u32x4& b, f32x4 c) { return mfma(al, b, mfma(ah, b, c)); }
};

// Three-way split (option f32_split = 2): x = h + m + l holds all 24 significand bits of an fp32 value in three bf16 parts, and six of
// the nine part products reach ~2^-24 per product in THREE MFMAs per 16 channels: (xh, xm) x (wh, wh) -> xh wh + xm wh;
// (xh, xm) x (wm, wm) -> xh wm + xm wm; (xh, xl) x (wl, wh) -> xh wl + xl wh (dropped: xm wl, xl wm, xl wl <= 2^-24 of the product).
// 48 matrix cycles per 16 channels against 128 for the four v_mfma_f32_16x16x4_f32, at the accuracy of the fp32 kernel.
struct Split3 {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  static __device__ __forceinline__ unsigned cvt2(f2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2)); }
  static __device__ __forceinline__ f2 up2(unsigned p) { return f2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xFFFF0000u)}; }
  // 4 fp32 -> words (h | m << 16) and (h | l << 16) of the same four channels
  static __device__ __forceinline__ void act(const u32x4& raw, u32x4& hm, u32x4& hl) {
    const f32x4 x = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 v = {x[2 * p], x[2 * p + 1]};
      const unsigned hp = cvt2(v);
      const f2 r1 = v - up2(hp);
      const unsigned mp = cvt2(r1);
      const unsigned lp = cvt2(r1 - up2(mp));
      hm[2 * p] = __builtin_amdgcn_perm(mp, hp, 0x05040100u); hm[2 * p + 1] = __builtin_amdgcn_perm(mp, hp, 0x07060302u);
      hl[2 * p] = __builtin_amdgcn_perm(lp, hp, 0x05040100u); hl[2 * p + 1] = __builtin_amdgcn_perm(lp, hp, 0x07060302u);
    }
  }
  // weights: 4 fp32 -> words (h | m << 16) and the four l parts as two words (l0 | l1 << 16), (l2 | l3 << 16)
  static __device__ __forceinline__ void wgt(const u32x4& raw, u32x4& hm, unsigned& l01, unsigned& l23) {
    const f32x4 x = __builtin_bit_cast(f32x4, raw);
    unsigned lp[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f2 v = {x[2 * p], x[2 * p + 1]};
      const unsigned hp = cvt2(v);
      const f2 r1 = v - up2(hp);
      const unsigned mp = cvt2(r1);
      lp[p] = cvt2(r1 - up2(mp));
      hm[2 * p] = __builtin_amdgcn_perm(mp, hp, 0x05040100u); hm[2 * p + 1] = __builtin_amdgcn_perm(mp, hp, 0x07060302u);
    }
    l01 = lp[0]; l23 = lp[1];
  }
  // (l | h << 16) words of the B operand from the (h | m) words and the l pairs
  static __device__ __forceinline__ u32x4 lh(const u32x4& hm, unsigned l01, unsigned l23) {
    return u32x4{__builtin_amdgcn_perm(hm[0], l01, 0x05040100u), __builtin_amdgcn_perm(hm[1], l01, 0x05040302u),
                 __builtin_amdgcn_perm(hm[2], l23, 0x05040100u), __builtin_amdgcn_perm(hm[3], l23, 0x05040302u)};
  }
};

// Norm + LeakyReLU backward for one element (reference autograd of blocks.py:98-102): g = dz * lrelu'(scale*y + shift),
// dy = scale*(g - c1 - xhat*c2) = scale*g + ka*y + kb with ka = -scale*c2*xa, kb = -scale*(c1 + c2*xb).  ONE definition with explicit
// fmas, shared by the streaming apply pass (norm.hip) and the kernels that form dy on load (stem.hip: mia_stem_wgrad_fused), so
// both produce the same bits.
__device__ __forceinline__ float norm_bwd_dy(float g, float yv, float sc, float sf, float ka, float kb, float slope) {
  if (!(__builtin_fmaf(sc, yv, sf) > 0.f)) g *= slope;
  return __builtin_fmaf(sc, g, __builtin_fmaf(ka, yv, kb));
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
