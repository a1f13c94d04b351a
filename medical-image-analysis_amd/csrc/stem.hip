// Stem convolution (Cin = 1): Conv2d(1, C0, 3, padding=1) forward and its weight gradient, gfx950.
//
// Reference: first PlainBlock of the encoder (src/models/unet/unet.py:54-66 with input_channels=1,
// blocks.py:83-90).  With one input channel the conv is 9 FMAs per output: arithmetic intensity 4-9 FLOP/B,
// i.e. pure HBM streaming -- MFMA would waste 31/32 of its K.  Forward: each thread
// produces 8 (bf16) / 4 (fp32) consecutive output channels of one pixel from the 3x3 fp-image patch and
// writes one 16-byte unit; per-(image, slab, channel) sum / sum-of-squares partials for the norm come from
// an LDS reduction.  Weight gradient: dW[co][t] = sum_p x[p+t] * dy[p][co], a 9 x C0 reduction over all
// pixels, two-stage (block partials, then the generic slab reduce); no float atomics.
#include "common.h"

template <typename T, typename TI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ w /*[C0][9]*/,
                                                       const float* __restrict__ bias, T* __restrict__ y, int h, int wd, int c0,
                                                       int slabs, float* __restrict__ stats /*[N][slabs][C0][2]*/) {
  constexpr int EPU = Elem<T>::EPU;
  extern __shared__ float red[];  // [2][lanes][c0]
  const int upp = c0 / EPU, lanes = 256 / upp;
  const int u = threadIdx.x % upp, pl = threadIdx.x / upp;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int hw = h * wd, per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  const TI* img = x + (size_t)n * hw;
  // a thread owns one 16-byte channel unit for its whole life: its 9 x EPU weights and EPU biases live in registers
  float wr[9][EPU], br[EPU];
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    const int co = u * EPU + e;
    br[e] = bias ? bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][e] = w[co * 9 + t];
  }
  float s1[EPU], s2[EPU];
#pragma unroll
  for (int e = 0; e < EPU; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (pl < lanes && r0 + pl < r1) {
    int p = r0 + pl;
    int py = p / wd, px = p - py * wd;  // one division per thread; the walk below is incremental
    T* yrow = y + ((size_t)n * hw) * c0 + u * EPU;
    for (; p < r1; p += lanes) {
      float xv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        xv[t] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) ? Elem<TI>::ld(img + yy * wd + xx) : 0.f;
      }
      alignas(16) T out[EPU];
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        float a = br[e];
#pragma unroll
        for (int t = 0; t < 9; ++t) a += xv[t] * wr[t][e];
        s1[e] += a; s2[e] += a * a;
        out[e] = Elem<T>::cvt(a);
      }
      *reinterpret_cast<u32x4*>(yrow + (size_t)p * c0) = *reinterpret_cast<const u32x4*>(out);
      px += lanes;
      while (px >= wd) { px -= wd; ++py; }
    }
  }
  if (stats != nullptr) {
    if (pl < lanes)
#pragma unroll
      for (int e = 0; e < EPU; ++e) { red[pl * c0 + u * EPU + e] = s1[e]; red[(lanes + pl) * c0 + u * EPU + e] = s2[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c0; i += 256) {
      const int k = i / c0, ch = i % c0;
      float t = 0.f;
      for (int j = 0; j < lanes; ++j) t += red[(k * lanes + j) * c0 + ch];
      stats[(((size_t)n * slabs + s) * c0 + ch) * 2 + k] = t;
    }
  }
}

// partial[blk][9][c0] = sum over the block's pixels of x[p + t] * dy[p][co]
template <typename T, typename TI>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const TI* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                         int slabs, int h, int wd, int c0) {
  constexpr int EPU = Elem<T>::EPU;
  extern __shared__ float red[];  // [lanes][c0 + 1]
  const int upp = c0 / EPU, lanes = 256 / upp;
  const int u = threadIdx.x % upp, pl = threadIdx.x / upp;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;  // block = one pixel slab of one image
  const int hw = h * wd, per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float acc[9][EPU];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[t][e] = 0.f;
  if (pl < lanes && r0 + pl < r1) {
    const TI* img = x + (size_t)n * hw;
    const T* grow = dy + ((size_t)n * hw) * c0 + u * EPU;
    int p = r0 + pl;
    int py = p / wd, px = p - py * wd;  // one division per thread; the walk below is incremental
    for (; p < r1; p += lanes) {
      alignas(16) T g[EPU];
      *reinterpret_cast<u32x4*>(g) = *reinterpret_cast<const u32x4*>(grow + (size_t)p * c0);
      float gf[EPU];
#pragma unroll
      for (int e = 0; e < EPU; ++e) gf[e] = Elem<T>::ld(g + e);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        const float xv = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) ? Elem<TI>::ld(img + yy * wd + xx) : 0.f;
#pragma unroll
        for (int e = 0; e < EPU; ++e) acc[t][e] += xv * gf[e];
      }
      px += lanes;
      while (px >= wd) { px -= wd; ++py; }
    }
  }
  const int shs = c0 + 1;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (pl < lanes)
#pragma unroll
      for (int e = 0; e < EPU; ++e) red[pl * shs + u * EPU + e] = acc[t][e];
    __syncthreads();
    if (threadIdx.x < c0) {
      float sm = 0.f;
      for (int j = 0; j < lanes; ++j) sm += red[j * shs + threadIdx.x];
      part[((size_t)blockIdx.x * 9 + t) * c0 + threadIdx.x] = sm;
    }
  }
}

// grad[co][t] (+)= sum_blk part[blk][t][co]; block = 16 outputs x 16 lanes
__global__ void stem_wgrad_final_kernel(const float* __restrict__ part, int nblk, int c0, float* __restrict__ grad, int accumulate) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl, tot = 9 * c0;
  float s = 0.f;
  if (i < tot)
    for (int b = tl; b < nblk; b += 16) s += part[(size_t)b * tot + i];
  sh[tl][cl] = s;
  __syncthreads();
  if (tl == 0 && i < tot) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += sh[j][cl];
    const int tap = i / c0, co = i % c0;
    grad[co * 9 + tap] = accumulate ? grad[co * 9 + tap] + t : t;
  }
}

#define STEM_SLABS 64
#define STEM_WBLOCKS 2048
extern "C" int mia_stem_slabs(void) { return STEM_SLABS; }
extern "C" int mia_stem_wgrad_workspace(int c0) { return STEM_WBLOCKS * 9 * c0; }

static bool stem_ok(int dtype, int c0) {
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  return c0 % epu == 0 && c0 / epu <= 256 && c0 <= 256;
}

extern "C" int mia_stem_fwd(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype, float* stat_partials,
                            int n, int h, int wd, int c0, void* stream) {
  MIA_CHECK_ARG(x && w && y && n > 0 && h > 0 && wd > 0 && c0 > 0, "mia_stem_fwd: bad arguments");
  MIA_CHECK_ARG((dtype == MIA_BF16 || dtype == MIA_F32) && (x_dtype == MIA_BF16 || x_dtype == MIA_F32), "mia_stem_fwd: bad dtype");
  MIA_CHECK_ARG(stem_ok(dtype, c0), "mia_stem_fwd: c0=%d must be a multiple of the 16-byte unit and <= 256", c0);
  const int epu = dtype == MIA_BF16 ? 8 : 4, lanes = 256 / (c0 / epu);
  MIA_CHECK_ARG((int64_t)h * wd < ((int64_t)1 << 31), "mia_stem_fwd: image too large");
  const size_t shb = (size_t)(2 * lanes * c0) * 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(n * STEM_SLABS);
#define SF(T, TI) hipLaunchKernelGGL((stem_fwd_kernel<T, TI>), grid, dim3(256), shb, st, static_cast<const TI*>(x), w, bias, static_cast<T*>(y), h, wd, c0, STEM_SLABS, stat_partials)
  if (dtype == MIA_BF16 && x_dtype == MIA_F32) SF(bf16_t, float);
  else if (dtype == MIA_BF16) SF(bf16_t, bf16_t);
  else if (x_dtype == MIA_F32) SF(float, float);
  else SF(float, bf16_t);
#undef SF
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_stem_wgrad(const void* x, int x_dtype, const void* dy, int dtype, float* workspace, float* grad, int n, int h, int wd,
                              int c0, int accumulate, void* stream) {
  MIA_CHECK_ARG(x && dy && workspace && grad && n > 0 && h > 0 && wd > 0, "mia_stem_wgrad: bad arguments");
  MIA_CHECK_ARG((dtype == MIA_BF16 || dtype == MIA_F32) && (x_dtype == MIA_BF16 || x_dtype == MIA_F32), "mia_stem_wgrad: bad dtype");
  MIA_CHECK_ARG(stem_ok(dtype, c0), "mia_stem_wgrad: c0=%d must be a multiple of the 16-byte unit and <= 256", c0);
  const int epu = dtype == MIA_BF16 ? 8 : 4, lanes = 256 / (c0 / epu);
  MIA_CHECK_ARG((int64_t)h * wd < ((int64_t)1 << 31) && n <= STEM_WBLOCKS, "mia_stem_wgrad: image or batch too large");
  // one block = one pixel slab of one image (n * slabs <= STEM_WBLOCKS partial rows in the workspace)
  const int64_t hw = (int64_t)h * wd;
  int slabs = STEM_WBLOCKS / n;
  if (slabs > (int)((hw + 255) / 256)) slabs = (int)((hw + 255) / 256);
  if (slabs < 1) slabs = 1;
  const int blocks = n * slabs;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t shb = (size_t)lanes * (c0 + 1) * 4;
#define SW(T, TI) hipLaunchKernelGGL((stem_wgrad_kernel<T, TI>), dim3(blocks), dim3(256), shb, st, static_cast<const TI*>(x), static_cast<const T*>(dy), workspace, slabs, h, wd, c0)
  if (dtype == MIA_BF16 && x_dtype == MIA_F32) SW(bf16_t, float);
  else if (dtype == MIA_BF16) SW(bf16_t, bf16_t);
  else if (x_dtype == MIA_F32) SW(float, float);
  else SW(float, bf16_t);
#undef SW
  hipLaunchKernelGGL(stem_wgrad_final_kernel, dim3(ceil_div(9 * c0, 16)), dim3(256), 0, st, workspace, blocks, c0, grad, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
