// Instance / batch normalisation + channel dropout + LeakyReLU(0.01), NHWC, gfx950.
//
// Reference block: conv -> Dropout2d -> InstanceNorm2d|BatchNorm2d(eps=1e-5, affine) -> LeakyReLU
// (src/models/unet/blocks.py:92-102).  All three collapse to per-(image, channel) coefficients:
//     y' = m*y                      (m = 0 or 1/(1-p), Dropout2d channel mask)
//     xhat = (y' - mean')*rstd' = xa*y + xb
//     z = lrelu(gamma*xhat + beta) = lrelu(scale*y + shift)
// The conv epilogue (conv_mma.hip) already produced per-tile sum / sum-of-squares partials, so the
// statistics cost no extra pass over the activation; the apply passes are pure HBM streams with
// 16-byte accesses, reductions use wave shuffles + one LDS hop, and no float atomics are used.
#include "common.h"
#include "options.h"
#include <stdlib.h>

#define NORM_INSTANCE 0
#define NORM_BATCH 1

// ---------------------------------------------------------------- forward statistics finalize
// partials: [N][T][C][2] (sum, sumsq) of the raw conv output, from the conv epilogue.  Two tiny launches, fully
// parallel, fixed summation order, double accumulation (var = E[y^2] - mean^2 must not cancel in fp32):
//  K1  grid (image, 16-channel group): the 16 lanes of a channel sweep the tile partials -> per-(n,c) sum / sumsq,
//      parked in xa / xb.
//  K2  grid (16-channel group): lanes sweep images.  Instance norm: coefficients per (n,c).  Batch norm: combine
//      over the batch (with the Dropout2d masks folded in), update running statistics, then per-(n,c) coefficients.
// ysum[N][C] (optional) keeps sum(y) for the closed-form conv-bias gradient of the backward pass.
// INST: instance norm needs nothing beyond its own (n, c) sums, so the coefficients are finished here (same arithmetic as
// norm_finalize_kernel's instance branch on the same float-rounded sums) and the finalize launch is skipped.
template <bool INST>
__global__ __launch_bounds__(1024) void norm_fwd_sum_kernel(const float* __restrict__ part, int tiles, int c, float* __restrict__ xa, float* __restrict__ xb,
                                    int64_t hw, const float* __restrict__ drop, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ ysum) {
  // 1024 threads = 16 channels x 64 tile lanes, four independent 8-byte loads in flight per lane: the sums are latency-bound
  // (16 lanes x one load at a time took 20 us for 1024 tiles)
  __shared__ double sh1[16][17], sh2[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int n = blockIdx.x, ch = blockIdx.y * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (ch < c) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t* p = reinterpret_cast<const f2_t*>(part) + (size_t)n * tiles * c + ch;
    int t = tl;
    for (; t + 192 < tiles; t += 256) {
      const f2_t v0 = p[(size_t)t * c], v1 = p[(size_t)(t + 64) * c], v2 = p[(size_t)(t + 128) * c], v3 = p[(size_t)(t + 192) * c];
      s1 += (double)v0.x + (double)v1.x + (double)v2.x + (double)v3.x;
      s2 += (double)v0.y + (double)v1.y + (double)v2.y + (double)v3.y;
    }
    for (; t < tiles; t += 64) { const f2_t v = p[(size_t)t * c]; s1 += v.x; s2 += v.y; }
  }
  // the four tile lanes of a wave first (lanes l, l + 16, l + 32, l + 48 hold one channel), then the 16 waves through LDS
  s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
  s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
  if ((tl & 3) == 0) { sh1[tl >> 2][cl] = s1; sh2[tl >> 2][cl] = s2; }
  __syncthreads();
  if (tl == 0 && ch < c) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1 += sh1[j][cl]; s2 += sh2[j][cl]; }
    const size_t idx = (size_t)n * c + ch;
    if (!INST) { xa[idx] = (float)s1; xb[idx] = (float)s2; return; }
    const double f1 = (double)(float)s1, f2 = (double)(float)s2, M = (double)hw;
    const double m = drop ? (double)drop[idx] : 1.0;
    const double mean = f1 / M;
    double var = f2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(m * m * var + (double)eps);
    const float a_ = (float)(m * rstd), b_ = (float)(-m * mean * rstd);
    xa[idx] = a_; xb[idx] = b_;
    scale[idx] = gamma[ch] * a_; shift[idx] = gamma[ch] * b_ + beta[ch];
    if (ysum) ysum[idx] = (float)f1;
  }
}

__global__ void norm_finalize_kernel(int n_img, int c, int64_t hw, int mode, int training, const float* __restrict__ drop,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                     float* running_mean, float* running_var, long long* num_batches, float* __restrict__ xa,
                                     float* __restrict__ xb, float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ ysum, const float* __restrict__ group, int world,
                                     const float* __restrict__ part, int tiles) {
  __shared__ double sh1[16][17], sh2[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  const double M = (double)hw;
  if (part != nullptr && ch < c) {
    // small problems (few images x tiles): the per-(n, c) tile sums of norm_fwd_sum_kernel are done here, by the same
    // lane that consumes them below, and the separate launch is skipped
    for (int n = tl; n < n_img; n += 16) {
      double s1 = 0.0, s2 = 0.0;
      for (int t = 0; t < tiles; ++t) {
        const float* p = part + (((size_t)n * tiles + t) * c + ch) * 2;
        s1 += p[0]; s2 += p[1];
      }
      xa[(size_t)n * c + ch] = (float)s1; xb[(size_t)n * c + ch] = (float)s2;
    }
  }
  if (mode == NORM_INSTANCE) {
    if (ch < c)
      for (int n = tl; n < n_img; n += 16) {
        const size_t idx = (size_t)n * c + ch;
        const double s1 = xa[idx], s2 = xb[idx];
        const double m = drop ? (double)drop[idx] : 1.0;
        const double mean = s1 / M;
        double var = s2 / M - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(m * m * var + (double)eps);
        const float a_ = (float)(m * rstd), b_ = (float)(-m * mean * rstd);
        xa[idx] = a_; xb[idx] = b_;
        scale[idx] = gamma[ch] * a_; shift[idx] = gamma[ch] * b_ + beta[ch];
        if (ysum) ysum[idx] = (float)s1;
      }
    return;
  }
  double rstd = 0.0, meanp = 0.0;
  if (training) {
    double e1 = 0.0, e2 = 0.0;
    if (ch < c)
      for (int n = tl; n < n_img; n += 16) {
        const size_t idx = (size_t)n * c + ch;
        const double m = drop ? (double)drop[idx] : 1.0;
        e1 += m * xa[idx]; e2 += m * m * xb[idx];
        if (ysum) ysum[idx] = xa[idx];
      }
    sh1[tl][cl] = e1; sh2[tl][cl] = e2;
    __syncthreads();
    e1 = 0.0; e2 = 0.0;
    for (int j = 0; j < 16; ++j) { e1 += sh1[j][cl]; e2 += sh2[j][cl]; }
    double cnt = (double)n_img * M;
    meanp = e1 / cnt;
    double var = e2 / cnt - meanp * meanp;
    if (group && ch < c) {
      // synchronised batch norm: combine every rank's (mean, M2, count) with the pairwise update of Chan et al.
      double gm = 0.0, gm2 = 0.0, gc = 0.0;
      for (int r = 0; r < world; ++r) {
        const float* g = group + (size_t)r * 3 * c;
        const double rm = g[ch], rm2 = g[c + ch], rc = g[2 * c + ch];
        const double d = rm - gm, tot = gc + rc;
        if (tot > 0.0) { gm += d * rc / tot; gm2 += rm2 + d * d * gc * rc / tot; }
        gc = tot;
      }
      cnt = gc; meanp = gm; var = gm2 / gc;
    }
    if (var < 0.0) var = 0.0;
    rstd = 1.0 / sqrt(var + (double)eps);
    if (tl == 0 && ch < c && running_mean) {
      const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * meanp);
      running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
      if (ch == 0 && num_batches) *num_batches += 1;
    }
    __syncthreads();  // all lanes have read the raw sums parked in xa / xb before they are overwritten
  } else if (ch < c) {
    meanp = running_mean[ch];
    rstd = 1.0 / sqrt((double)running_var[ch] + (double)eps);
  }
  if (ch < c)
    for (int n = tl; n < n_img; n += 16) {
      const size_t idx = (size_t)n * c + ch;
      const double m = (training && drop) ? (double)drop[idx] : 1.0;
      const float a_ = (float)(m * rstd), b_ = (float)(-meanp * rstd);
      xa[idx] = a_; xb[idx] = b_;
      scale[idx] = gamma[ch] * a_; shift[idx] = gamma[ch] * b_ + beta[ch];
    }
}

extern "C" int mia_norm_finalize(const float* partials, int n, int tiles, int c, int64_t hw, int mode, int training,
                                 const float* drop_scale, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, long long* num_batches, float* xa, float* xb,
                                 float* scale, float* shift, float* ysum, void* stream) {
  MIA_CHECK_ARG(mode == NORM_INSTANCE || mode == NORM_BATCH, "mia_norm_finalize: bad mode %d", mode);
  MIA_CHECK_ARG(gamma && beta && xa && xb && scale && shift && n > 0 && c > 0 && hw > 0, "mia_norm_finalize: bad arguments");
  MIA_CHECK_ARG(partials || (mode == NORM_BATCH && !training), "mia_norm_finalize: partials required");
  MIA_CHECK_ARG(mode == NORM_INSTANCE || training || (running_mean && running_var), "mia_norm_finalize: eval batch norm needs running stats");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cgroups = ceil_div(c, 16);
  const bool need_sums = mode == NORM_INSTANCE || training;
  const bool inline_sums = need_sums && (int64_t)n * tiles <= 1024;  // one launch instead of two when the sums are short
  if (need_sums && !inline_sums && mode == NORM_INSTANCE) {  // long sums, instance norm: sums and coefficients in ONE launch
    hipLaunchKernelGGL(norm_fwd_sum_kernel<true>, dim3(n, cgroups), dim3(1024), 0, st, partials, tiles, c, xa, xb, hw, drop_scale, gamma, beta,
                       eps, scale, shift, ysum);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  if (need_sums && !inline_sums)
    hipLaunchKernelGGL(norm_fwd_sum_kernel<false>, dim3(n, cgroups), dim3(1024), 0, st, partials, tiles, c, xa, xb, hw, nullptr, nullptr, nullptr,
                       0.f, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(cgroups), dim3(256), 0, st, n, c, hw, mode, training, drop_scale, gamma, beta, eps,
                     momentum, running_mean, running_var, num_batches, xa, xb, scale, shift, need_sums ? ysum : nullptr, nullptr, 0,
                     inline_sums ? partials : nullptr, tiles);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- synchronised batch norm (data-parallel exactness)
// One process per GPU, each with its shard of the minibatch: the batch statistics must cover every rank's shard to
// reproduce a single-process run.  Forward: local (mean, M2, count) per channel -> all-gather
// (3*C floats per rank, done by the caller over RCCL) -> mia_norm_finalize_sync.  Backward: local (sum g, sum g*xhat)
// -> all-reduce (2*C floats) -> mia_norm_act_bwd_apply_sync.
__global__ void bn_local_stats_kernel(int n_img, int c, int64_t hw, const float* __restrict__ drop, const float* __restrict__ xa,
                                      const float* __restrict__ xb, float* __restrict__ local) {
  __shared__ double sh1[16][17], sh2[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double e1 = 0.0, e2 = 0.0;
  if (ch < c)
    for (int n = tl; n < n_img; n += 16) {
      const size_t idx = (size_t)n * c + ch;
      const double m = drop ? (double)drop[idx] : 1.0;
      e1 += m * xa[idx]; e2 += m * m * xb[idx];
    }
  sh1[tl][cl] = e1; sh2[tl][cl] = e2;
  __syncthreads();
  if (tl == 0 && ch < c) {
    e1 = 0.0; e2 = 0.0;
    for (int j = 0; j < 16; ++j) { e1 += sh1[j][cl]; e2 += sh2[j][cl]; }
    const double cnt = (double)n_img * (double)hw, mean = e1 / cnt;
    double m2 = e2 - cnt * mean * mean;
    if (m2 < 0.0) m2 = 0.0;
    local[ch] = (float)mean; local[c + ch] = (float)m2; local[2 * c + ch] = (float)cnt;
  }
}

extern "C" int mia_bn_sync_local_stats(const float* partials, int n, int tiles, int c, int64_t hw, const float* drop_scale,
                                       float* xa, float* xb, float* local, void* stream) {
  MIA_CHECK_ARG(partials && xa && xb && local && n > 0 && tiles > 0 && c > 0 && hw > 0, "mia_bn_sync_local_stats: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cgroups = ceil_div(c, 16);
  hipLaunchKernelGGL(norm_fwd_sum_kernel<false>, dim3(n, cgroups), dim3(1024), 0, st, partials, tiles, c, xa, xb, hw, nullptr, nullptr, nullptr,
                     0.f, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(bn_local_stats_kernel, dim3(cgroups), dim3(256), 0, st, n, c, hw, drop_scale, xa, xb, local);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_norm_finalize_sync(const float* gathered, int world, int n, int c, int64_t hw, const float* drop_scale,
                                      const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, long long* num_batches, float* xa, float* xb, float* scale,
                                      float* shift, float* ysum, void* stream) {
  MIA_CHECK_ARG(gathered && world > 0 && gamma && beta && xa && xb && scale && shift && n > 0 && c > 0 && hw > 0,
                "mia_norm_finalize_sync: bad arguments");
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, static_cast<hipStream_t>(stream), n, c, hw,
                     NORM_BATCH, 1, drop_scale, gamma, beta, eps, momentum, running_mean, running_var, num_batches, xa, xb, scale,
                     shift, ysum, gathered, world, nullptr, 0);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- vectorised per-channel reductions (C % 64 == 0)
// block = (64 channels = UPB 16-byte units) x (256/UPB pixel lanes); grid = (n*slabs, C/64).  Each thread streams
// 16-byte units of its slab, keeps EPU x 2 running sums in registers, lanes are combined through LDS.
template <typename T, bool BWD, int CG, bool TWO = false>
__global__ __launch_bounds__(256) void colreduce_vec_kernel(const T* __restrict__ a0, const T* __restrict__ a1,
                                                            const T* __restrict__ a2 /* TWO: second gradient piece, dz = a0 + a2 */,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ xa, const float* __restrict__ xb,
                                                            int64_t hw, int c, int slabs, float slope, float* __restrict__ part) {
  // CG = channels per block: 64; 96 when c is a multiple of 96 only (a whole 192-byte pixel row of cfg5's first level per block: 21 pixel
  // lanes, 4 idle threads); else 32
  constexpr int EPU = Elem<T>::EPU, UPB = CG / EPU, LANES = 256 / UPB;
  __shared__ float sh[2][LANES][CG + 1];
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int u = threadIdx.x % UPB, pl = threadIdx.x / UPB;
  const bool active = (256 % UPB == 0) || pl < LANES;  // 256 threads are not a multiple of UPB at CG = 96
  const int ch0 = blockIdx.y * CG + u * EPU;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  const size_t base = (size_t)n * hw * c + ch0;
  float s1[EPU], s2[EPU], sc[EPU], sf[EPU], ka[EPU], kb[EPU];
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    s1[e] = 0.f; s2[e] = 0.f;
    if (BWD) {
      const size_t o = (size_t)n * c + ch0 + e;
      sc[e] = scale[o]; sf[e] = shift[o]; ka[e] = xa[o]; kb[e] = xb[o];
    }
  }
  auto body = [&](const u32x4& r0v, const u32x4& r1v, const u32x4& r2v) {
    alignas(16) T v0[EPU]; alignas(16) T v1[EPU]; alignas(16) T v2[EPU];
    *reinterpret_cast<u32x4*>(v0) = r0v;
    *reinterpret_cast<u32x4*>(v1) = r1v;
    *reinterpret_cast<u32x4*>(v2) = r2v;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      if (BWD) {  // a0 (+ a2) = dz, a1 = y
        const float yv = Elem<T>::ld(v1 + e);
        float g = Elem<T>::ld(v0 + e);
        if (TWO) g += Elem<T>::ld(v2 + e);
        if (!(sc[e] * yv + sf[e] > 0.f)) g *= slope;
        s1[e] += g; s2[e] += g * (ka[e] * yv + kb[e]);
      } else {
        const float v = Elem<T>::ld(v0 + e);
        s1[e] += v; s2[e] += v * v;
      }
    }
  };
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
  auto ld = [&](const T* ptr, int64_t r) { return *reinterpret_cast<const u32x4*>(ptr + base + r * c); };
  int64_t r = active ? r0 + pl : r1;
  for (; r + LANES < r1; r += 2 * LANES) {  // two rows (up to six 16-byte loads) in flight per thread
    const u32x4 p0 = ld(a0, r), q0 = BWD ? ld(a1, r) : zero, t0 = TWO ? ld(a2, r) : zero;
    const u32x4 p1 = ld(a0, r + LANES), q1 = BWD ? ld(a1, r + LANES) : zero, t1 = TWO ? ld(a2, r + LANES) : zero;
    body(p0, q0, t0); body(p1, q1, t1);
  }
  for (; r < r1; r += LANES) body(ld(a0, r), BWD ? ld(a1, r) : zero, TWO ? ld(a2, r) : zero);
#pragma unroll
  for (int e = 0; e < EPU; ++e) if (active) { sh[0][pl][u * EPU + e] = s1[e]; sh[1][pl][u * EPU + e] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * CG) {
    const int k = threadIdx.x / CG, chl = threadIdx.x % CG;
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < LANES; ++j) t += sh[k][j][chl];
    part[(((size_t)n * slabs + s) * c + blockIdx.y * CG + chl) * 2 + k] = t;
  }
}

// ---------------------------------------------------------------- stand-alone statistics (no conv epilogue available)
// partial sums over a pixel slab for every channel: part[n][s][c][2]
template <typename T>
__global__ void norm_stats_kernel(const T* __restrict__ y, int64_t hw, int c, int slabs, float* __restrict__ part) {
  extern __shared__ float sh[];
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int cw = blockDim.x >= c ? c : blockDim.x, rows_par = blockDim.x / cw;
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  const T* base = y + (size_t)n * hw * c;
  for (int cb = blockIdx.y * cw; cb < c; cb += gridDim.y * cw) {
    const int ch = cb + tc;
    float s1 = 0.f, s2 = 0.f;
    if (ch < c && tr < rows_par)
      for (int64_t r = r0 + tr; r < r1; r += rows_par) { const float v = Elem<T>::ld(base + r * c + ch); s1 += v; s2 += v * v; }
    sh[threadIdx.x] = s1; sh[blockDim.x + threadIdx.x] = s2;
    __syncthreads();
    if (tr == 0 && ch < c) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < rows_par; ++j) { t1 += sh[j * cw + tc]; t2 += sh[blockDim.x + j * cw + tc]; }
      float* dst = part + (((size_t)n * slabs + s) * c + ch) * 2;
      dst[0] = t1; dst[1] = t2;
    }
    __syncthreads();
  }
}

// launch helper: 64-channel blocks, or 32-channel blocks when c is only a multiple of 32 (e.g. 96-channel layers);
// A2 != nullptr selects the two-piece gradient variant (dz = A0 + A2)
#define CRV_(T, BWDF, CGW, TWOF, A0, A1, A2, SC, SF, XA, XB, SLOPE)                                                       \
  hipLaunchKernelGGL((colreduce_vec_kernel<T, BWDF, CGW, TWOF>), dim3(n * slabs, c / CGW), dim3(256), 0, st, A0, A1, A2, SC, \
                     SF, XA, XB, hw, c, slabs, SLOPE, partials)
#define CRV(T, BWDF, A0, A1, A2, SC, SF, XA, XB, SLOPE)                                                                   \
  do {                                                                                                                    \
    if (c % 64 == 0) { if ((A2) != nullptr) CRV_(T, BWDF, 64, true, A0, A1, A2, SC, SF, XA, XB, SLOPE);                   \
                       else CRV_(T, BWDF, 64, false, A0, A1, A2, SC, SF, XA, XB, SLOPE); }                                \
    else if (c % 96 == 0) { if ((A2) != nullptr) CRV_(T, BWDF, 96, true, A0, A1, A2, SC, SF, XA, XB, SLOPE);              \
                            else CRV_(T, BWDF, 96, false, A0, A1, A2, SC, SF, XA, XB, SLOPE); }                           \
    else { if ((A2) != nullptr) CRV_(T, BWDF, 32, true, A0, A1, A2, SC, SF, XA, XB, SLOPE);                               \
           else CRV_(T, BWDF, 32, false, A0, A1, A2, SC, SF, XA, XB, SLOPE); }                                            \
  } while (0)

extern "C" int mia_norm_stats(const void* y, int dtype, int n, int64_t hw, int c, int slabs, float* partials, void* stream) {
  MIA_CHECK_ARG(y && partials && n > 0 && hw > 0 && c > 0 && slabs > 0, "mia_norm_stats: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c % 32 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (dtype == MIA_BF16 || dtype == MIA_F32)) {
    if (dtype == MIA_BF16)
      CRV(bf16_t, false, static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(nullptr), static_cast<const bf16_t*>(nullptr), nullptr, nullptr, nullptr, nullptr, 0.f);
    else
      CRV(float, false, static_cast<const float*>(y), static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), nullptr, nullptr, nullptr, nullptr, 0.f);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  if (dtype == MIA_BF16)
    hipLaunchKernelGGL(norm_stats_kernel<bf16_t>, dim3(n * slabs, ceil_div(c, 256)), dim3(256), 512 * sizeof(float), st, static_cast<const bf16_t*>(y), hw, c, slabs, partials);
  else if (dtype == MIA_F32)
    hipLaunchKernelGGL(norm_stats_kernel<float>, dim3(n * slabs, ceil_div(c, 256)), dim3(256), 512 * sizeof(float), st, static_cast<const float*>(y), hw, c, slabs, partials);
  else { mia_set_error("mia_norm_stats: bad dtype"); return MIA_EARG; }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Running max |x| of the values a streaming kernel WRITES (fp32 tensors; the scale source of the split-f16 convs that consume them,
// common.h SplitF16): a lane folds the fp32 bit patterns of its outputs, a wave folds its lanes, one atomic unsigned maximum per wave
// and only when the slot does not hold at least that much already.  Every thread of the BLOCK must arrive here (no early return).
// The caller hands in a ZEROED slot (or a running maximum to fold into): the kernels never reset it.
extern "C" int mia_amax(const float* x, int64_t n, void* slot, int reset, void* stream);
template <typename T> __device__ __forceinline__ void amax_fold(unsigned& m, const T* out) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < Elem<T>::EPU; ++e) { const unsigned b = __builtin_bit_cast(unsigned, out[e]) & 0x7FFFFFFFu; m = b > m ? b : m; }
  }
}
__device__ __forceinline__ void amax_publish(unsigned m, unsigned* __restrict__ slot) {
  // lanes -> wave (DPP) -> block (LDS) -> ONE coherent load + conditional atomic per block: with one per wave (131 k waves per launch
  // on one address) the forward apply pass ran 50 % slower (rocprofv3, cfg2: 70 -> 108 us average)
  __shared__ unsigned blk;
  if (threadIdx.x == 0) blk = 0u;
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(&blk, m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned b = blk;
    if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
  }
}

// ---------------------------------------------------------------- streaming apply kernels (C % EPU == 0)
// grid = (n * slabs, ceil(UPP / UPB)); a thread owns ONE 16-byte channel unit of image n, keeps that unit's
// per-(n,c) coefficients in registers and streams its slab's pixels: no per-element coefficient loads or divisions.
template <typename T>
__global__ __launch_bounds__(256) void norm_act_fwd_stream_kernel(const T* __restrict__ y, T* __restrict__ z,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  int64_t hw, int c, int slabs, int upb, float slope,
                                                                  unsigned* __restrict__ amax) {
  constexpr int EPU = Elem<T>::EPU;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int lanes = 256 / upb;
  const int u = blockIdx.y * upb + threadIdx.x % upb, pl = threadIdx.x / upb;
  const bool live = u * EPU < c;  // (lanes beyond the channel count idle through the loops: the maximum needs the whole wave at the end)
  if (!live && (sizeof(T) != 4 || amax == nullptr)) return;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = live ? (r0 + per < hw ? r0 + per : hw) : r0;
  float sc[EPU], sf[EPU];
  unsigned am = 0;
#pragma unroll
  for (int e = 0; e < EPU; ++e) { sc[e] = live ? scale[(size_t)n * c + u * EPU + e] : 0.f; sf[e] = live ? shift[(size_t)n * c + u * EPU + e] : 0.f; }
  const size_t base = (size_t)n * hw * c + (size_t)u * EPU;
  auto body = [&](const u32x4& raw, int64_t r) {
    alignas(16) T in[EPU]; alignas(16) T out[EPU];
    *reinterpret_cast<u32x4*>(in) = raw;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float v = sc[e] * Elem<T>::ld(in + e) + sf[e];
      out[e] = Elem<T>::cvt(v > 0.f ? v : v * slope);
    }
    amax_fold<T>(am, out);
    store_data_fence();
    *reinterpret_cast<u32x4*>(z + base + r * c) = *reinterpret_cast<const u32x4*>(out);
    store_data_pad();
  };
  // (non-temporal loads / stores for the > 256 MB tensors were measured: 0.403 -> 0.396 ms on the 2.1 GB launch, within noise)
  auto ld = [&](int64_t rr) -> u32x4 { return *reinterpret_cast<const u32x4*>(y + base + rr * c); };
  int64_t r = r0 + pl;
  for (; r + 3 * lanes < r1; r += 4 * lanes) {  // four independent 16-byte loads in flight per thread
    const u32x4 a0 = ld(r), a1 = ld(r + lanes), a2 = ld(r + 2 * lanes), a3 = ld(r + 3 * lanes);
    body(a0, r); body(a1, r + lanes); body(a2, r + 2 * lanes); body(a3, r + 3 * lanes);
  }
  for (; r < r1; r += lanes) body(ld(r), r);
  if constexpr (sizeof(T) == 4) { if (amax != nullptr) amax_publish(am, amax); }
}

// dy = scale*(g - c1 - xhat*c2) = scale*g + ka*y + kb  with ka = -scale*c2*xa, kb = -scale*(c1 + c2*xb)
template <typename T, bool TWO = false>
__global__ __launch_bounds__(256) void norm_act_bwd_stream_kernel(const T* __restrict__ dz, const T* __restrict__ dz2,
                                                                  const T* __restrict__ y, T* __restrict__ dy,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ xa, const float* __restrict__ xb,
                                                                  const float* __restrict__ c1, const float* __restrict__ c2,
                                                                  int64_t hw, int c, int slabs, int upb, float slope,
                                                                  unsigned* __restrict__ amax) {
  constexpr int EPU = Elem<T>::EPU;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int lanes = 256 / upb;
  const int u = blockIdx.y * upb + threadIdx.x % upb, pl = threadIdx.x / upb;
  const bool live = u * EPU < c;
  if (!live && (sizeof(T) != 4 || amax == nullptr)) return;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = live ? (r0 + per < hw ? r0 + per : hw) : r0;
  float sc[EPU], sf[EPU], ka[EPU], kb[EPU];
  unsigned am = 0;
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    const size_t o = live ? (size_t)n * c + u * EPU + e : 0;
    sc[e] = scale[o]; sf[e] = shift[o];
    ka[e] = -sc[e] * c2[o] * xa[o];
    kb[e] = -sc[e] * (c1[o] + c2[o] * xb[o]);
  }
  const size_t base = (size_t)n * hw * c + (size_t)u * EPU;
  auto body = [&](const u32x4& graw, const u32x4& g2raw, const u32x4& yraw, int64_t r) {
    alignas(16) T gin[EPU]; alignas(16) T g2in[EPU]; alignas(16) T yin[EPU]; alignas(16) T out[EPU];
    *reinterpret_cast<u32x4*>(gin) = graw;
    *reinterpret_cast<u32x4*>(g2in) = g2raw;
    *reinterpret_cast<u32x4*>(yin) = yraw;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float yv = Elem<T>::ld(yin + e);
      float g = Elem<T>::ld(gin + e);
      if (TWO) g += Elem<T>::ld(g2in + e);
      out[e] = Elem<T>::cvt(norm_bwd_dy(g, yv, sc[e], sf[e], ka[e], kb[e], slope));
    }
    amax_fold<T>(am, out);
    store_data_fence();
    *reinterpret_cast<u32x4*>(dy + base + r * c) = *reinterpret_cast<const u32x4*>(out);
    store_data_pad();
  };
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
  auto ld = [&](const T* ptr, int64_t r) { return *reinterpret_cast<const u32x4*>(ptr + base + r * c); };
  int64_t r = r0 + pl;
  for (; r + lanes < r1; r += 2 * lanes) {  // four (six) independent 16-byte loads in flight per thread
    const u32x4 g0 = ld(dz, r), h0 = TWO ? ld(dz2, r) : zero, y0 = ld(y, r);
    const u32x4 g1 = ld(dz, r + lanes), h1 = TWO ? ld(dz2, r + lanes) : zero, y1 = ld(y, r + lanes);
    body(g0, h0, y0, r); body(g1, h1, y1, r + lanes);
  }
  for (; r < r1; r += lanes) body(ld(dz, r), TWO ? ld(dz2, r) : zero, ld(y, r), r);
  if constexpr (sizeof(T) == 4) { if (amax != nullptr) amax_publish(am, amax); }
}

static void stream_geometry(int n, int64_t hw, int c, int epu, int* slabs, int* upb, int* gy) {
  const int upp = c / epu;
  int ub = 1;
  while (ub < upp && ub < 64) ub <<= 1;  // units per block (power of two <= 64): >= 4 pixel lanes
  *upb = ub;
  *gy = (upp + ub - 1) / ub;
  // ~32768 blocks in total (swept 1k .. 128k on the cfg3 step, MIA_STREAM_BLOCKS: 4096 -> 32768 blocks = 5.60 -> 5.85 TB/s on
  // the forward apply, 5.39 -> 5.55 on the backward; flat beyond), each lane streaming >= 8 pixels
  const int lanes = 256 / ub;
  const int target = mia_options().stream_blocks;
  int64_t sl = target / ((int64_t)n * *gy);
  const int64_t maxsl = hw / (lanes * 8) > 0 ? hw / (lanes * 8) : 1;
  if (sl > maxsl) sl = maxsl;
  if (sl < 1) sl = 1;
  *slabs = (int)sl;
}

// ---------------------------------------------------------------- forward apply: z = lrelu(scale*y + shift)
template <typename T, bool VEC>
__global__ void norm_act_fwd_kernel(const T* __restrict__ y, T* __restrict__ z, const float* __restrict__ scale,
                                    const float* __restrict__ shift, int64_t hw, int c, int64_t total_units, float slope) {
  constexpr int EPU = VEC ? Elem<T>::EPU : 1;
  const int upp = c / EPU;  // units per pixel
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = u / upp;
    const int ch = (int)(u - pix * upp) * EPU;
    const int n = (int)(pix / hw);
    const float* sc = scale + (size_t)n * c + ch;
    const float* sf = shift + (size_t)n * c + ch;
    if constexpr (VEC) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(y + u * EPU);
      alignas(16) T in[EPU]; alignas(16) T out[EPU];
      *reinterpret_cast<u32x4*>(in) = raw;
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        const float v = sc[e] * Elem<T>::ld(in + e) + sf[e];
        out[e] = Elem<T>::cvt(v > 0.f ? v : v * slope);
      }
      *reinterpret_cast<u32x4*>(z + u * EPU) = *reinterpret_cast<const u32x4*>(out);
    } else {
      const float v = sc[0] * Elem<T>::ld(y + u) + sf[0];
      Elem<T>::st(z + u, v > 0.f ? v : v * slope);
    }
  }
}

extern "C" int mia_norm_act_fwd(const void* y, void* z, int dtype, const float* scale, const float* shift, int n, int64_t hw,
                                int c, float slope, void* amax_out, void* stream) {
  MIA_CHECK_ARG(y && z && scale && shift && n > 0 && hw > 0 && c > 0, "mia_norm_act_fwd: bad arguments");
  if (dtype != MIA_F32) amax_out = nullptr;  // the maximum serves fp32 consumers only
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  const bool vec = (c % epu == 0) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z)) & 15) == 0;
  const int64_t units = (int64_t)n * hw * (vec ? c / epu : c);
  const int blocks = (int)((units + 255) / 256 < 16384 ? (units + 255) / 256 : 16384);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec && (dtype == MIA_BF16 || dtype == MIA_F32)) {
    int sl, upb, gy;
    stream_geometry(n, hw, c, epu, &sl, &upb, &gy);
    if (dtype == MIA_BF16)
      hipLaunchKernelGGL(norm_act_fwd_stream_kernel<bf16_t>, dim3(n * sl, gy), dim3(256), 0, st, static_cast<const bf16_t*>(y), static_cast<bf16_t*>(z), scale, shift, hw, c, sl, upb, slope, (unsigned*)nullptr);
    else
      hipLaunchKernelGGL(norm_act_fwd_stream_kernel<float>, dim3(n * sl, gy), dim3(256), 0, st, static_cast<const float*>(y), static_cast<float*>(z), scale, shift, hw, c, sl, upb, slope, static_cast<unsigned*>(amax_out));
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
#define NA(T, V) hipLaunchKernelGGL((norm_act_fwd_kernel<T, V>), dim3(blocks), dim3(256), 0, st, static_cast<const T*>(y), \
                                    static_cast<T*>(z), scale, shift, hw, c, units, slope)
  if (dtype == MIA_BF16) { if (vec) NA(bf16_t, true); else NA(bf16_t, false); }
  else if (dtype == MIA_F32) { if (vec) NA(float, true); else NA(float, false); }
  else { mia_set_error("mia_norm_act_fwd: bad dtype"); return MIA_EARG; }
#undef NA
  MIA_LAUNCH_CHECK();
  if (amax_out) return mia_amax(static_cast<const float*>(z), (int64_t)n * hw * c, amax_out, 0, stream);  // scalar fallback shapes: a separate pass
  return MIA_OK;
}

// ---------------------------------------------------------------- backward pass 1: per-(n, slab, c) sums of g and g*xhat
template <typename T>
__global__ void norm_act_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ scale,
                                           const float* __restrict__ shift, const float* __restrict__ xa,
                                           const float* __restrict__ xb, int64_t hw, int c, int slabs, float slope,
                                           float* __restrict__ part) {
  extern __shared__ float sh[];
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int cw = blockDim.x >= c ? c : blockDim.x, rows_par = blockDim.x / cw;
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  const size_t base = (size_t)n * hw * c;
  for (int cb = blockIdx.y * cw; cb < c; cb += gridDim.y * cw) {
    const int ch = cb + tc;
    float s1 = 0.f, s2 = 0.f;
    if (ch < c && tr < rows_par) {
      const float sc = scale[(size_t)n * c + ch], sf = shift[(size_t)n * c + ch];
      const float a = xa[(size_t)n * c + ch], b = xb[(size_t)n * c + ch];
      for (int64_t r = r0 + tr; r < r1; r += rows_par) {
        const float yv = Elem<T>::ld(y + base + r * c + ch);
        float g = Elem<T>::ld(dz + base + r * c + ch);
        if (!(sc * yv + sf > 0.f)) g *= slope;
        s1 += g; s2 += g * (a * yv + b);
      }
    }
    sh[threadIdx.x] = s1; sh[blockDim.x + threadIdx.x] = s2;
    __syncthreads();
    if (tr == 0 && ch < c) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < rows_par; ++j) { t1 += sh[j * cw + tc]; t2 += sh[blockDim.x + j * cw + tc]; }
      float* dst = part + (((size_t)n * slabs + s) * c + ch) * 2;
      dst[0] = t1; dst[1] = t2;
    }
    __syncthreads();
  }
}

// pass 1b (two tiny launches, fully parallel, fixed summation order):
//  K1  grid (image, 16-channel group): the 16 lanes of each channel sweep the slab partials -> per-(n,c) sums
//      Sg = sum g, Sgx = sum g*xhat, parked in c1 / c2.
//  K2  grid (16-channel group): lanes sweep images -> dgamma = sum_n Sgx, dbeta = sum_n Sg, the group means
//      c1 = mean(g), c2 = mean(g*xhat) (per image for instance norm, over the batch for batch norm; 0 for frozen
//      statistics), and the gradient of the conv bias in front of the norm in closed form:
//        sum_p dy = scale * (Sg - M*c1 - c2*Sxhat),  Sxhat = xa*Sy + M*xb   (M = H*W)
//      (analytically 0 for instance norm -- the reference's autograd produces rounding noise there too).
__global__ void norm_bwd_sum_kernel(const float* __restrict__ part, int slabs, int c, float* __restrict__ c1,
                                    float* __restrict__ c2) {
  __shared__ double sh1[16][17], sh2[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int n = blockIdx.x, ch = blockIdx.y * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (ch < c)
    for (int s = tl; s < slabs; s += 16) {
      const float* p = part + (((size_t)n * slabs + s) * c + ch) * 2;
      s1 += p[0]; s2 += p[1];
    }
  sh1[tl][cl] = s1; sh2[tl][cl] = s2;
  __syncthreads();
  if (tl == 0 && ch < c) {
    s1 = 0.0; s2 = 0.0;
    for (int j = 0; j < 16; ++j) { s1 += sh1[j][cl]; s2 += sh2[j][cl]; }
    c1[(size_t)n * c + ch] = (float)s1; c2[(size_t)n * c + ch] = (float)s2;
  }
}

__global__ void norm_bwd_finalize_kernel(int n_img, int c, int64_t hw, int mode, int fixed_stats, const float* __restrict__ scale,
                                         const float* __restrict__ xa, const float* __restrict__ xb, const float* __restrict__ ysum,
                                         float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ dgamma,
                                         float* __restrict__ dbeta, float* __restrict__ dbias, int accumulate,
                                         const float* __restrict__ group_tot, const float* __restrict__ part, int slabs) {
  __shared__ double sh1[16][17], sh2[16][17], sh3[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  const double M = (double)hw;
  if (part != nullptr && ch < c) {  // short sums: norm_bwd_sum_kernel's work done inline (see norm_finalize_kernel)
    for (int n = tl; n < n_img; n += 16) {
      double s1 = 0.0, s2 = 0.0;
      for (int t = 0; t < slabs; ++t) {
        const float* p = part + (((size_t)n * slabs + t) * c + ch) * 2;
        s1 += p[0]; s2 += p[1];
      }
      c1[(size_t)n * c + ch] = (float)s1; c2[(size_t)n * c + ch] = (float)s2;
    }
  }
  double tg = 0.0, tgx = 0.0;
  if (ch < c)
    for (int n = tl; n < n_img; n += 16) { tg += c1[(size_t)n * c + ch]; tgx += c2[(size_t)n * c + ch]; }
  sh1[tl][cl] = tg; sh2[tl][cl] = tgx;
  __syncthreads();
  tg = 0.0; tgx = 0.0;
  for (int j = 0; j < 16; ++j) { tg += sh1[j][cl]; tgx += sh2[j][cl]; }
  __syncthreads();
  const double cnt = (double)n_img * M;
  double tb = 0.0;
  if (ch < c)
    for (int n = tl; n < n_img; n += 16) {
      const size_t idx = (size_t)n * c + ch;
      const double sg = c1[idx], sgx = c2[idx];
      double m1 = 0.0, m2 = 0.0;
      if (!fixed_stats) {
        if (mode == NORM_INSTANCE) { m1 = sg / M; m2 = sgx / M; }
        else if (group_tot) { const double gc = group_tot[2 * c + ch]; m1 = (double)group_tot[ch] / gc; m2 = (double)group_tot[c + ch] / gc; }
        else { m1 = tg / cnt; m2 = tgx / cnt; }
      }
      c1[idx] = (float)m1; c2[idx] = (float)m2;
      if (dbias) {
        const double sxh = ysum ? (double)xa[idx] * ysum[idx] + M * xb[idx] : 0.0;
        tb += (double)scale[idx] * (sg - M * m1 - m2 * sxh);
      }
    }
  sh3[tl][cl] = tb;
  __syncthreads();
  if (tl == 0 && ch < c) {
    tb = 0.0;
    for (int j = 0; j < 16; ++j) tb += sh3[j][cl];
    dgamma[ch] = accumulate ? dgamma[ch] + (float)tgx : (float)tgx;
    dbeta[ch] = accumulate ? dbeta[ch] + (float)tg : (float)tg;
    if (dbias) dbias[ch] = accumulate ? dbias[ch] + (float)tb : (float)tb;
  }
}

// pass 2: dy = xa*gamma*(g - c1 - xhat*c2)
template <typename T, bool VEC>
__global__ void norm_act_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y, T* __restrict__ dy,
                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ xa, const float* __restrict__ xb,
                                          const float* __restrict__ c1, const float* __restrict__ c2, int64_t hw, int c,
                                          int64_t total_units, float slope) {
  constexpr int EPU = VEC ? Elem<T>::EPU : 1;
  const int upp = c / EPU;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = u / upp;
    const int ch = (int)(u - pix * upp) * EPU;
    const size_t o = (size_t)(pix / hw) * c + ch;
    alignas(16) T gin[EPU]; alignas(16) T yin[EPU]; alignas(16) T out[EPU];
    if constexpr (VEC) {
      *reinterpret_cast<u32x4*>(gin) = *reinterpret_cast<const u32x4*>(dz + u * EPU);
      *reinterpret_cast<u32x4*>(yin) = *reinterpret_cast<const u32x4*>(y + u * EPU);
    } else { gin[0] = dz[u]; yin[0] = y[u]; }
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float yv = Elem<T>::ld(yin + e);
      float g = Elem<T>::ld(gin + e);
      if (!(scale[o + e] * yv + shift[o + e] > 0.f)) g *= slope;
      const float xh = xa[o + e] * yv + xb[o + e];
      // scale = gamma*xa
      out[e] = Elem<T>::cvt(scale[o + e] * (g - c1[o + e] - xh * c2[o + e]));
    }
    if constexpr (VEC) *reinterpret_cast<u32x4*>(dy + u * EPU) = *reinterpret_cast<const u32x4*>(out);
    else dy[u] = out[0];
  }
}

static bool bwd_vec_ok(const void* dz, const void* y, const void* dy, int dtype, int c) {
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  return (c % epu == 0) &&
         ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
}

// pass 1: slab partials of (sum g, sum g*xhat), then per-(n,c) sums parked in c1 / c2
static void bwd_reduce_launch(const void* dz, const void* dz2, const void* y, int dtype, const float* scale, const float* shift, const float* xa,
                              const float* xb, int n, int64_t hw, int c, float slope, int slabs, float* partials, float* c1,
                              float* c2, hipStream_t st, bool do_sum = true) {
  const bool vec = bwd_vec_ok(dz, y, nullptr, dtype, c);
#define RD(T) hipLaunchKernelGGL(norm_act_bwd_reduce_kernel<T>, dim3(n * slabs, ceil_div(c, 256)), dim3(256), 512 * sizeof(float), st,   \
                                 static_cast<const T*>(dz), static_cast<const T*>(y), scale, shift, xa, xb, hw, c, slabs, \
                                 slope, partials)
  if (vec && c % 32 == 0) {
    if (dtype == MIA_BF16)
      CRV(bf16_t, true, static_cast<const bf16_t*>(dz), static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(dz2), scale, shift, xa, xb, slope);
    else
      CRV(float, true, static_cast<const float*>(dz), static_cast<const float*>(y), static_cast<const float*>(dz2), scale, shift, xa, xb, slope);
  } else if (dtype == MIA_BF16) RD(bf16_t); else RD(float);
#undef RD
  if (do_sum) hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3(n, ceil_div(c, 16)), dim3(256), 0, st, partials, slabs, c, c1, c2);
}

// pass 2: dy from the finalized group means
static void bwd_apply_launch(const void* dz, const void* dz2, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                             const float* xa, const float* xb, const float* c1, const float* c2, int n, int64_t hw, int c,
                             float slope, hipStream_t st, void* amax_out = nullptr) {
  // amax_out (fp32 only): max |dy| for the split-f16 convs that consume dy -- folded into the streaming kernel, a separate pass otherwise
  if (dtype != MIA_F32) amax_out = nullptr;
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  const bool vec = bwd_vec_ok(dz, y, dy, dtype, c);
  const int64_t units = (int64_t)n * hw * (vec ? c / epu : c);
  const int blocks = (int)((units + 255) / 256 < 16384 ? (units + 255) / 256 : 16384);
#define AP(T, V) hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, V>), dim3(blocks), dim3(256), 0, st,                   \
                                    static_cast<const T*>(dz), static_cast<const T*>(y), static_cast<T*>(dy), scale, shift, \
                                    xa, xb, c1, c2, hw, c, units, slope)
  if (vec) {
    int sl, upb, gy;
    stream_geometry(n, hw, c, epu, &sl, &upb, &gy);
#define BS(T, TWOF) hipLaunchKernelGGL((norm_act_bwd_stream_kernel<T, TWOF>), dim3(n * sl, gy), dim3(256), 0, st, static_cast<const T*>(dz), \
                                       static_cast<const T*>(dz2), static_cast<const T*>(y), static_cast<T*>(dy), scale, shift, xa, xb, \
                                       c1, c2, hw, c, sl, upb, slope, static_cast<unsigned*>(amax_out))
    if (dtype == MIA_BF16) { if (dz2) BS(bf16_t, true); else BS(bf16_t, false); }
    else { if (dz2) BS(float, true); else BS(float, false); }
#undef BS
  } else {
    if (dtype == MIA_BF16) AP(bf16_t, false);
    else AP(float, false);
    if (amax_out) (void)mia_amax(static_cast<const float*>(dy), (int64_t)n * hw * c, amax_out, 0, st);
  }
#undef AP
}

// dz2 (optional): second piece of the output gradient, dz = dz + dz2 summed on load in fp32 -- a skip tensor's two
// consumers (unet.py:213 and the next encoder level) each deliver their own gradient and no `add` pass is needed.
// Needs the vectorised path: c % 32 == 0 and 16-byte aligned tensors (mia_norm_two_piece_ok).
extern "C" int mia_norm_two_piece_ok(int dtype, int c) { return (dtype == MIA_BF16 || dtype == MIA_F32) && c % 32 == 0; }
static bool two_piece_ok(const void* dz, const void* dz2, const void* y, const void* dy, int dtype, int c) {
  return dz2 == nullptr || (c % 32 == 0 && bwd_vec_ok(dz, y, dy, dtype, c) && (reinterpret_cast<uintptr_t>(dz2) & 15) == 0);
}

extern "C" int mia_norm_act_bwd(const void* dz, const void* dz2, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                                const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                                int fixed_stats, float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma,
                                float* dbeta, float* dbias, int accumulate, void* amax_out, void* stream) {
  MIA_CHECK_ARG(dz && y && dy && scale && shift && xa && xb && partials && c1 && c2 && dgamma && dbeta,
                "mia_norm_act_bwd: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && c > 0 && slabs > 0, "mia_norm_act_bwd: bad shape");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_act_bwd: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  MIA_CHECK_ARG(two_piece_ok(dz, dz2, y, dy, dtype, c), "mia_norm_act_bwd: two-piece gradient needs c %% 32 == 0 and aligned tensors");
  const bool inline_sums = (int64_t)n * slabs <= 1024;  // short slab sums are folded into the finalize launch
  bwd_reduce_launch(dz, dz2, y, dtype, scale, shift, xa, xb, n, hw, c, slope, slabs, partials, c1, c2, st, !inline_sums);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, mode, fixed_stats, scale, xa,
                     xb, ysum, c1, c2, dgamma, dbeta, dbias, accumulate, nullptr, inline_sums ? partials : nullptr, slabs);
  bwd_apply_launch(dz, dz2, y, dy, dtype, scale, shift, xa, xb, c1, c2, n, hw, c, slope, st, amax_out);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// mia_norm_act_bwd without its apply pass: the reduction over (dz, y) and the finalize (c1, c2 = the group means of g and
// g*xhat; dgamma, dbeta, dbias) -- for a block whose only consumer of dy forms it on load (mia_stem_wgrad_fused).
extern "C" int mia_norm_bwd_sums(const void* dz, const void* dz2, const void* y, int dtype, const float* scale, const float* shift,
                                 const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                                 int fixed_stats, float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma,
                                 float* dbeta, float* dbias, int accumulate, void* stream) {
  MIA_CHECK_ARG(dz && y && scale && shift && xa && xb && partials && c1 && c2 && dgamma && dbeta, "mia_norm_bwd_sums: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && c > 0 && slabs > 0, "mia_norm_bwd_sums: bad shape");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_bwd_sums: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  MIA_CHECK_ARG(two_piece_ok(dz, dz2, y, nullptr, dtype, c), "mia_norm_bwd_sums: two-piece gradient needs c %% 32 == 0 and aligned tensors");
  const bool inline_sums = (int64_t)n * slabs <= 1024;
  bwd_reduce_launch(dz, dz2, y, dtype, scale, shift, xa, xb, n, hw, c, slope, slabs, partials, c1, c2, st, !inline_sums);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, mode, fixed_stats, scale, xa,
                     xb, ysum, c1, c2, dgamma, dbeta, dbias, accumulate, nullptr, inline_sums ? partials : nullptr, slabs);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

#ifdef MIA_EXPERIMENTS  // consumer side of the column-reduce epilogue (probe builds only)
// mia_norm_act_bwd whose reduction pass already ran somewhere else: `partials` [n][parts][c][2] were filled by the epilogue of the
// input-gradient conv that produced dz (mia_conv_mma_cr).  Sums + finalize here; the apply pass only when dy != nullptr.
extern "C" int mia_norm_act_bwd_pre(const void* dz, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                                    const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                                    int fixed_stats, float slope, int parts, const float* partials, float* c1, float* c2,
                                    float* dgamma, float* dbeta, float* dbias, int accumulate, void* amax_out, void* stream) {
  MIA_CHECK_ARG(scale && shift && xa && xb && partials && c1 && c2 && dgamma && dbeta, "mia_norm_act_bwd_pre: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && c > 0 && parts > 0, "mia_norm_act_bwd_pre: bad shape");
  MIA_CHECK_ARG(dy == nullptr || (dz && y), "mia_norm_act_bwd_pre: the apply pass needs dz and y");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_act_bwd_pre: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3(n, ceil_div(c, 16)), dim3(256), 0, st, partials, parts, c, c1, c2);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, mode, fixed_stats, scale, xa,
                     xb, ysum, c1, c2, dgamma, dbeta, dbias, accumulate, nullptr, nullptr, 0);
  if (dy != nullptr) bwd_apply_launch(dz, nullptr, y, dy, dtype, scale, shift, xa, xb, c1, c2, n, hw, c, slope, st, amax_out);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

#endif

// local batch totals tot[3][C] = (sum g, sum g*xhat, pixel count) over this rank's images, from the per-(n,c) sums in c1 / c2
__global__ void bn_bwd_local_tot_kernel(int n_img, int c, int64_t hw, const float* __restrict__ c1, const float* __restrict__ c2,
                                        float* __restrict__ tot) {
  __shared__ double sh1[16][17], sh2[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  double tg = 0.0, tgx = 0.0;
  if (ch < c)
    for (int n = tl; n < n_img; n += 16) { tg += c1[(size_t)n * c + ch]; tgx += c2[(size_t)n * c + ch]; }
  sh1[tl][cl] = tg; sh2[tl][cl] = tgx;
  __syncthreads();
  if (tl == 0 && ch < c) {
    tg = 0.0; tgx = 0.0;
    for (int j = 0; j < 16; ++j) { tg += sh1[j][cl]; tgx += sh2[j][cl]; }
    tot[ch] = (float)tg; tot[c + ch] = (float)tgx; tot[2 * c + ch] = (float)((double)n_img * (double)hw);
  }
}

extern "C" int mia_norm_act_bwd_reduce(const void* dz, const void* dz2, const void* y, int dtype, const float* scale, const float* shift,
                                       const float* xa, const float* xb, int n, int64_t hw, int c, float slope, int slabs,
                                       float* partials, float* c1, float* c2, float* tot, void* stream) {
  MIA_CHECK_ARG(dz && y && scale && shift && xa && xb && partials && c1 && c2 && tot, "mia_norm_act_bwd_reduce: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && c > 0 && slabs > 0, "mia_norm_act_bwd_reduce: bad shape");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_act_bwd_reduce: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  MIA_CHECK_ARG(two_piece_ok(dz, dz2, y, nullptr, dtype, c), "mia_norm_act_bwd_reduce: two-piece gradient needs c %% 32 == 0 and aligned tensors");
  bwd_reduce_launch(dz, dz2, y, dtype, scale, shift, xa, xb, n, hw, c, slope, slabs, partials, c1, c2, st);
  hipLaunchKernelGGL(bn_bwd_local_tot_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, c1, c2, tot);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_norm_act_bwd_apply_sync(const void* dz, const void* dz2, const void* y, void* dy, int dtype, const float* scale,
                                           const float* shift, const float* xa, const float* xb, const float* ysum, int n,
                                           int64_t hw, int c, float slope, float* c1, float* c2, const float* group_tot,
                                           float* dgamma, float* dbeta, float* dbias, int accumulate, void* amax_out, void* stream) {
  MIA_CHECK_ARG(dz && y && dy && scale && shift && xa && xb && c1 && c2 && group_tot && dgamma && dbeta,
                "mia_norm_act_bwd_apply_sync: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && c > 0, "mia_norm_act_bwd_apply_sync: bad shape");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_act_bwd_apply_sync: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, NORM_BATCH, 0, scale, xa, xb,
                     ysum, c1, c2, dgamma, dbeta, dbias, accumulate, group_tot, nullptr, 0);
  MIA_CHECK_ARG(two_piece_ok(dz, dz2, y, dy, dtype, c), "mia_norm_act_bwd_apply_sync: two-piece gradient needs c %% 32 == 0 and aligned tensors");
  bwd_apply_launch(dz, dz2, y, dy, dtype, scale, shift, xa, xb, c1, c2, n, hw, c, slope, st, amax_out);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- norm backward fed by the 1x1 head
// When the block's only consumer is the segmentation head, its output gradient is dz[p][c] = sum_k dl[p][k] * w[k][c]:
// three FMAs per value from 12 bytes per pixel.  Recomputing it here (weights in registers) removes the head's
// input-gradient kernel and both reads of dz: nothing activation-sized flows between the head and this block.
// HWG (round 4): the same pass also accumulates the head's weight / bias gradient dW[k][c] = sum_p dl[p][k] * lrelu(scale*y + shift),
// db[k] = sum_p dl[p][k] -- both kernels read exactly (dl, y), so mia_head_norm_wgrad's pass over them disappears.  Needs c == CG
// (one channel group per block row); wpart: [gridDim.x][K1][c + 1].
template <typename T, int K1, int CG, bool HWG = false>
__global__ __launch_bounds__(256) void colreduce_head_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                             const T* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ xa,
                                                             const float* __restrict__ xb, int hw, int c, int slabs, float slope,
                                                             int64_t gsn, int64_t gsp, int64_t gsk, float* __restrict__ part, float* __restrict__ wpart) {
  constexpr int EPU = Elem<T>::EPU, UPB = CG / EPU, LANES = 256 / UPB;
  __shared__ float sh[2][LANES][CG + 1];
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int u = threadIdx.x % UPB, pl = threadIdx.x / UPB;
  const int ch0 = blockIdx.y * CG + u * EPU;
  const int per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float s1[EPU], s2[EPU], sc[EPU], sf[EPU], ka[EPU], kb[EPU], wr[K1][EPU];
  float wacc[HWG ? K1 : 1][EPU], bacc[HWG ? K1 : 1];
#pragma unroll
  for (int k = 0; k < (HWG ? K1 : 1); ++k) {
    bacc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) wacc[k][e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    s1[e] = 0.f; s2[e] = 0.f;
    const size_t o = (size_t)n * c + ch0 + e;
    sc[e] = scale[o]; sf[e] = shift[o]; ka[e] = xa[o]; kb[e] = xb[o];
#pragma unroll
    for (int k = 0; k < K1; ++k) wr[k][e] = w[k * c + ch0 + e];
  }
  const T* yb = y + (size_t)n * hw * c + ch0;
  const float* gb = dl + (int64_t)n * gsn;
  auto body = [&](const u32x4& raw, const float (&gv)[K1]) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = raw;
    if constexpr (HWG) {
#pragma unroll
      for (int k = 0; k < K1; ++k) bacc[k] += gv[k];
    }
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float yv = Elem<T>::ld(v + e);
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) g += gv[k] * wr[k][e];
      const float uv = sc[e] * yv + sf[e];
      if (!(uv > 0.f)) g *= slope;
      s1[e] += g; s2[e] += g * (ka[e] * yv + kb[e]);
      if constexpr (HWG) {  // the head's input x = lrelu(u), as mia_head_norm_wgrad forms it
        const float xv = uv > 0.f ? uv : uv * slope;
#pragma unroll
        for (int k = 0; k < K1; ++k) wacc[k][e] += gv[k] * xv;
      }
    }
  };
  int r = r0 + pl;
  for (; r + 3 * LANES < r1; r += 4 * LANES) {  // four pixels' loads in flight per thread (one was 2.3 TB/s on the cfg3 launch)
    u32x4 raw[4];
    float gv[4][K1];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const u32x4*>(yb + (size_t)(r + j * LANES) * c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < K1; ++k) gv[j][k] = gb[(int64_t)(r + j * LANES) * gsp + k * gsk];
#pragma unroll
    for (int j = 0; j < 4; ++j) body(raw[j], gv[j]);
  }
  for (; r < r1; r += LANES) {
    float gv[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) gv[k] = gb[(int64_t)r * gsp + k * gsk];
    body(*reinterpret_cast<const u32x4*>(yb + (size_t)r * c), gv);
  }
#pragma unroll
  for (int e = 0; e < EPU; ++e) { sh[0][pl][u * EPU + e] = s1[e]; sh[1][pl][u * EPU + e] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * CG) {
    const int k = threadIdx.x / CG, chl = threadIdx.x % CG;
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < LANES; ++j) t += sh[k][j][chl];
    part[(((size_t)n * slabs + s) * c + blockIdx.y * CG + chl) * 2 + k] = t;
  }
  if constexpr (HWG) {  // per-block head gradient partials: [K1][c + 1] (the bias column last), lanes combined through the same LDS rows
#pragma unroll
    for (int k = 0; k < K1; ++k) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPU; ++e) sh[0][pl][u * EPU + e] = wacc[k][e];
      if (u == 0) sh[0][pl][CG] = bacc[k];
      __syncthreads();
      if (threadIdx.x <= CG) {
        float t = 0.f;
#pragma unroll 8
        for (int j = 0; j < LANES; ++j) t += sh[0][j][threadIdx.x];
        wpart[((size_t)blockIdx.x * K1 + k) * (CG + 1) + threadIdx.x] = t;
      }
    }
  }
}

// dw[k][c] / db[k] (+)= sum over blocks of the partials above (fixed order)
__global__ void head_w_final_kernel(const float* __restrict__ part, int nblk, int k1, int c0, float* __restrict__ dw, float* __restrict__ db,
                                    int accumulate) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl, tot = k1 * (c0 + 1);
  float sm = 0.f;
  if (i < tot)
    for (int b = tl; b < nblk; b += 16) sm += part[(size_t)b * tot + i];
  sh[tl][cl] = sm;
  __syncthreads();
  if (tl == 0 && i < tot) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += sh[j][cl];
    const int kk = i / (c0 + 1), ch = i % (c0 + 1);
    if (ch < c0) dw[kk * c0 + ch] = accumulate ? dw[kk * c0 + ch] + t : t;
    else db[kk] = accumulate ? db[kk] + t : t;
  }
}



template <typename T, int K1>
__global__ __launch_bounds__(256) void norm_act_bwd_stream_head_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                       const T* __restrict__ y, T* __restrict__ dy,
                                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                                       const float* __restrict__ xa, const float* __restrict__ xb,
                                                                       const float* __restrict__ c1, const float* __restrict__ c2,
                                                                       int hw, int c, int slabs, int upb, float slope, int64_t gsn,
                                                                       int64_t gsp, int64_t gsk, unsigned* __restrict__ amax) {
  constexpr int EPU = Elem<T>::EPU;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int lanes = 256 / upb;
  const int u = blockIdx.y * upb + threadIdx.x % upb, pl = threadIdx.x / upb;
  const bool live = u * EPU < c;
  if (!live && (sizeof(T) != 4 || amax == nullptr)) return;
  const int per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = live ? (r0 + per < hw ? r0 + per : hw) : r0;
  float sc[EPU], sf[EPU], ka[EPU], kb[EPU], wr[K1][EPU];
  unsigned am = 0;
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    const size_t o = live ? (size_t)n * c + u * EPU + e : 0;
    sc[e] = scale[o]; sf[e] = shift[o];
    ka[e] = -sc[e] * c2[o] * xa[o];
    kb[e] = -sc[e] * (c1[o] + c2[o] * xb[o]);
#pragma unroll
    for (int k = 0; k < K1; ++k) wr[k][e] = live ? w[k * c + u * EPU + e] : 0.f;
  }
  const size_t base = (size_t)n * hw * c + (size_t)u * EPU;
  const float* gb = dl + (int64_t)n * gsn;
  auto body = [&](const u32x4& raw, const float (&gv)[K1], int r) {
    alignas(16) T yin[EPU]; alignas(16) T out[EPU];
    *reinterpret_cast<u32x4*>(yin) = raw;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float yv = Elem<T>::ld(yin + e);
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) g += gv[k] * wr[k][e];
      if (!(sc[e] * yv + sf[e] > 0.f)) g *= slope;
      out[e] = Elem<T>::cvt(sc[e] * g + ka[e] * yv + kb[e]);
    }
    amax_fold<T>(am, out);
    store_data_fence();
    *reinterpret_cast<u32x4*>(dy + base + (size_t)r * c) = *reinterpret_cast<const u32x4*>(out);
    store_data_pad();
  };
  int r = r0 + pl;
  for (; r + 3 * lanes < r1; r += 4 * lanes) {  // four pixels' loads in flight per thread
    u32x4 raw[4];
    float gv[4][K1];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const u32x4*>(y + base + (size_t)(r + j * lanes) * c);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < K1; ++k) gv[j][k] = gb[(int64_t)(r + j * lanes) * gsp + k * gsk];
#pragma unroll
    for (int j = 0; j < 4; ++j) body(raw[j], gv[j], r + j * lanes);
  }
  for (; r < r1; r += lanes) {
    float gv[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) gv[k] = gb[(int64_t)r * gsp + k * gsk];
    body(*reinterpret_cast<const u32x4*>(y + base + (size_t)r * c), gv, r);
  }
  if constexpr (sizeof(T) == 4) { if (amax != nullptr) amax_publish(am, amax); }
}

// mia_norm_act_bwd with dz = W^T dl recomputed on the fly (w: [k1][c] fp32, dl: fp32 logits gradient with element strides
// gsn / gsk / gsp, pixel-linear).  Contract: c % 32 == 0, 2 <= k1 <= 4, 16-byte aligned y / dy, hw < 2^31.
static int norm_act_bwd_head_run(const float* dlogits, const float* w, int k1, int64_t gsn, int64_t gsk, int64_t gsp,
                                 const void* y, void* dy, int dtype, const float* scale, const float* shift, const float* xa,
                                 const float* xb, const float* ysum, int n, int64_t hw, int c, int mode, int fixed_stats,
                                 float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma, float* dbeta,
                                 float* dbias, int accumulate, void* stream, float* wpart, float* dwh, float* dbh, int acc_head,
                                 void* amax_out) {
  MIA_CHECK_ARG(dlogits && w && y && dy && scale && shift && xa && xb && partials && c1 && c2 && dgamma && dbeta,
                "mia_norm_act_bwd_head: null pointer");
  MIA_CHECK_ARG(n > 0 && hw > 0 && hw < ((int64_t)1 << 31) && c > 0 && slabs > 0 && k1 >= 2 && k1 <= 4 && c % 32 == 0,
                "mia_norm_act_bwd_head: bad shape (c=%d k1=%d)", c, k1);
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_norm_act_bwd_head: bad dtype"); return MIA_EARG; }
  MIA_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, "mia_norm_act_bwd_head: unaligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int epu = dtype == MIA_BF16 ? 8 : 4;
#define CRH(T, K, CGW, HG) hipLaunchKernelGGL((colreduce_head_kernel<T, K, CGW, HG>), dim3(n * slabs, c / CGW), dim3(256), 0, st, dlogits, w, \
                                              static_cast<const T*>(y), scale, shift, xa, xb, (int)hw, c, slabs, slope, gsn, gsp, gsk, partials, wpart)
#define CRHK(T, CGW, HG) do { if (k1 == 2) CRH(T, 2, CGW, HG); else if (k1 == 3) CRH(T, 3, CGW, HG); else CRH(T, 4, CGW, HG); } while (0)
  if (wpart != nullptr) {  // c == 64 (checked by the caller): the head's weight gradient rides in the same pass
    if (dtype == MIA_BF16) CRHK(bf16_t, 64, true); else CRHK(float, 64, true);
    hipLaunchKernelGGL(head_w_final_kernel, dim3(ceil_div(k1 * (c + 1), 16)), dim3(256), 0, st, wpart, n * slabs, k1, c, dwh, dbh, acc_head);
  }
  else if (c % 64 == 0) { if (dtype == MIA_BF16) CRHK(bf16_t, 64, false); else CRHK(float, 64, false); }
  else { if (dtype == MIA_BF16) CRHK(bf16_t, 32, false); else CRHK(float, 32, false); }
#undef CRHK
#undef CRH
  const bool inline_sums = (int64_t)n * slabs <= 1024;
  if (!inline_sums) hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3(n, ceil_div(c, 16)), dim3(256), 0, st, partials, slabs, c, c1, c2);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, n, c, hw, mode, fixed_stats, scale, xa,
                     xb, ysum, c1, c2, dgamma, dbeta, dbias, accumulate, nullptr, inline_sums ? partials : nullptr, slabs);
  int sl, upb, gy;
  stream_geometry(n, hw, c, epu, &sl, &upb, &gy);
  if (dtype != MIA_F32) amax_out = nullptr;
#define BSH(T, K) hipLaunchKernelGGL((norm_act_bwd_stream_head_kernel<T, K>), dim3(n * sl, gy), dim3(256), 0, st, dlogits, w, \
                                     static_cast<const T*>(y), static_cast<T*>(dy), scale, shift, xa, xb, c1, c2, (int)hw, c, sl, upb, \
                                     slope, gsn, gsp, gsk, static_cast<unsigned*>(amax_out))
#define BSHK(T) do { if (k1 == 2) BSH(T, 2); else if (k1 == 3) BSH(T, 3); else BSH(T, 4); } while (0)
  if (dtype == MIA_BF16) BSHK(bf16_t); else BSHK(float);
#undef BSHK
#undef BSH
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_norm_act_bwd_head(const float* dlogits, const float* w, int k1, int64_t gsn, int64_t gsk, int64_t gsp,
                                     const void* y, void* dy, int dtype, const float* scale, const float* shift, const float* xa,
                                     const float* xb, const float* ysum, int n, int64_t hw, int c, int mode, int fixed_stats,
                                     float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma, float* dbeta,
                                     float* dbias, int accumulate, void* amax_out, void* stream) {
  return norm_act_bwd_head_run(dlogits, w, k1, gsn, gsk, gsp, y, dy, dtype, scale, shift, xa, xb, ysum, n, hw, c, mode, fixed_stats, slope,
                               slabs, partials, c1, c2, dgamma, dbeta, dbias, accumulate, stream, nullptr, nullptr, nullptr, 0, amax_out);
}

// mia_norm_act_bwd_head + mia_head_norm_wgrad in one reduction pass: the head's dW / db are accumulated by the kernel that computes the
// block's norm-backward sums (both read exactly dl and y).  c == 64; head_workspace: n * slabs * k1 * (c + 1) floats.
extern "C" int mia_head_w_supported(int dtype, int c, int k1) {
  return ((dtype == MIA_BF16 || dtype == MIA_F32) && c == 64 && k1 >= 2 && k1 <= 4) ? 1 : 0;
}
extern "C" int mia_norm_act_bwd_head_w(const float* dlogits, const float* w, int k1, int64_t gsn, int64_t gsk, int64_t gsp,
                                       const void* y, void* dy, int dtype, const float* scale, const float* shift, const float* xa,
                                       const float* xb, const float* ysum, int n, int64_t hw, int c, int mode, int fixed_stats,
                                       float slope, int slabs, float* partials, float* c1, float* c2, float* dgamma, float* dbeta,
                                       float* dbias, int accumulate, float* head_workspace, float* dw_head, float* db_head,
                                       int accumulate_head, void* amax_out, void* stream) {
  MIA_CHECK_ARG(head_workspace && dw_head && db_head, "mia_norm_act_bwd_head_w: null head-gradient pointer");
  MIA_CHECK_ARG(mia_head_w_supported(dtype, c, k1), "mia_norm_act_bwd_head_w: unsupported shape (c=%d k1=%d)", c, k1);
  return norm_act_bwd_head_run(dlogits, w, k1, gsn, gsk, gsp, y, dy, dtype, scale, shift, xa, xb, ysum, n, hw, c, mode, fixed_stats, slope,
                               slabs, partials, c1, c2, dgamma, dbeta, dbias, accumulate, stream, head_workspace, dw_head, db_head,
                               accumulate_head, amax_out);
}
