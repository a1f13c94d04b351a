// Weight packing, layout conversion, casts, error plumbing (gfx950).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";
void mia_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mia_last_error(void) { return g_err; }
extern "C" int mia_version(void) { return 100; }

// src: fp32 [D0][D1][taps]  ->  dst: T [taps][npad][kpad], zero padded.
// n_from_d0 = 1: n indexes D0, k indexes D1;  0: n indexes D1, k indexes D0.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1, int taps, int npad,
                                   int kpad, int n_from_d0) {
  const int64_t total = (int64_t)taps * npad * kpad;
  const int nn = n_from_d0 ? d0 : d1, kk = n_from_d0 ? d1 : d0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % kpad);
    const int n = (int)((i / kpad) % npad);
    const int t = (int)(i / ((int64_t)kpad * npad));
    float v = 0.f;
    if (n < nn && k < kk) {
      const int a = n_from_d0 ? n : k, b = n_from_d0 ? k : n;
      v = src[((size_t)a * d1 + b) * taps + t];
    }
    dst[i] = Elem<T>::cvt(v);
  }
}

// Tiled variant: a block stages a [TA][TB][taps] brick of the source through LDS -- its rows are contiguous runs of
// TB * taps floats -- and writes the taps planes with k fastest (64 k per (tap, n) row = 128 bytes in bf16).  The k
// dimension always gets the 64-wide side of the brick: (TA, TB) = (16, 64) when k indexes D1, (64, 16) when k indexes D0.
template <typename T, int TA, int TB>
__global__ __launch_bounds__(256) void pack_weight_tiled_kernel(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1,
                                                                int taps, int npad, int kpad) {
  constexpr bool N_FROM_D0 = (TB == 64);  // k = D1 index
  extern __shared__ float tile[];        // [TA][TB * taps + 1]
  const int pitch = TB * taps + 1;
  const int a0 = blockIdx.y * TA, b0 = blockIdx.x * TB;
  const int run = TB * taps;
  for (int i = threadIdx.x; i < TA * run; i += 256) {
    const int a = i / run, r = i - a * run;  // r = b * taps + t
    const int b = r / taps;
    float v = 0.f;
    if (a0 + a < d0 && b0 + b < d1) v = src[((size_t)(a0 + a) * d1 + b0) * taps + r];
    tile[a * pitch + r] = v;
  }
  __syncthreads();
  constexpr int NK = 64, NN = 16;  // brick extent along k and n
  for (int i = threadIdx.x; i < taps * NN * NK; i += 256) {
    const int k = i % NK, n = (i / NK) % NN, t = i / (NK * NN);
    const int a = N_FROM_D0 ? n : k, b = N_FROM_D0 ? k : n;
    const int gn = (N_FROM_D0 ? a0 : b0) + n, gk = (N_FROM_D0 ? b0 : a0) + k;
    if (gn < npad && gk < kpad) dst[((size_t)t * npad + gn) * kpad + gk] = Elem<T>::cvt(tile[a * pitch + b * taps + t]);
  }
}

extern "C" int mia_pack_weight(const float* src, void* dst, int dtype, int d0, int d1, int taps, int npad, int kpad,
                               int n_from_d0, void* stream) {
  MIA_CHECK_ARG(src && dst && d0 > 0 && d1 > 0 && taps > 0, "mia_pack_weight: bad arguments");
  MIA_CHECK_ARG(npad >= (n_from_d0 ? d0 : d1) && kpad >= (n_from_d0 ? d1 : d0), "mia_pack_weight: padding too small");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_pack_weight: bad dtype"); return MIA_EARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (taps <= 16 && (int64_t)npad * kpad >= 64 * 64) {
    // bricks cover the PADDED destination so the zero padding is written too
    const int pa = n_from_d0 ? npad : kpad, pb = n_from_d0 ? kpad : npad;  // padded extents along D0 / D1
    if (n_from_d0) {
      const dim3 grid(ceil_div(pb, 64), ceil_div(pa, 16));
      const size_t shb = (size_t)16 * (64 * taps + 1) * 4;
      if (dtype == MIA_BF16) hipLaunchKernelGGL((pack_weight_tiled_kernel<bf16_t, 16, 64>), grid, dim3(256), shb, st, src, static_cast<bf16_t*>(dst), d0, d1, taps, npad, kpad);
      else hipLaunchKernelGGL((pack_weight_tiled_kernel<float, 16, 64>), grid, dim3(256), shb, st, src, static_cast<float*>(dst), d0, d1, taps, npad, kpad);
    } else {
      const dim3 grid(ceil_div(pb, 16), ceil_div(pa, 64));
      const size_t shb = (size_t)64 * (16 * taps + 1) * 4;
      if (dtype == MIA_BF16) hipLaunchKernelGGL((pack_weight_tiled_kernel<bf16_t, 64, 16>), grid, dim3(256), shb, st, src, static_cast<bf16_t*>(dst), d0, d1, taps, npad, kpad);
      else hipLaunchKernelGGL((pack_weight_tiled_kernel<float, 64, 16>), grid, dim3(256), shb, st, src, static_cast<float*>(dst), d0, d1, taps, npad, kpad);
    }
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  const int64_t total = (int64_t)taps * npad * kpad;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == MIA_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, src, static_cast<bf16_t*>(dst), d0, d1,
                       taps, npad, kpad, n_from_d0);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, st, src, static_cast<float*>(dst), d0, d1, taps,
                       npad, kpad, n_from_d0);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Batched form: ONE launch re-packs every weight of a model (the optimizer rewrites all of them each step, and a
// per-tensor launch costs more than the copy for most layers).  `descs` lives in device memory; a block finds its
// descriptor from the running brick counts, then does the same LDS-tiled brick copy as above with run-time orientation.
struct MiaPackDesc {
  const float* src; void* dst;
  int d0, d1, taps, npad, kpad, n_from_d0;
  int brick_begin, bricks_x;  // first brick of this tensor in the launch; bricks along D1
};

// One brick with a compile-time tap count (TAPS = 0: run-time) so the index arithmetic divides by constants.  Measured (round 4): neither
// this nor the one-latency descriptor lookup below moved the launch (0.21 ms per cfg3 step = 1.75 TB/s for 372 MB): what bounds it is
// the 128-byte destination segments scattered over the tap planes, i.e. the brick shape -- a 32 n x 128 k brick staged as bf16 is the
// next thing to try (0.3 % of the step).
template <typename T, int TAPS>
__device__ __forceinline__ void pack_brick(const MiaPackDesc& d, int brick, float* tile) {
  const int taps = TAPS ? TAPS : d.taps;
  const int TA = d.n_from_d0 ? 16 : 64, TB = d.n_from_d0 ? 64 : 16;
  const int pitch = TB * taps + 1, run = TB * taps;
  const int a0 = (brick / d.bricks_x) * TA, b0 = (brick % d.bricks_x) * TB;
  // source rows: TB * taps contiguous floats each; a thread walks one row position across the TA rows, EIGHT loads in flight at a
  // time: branch-free buffer loads (out-of-range rows / columns read zero through an out-of-range offset).  The first version
  // guarded each load with a branch and hipcc waited vmcnt(0) after every one of them -- 36-64 serialized memory round trips per
  // thread, ~30 us per brick, 1.75 TB/s for the launch.
  // (a source tensor is < 4 GiB: ops.PackPlan checks it when it builds the descriptors)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src), 0, d.d0 * d.d1 * taps * 4, 0x00020000);
  const unsigned rstep = (unsigned)(d.d1 * taps * 4);
  for (int r = threadIdx.x; r < run; r += 256) {
    const bool bok = b0 + r / taps < d.d1;
    const unsigned row0 = (unsigned)(((a0 * d.d1 + b0) * taps + r) * 4);
    for (int a = 0; a < TA; a += 8) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = bok && a0 + a + u < d.d0;
        v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (int)(row0 + (unsigned)(a + u) * rstep) : (int)0xFFFFFFF0u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) tile[(a + u) * pitch + r] = __builtin_bit_cast(float, v[u]);
    }
  }
  __syncthreads();
  T* dst = static_cast<T*>(d.dst);
  const int k = threadIdx.x & 63, n_lo = threadIdx.x >> 6;  // 64 consecutive k per (tap, n) row: 128-byte (bf16) segments
  for (int tn = n_lo; tn < taps * 16; tn += 4) {
    const int n = tn & 15, t = tn >> 4;
    const int a = d.n_from_d0 ? n : k, b = d.n_from_d0 ? k : n;
    const int gn = (d.n_from_d0 ? a0 : b0) + n, gk = (d.n_from_d0 ? b0 : a0) + k;
    if (gn < d.npad && gk < d.kpad) dst[((size_t)t * d.npad + gn) * d.kpad + gk] = Elem<T>::cvt(tile[a * pitch + b * taps + t]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const MiaPackDesc* __restrict__ descs, int count) {
  extern __shared__ float tile[];
  __shared__ int which;
  if (threadIdx.x < 64) {
    // descriptor of this block's brick: the brick_begin values ascend, so it is the last one <= blockIdx.x.  Wave 0 loads 64 of
    // them per step IN PARALLEL and counts with a ballot -- one memory latency instead of a bisection's ~6 dependent loads
    int below = 0;
    for (int base = 0; base < count; base += 64) {
      const int i = base + (int)threadIdx.x;
      const bool le = i < count && descs[i].brick_begin <= (int)blockIdx.x;
      below += __popcll(__ballot(le));
    }
    if (threadIdx.x == 0) which = below - 1;
  }
  __syncthreads();
  const MiaPackDesc d = descs[which];
  const int brick = blockIdx.x - d.brick_begin;
  if (d.taps == 9) pack_brick<T, 9>(d, brick, tile);
  else if (d.taps == 4) pack_brick<T, 4>(d, brick, tile);
  else if (d.taps == 1) pack_brick<T, 1>(d, brick, tile);
  else pack_brick<T, 0>(d, brick, tile);
}

extern "C" int mia_pack_desc_bytes(void) { return (int)sizeof(MiaPackDesc); }

extern "C" int mia_pack_weight_batch(const void* descs_dev, int count, int total_bricks, int max_taps, int dtype, void* stream) {
  MIA_CHECK_ARG(descs_dev && count > 0 && total_bricks > 0 && max_taps > 0 && max_taps <= 16, "mia_pack_weight_batch: bad arguments");
  if (dtype != MIA_BF16 && dtype != MIA_F32) { mia_set_error("mia_pack_weight_batch: bad dtype"); return MIA_EARG; }
  const size_t shb = (size_t)64 * (16 * max_taps + 1) * 4;  // the larger of the two brick orientations
  hipStream_t st = static_cast<hipStream_t>(stream);
  const MiaPackDesc* d = static_cast<const MiaPackDesc*>(descs_dev);
  if (dtype == MIA_BF16) hipLaunchKernelGGL(pack_weight_batch_kernel<bf16_t>, dim3(total_bricks), dim3(256), shb, st, d, count);
  else hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(total_bricks), dim3(256), shb, st, d, count);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- layout / dtype conversion.  src element (n, c, p) lives at n*sn + c*sc + p*sp (element strides), dst likewise.
template <typename TS, typename TD>
__global__ void relayout_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int n, int c, int64_t hw, int64_t ssn,
                                int64_t ssc, int64_t ssp, int64_t dsn, int64_t dsc, int64_t dsp, int dst_c_fast) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ni, ci, pi;
    if (dst_c_fast) { ci = i % c; pi = (i / c) % hw; ni = i / ((int64_t)c * hw); }
    else { pi = i % hw; ci = (i / hw) % c; ni = i / ((int64_t)c * hw); }
    const float v = Elem<TS>::ld(src + ni * ssn + ci * ssc + pi * ssp);
    Elem<TD>::st(dst + ni * dsn + ci * dsc + pi * dsp, v);
  }
}

extern "C" int mia_relayout(const void* src, int src_dtype, void* dst, int dst_dtype, int n, int c, int64_t hw,
                            int64_t ssn, int64_t ssc, int64_t ssp, int64_t dsn, int64_t dsc, int64_t dsp, void* stream) {
  MIA_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0, "mia_relayout: bad arguments");
  const int64_t total = (int64_t)n * c * hw;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cf = (dsc == 1);
#define RL(TS, TD) hipLaunchKernelGGL((relayout_kernel<TS, TD>), dim3(blocks), dim3(256), 0, st, static_cast<const TS*>(src), \
                                      static_cast<TD*>(dst), n, c, hw, ssn, ssc, ssp, dsn, dsc, dsp, cf)
  if (src_dtype == MIA_F32 && dst_dtype == MIA_F32) RL(float, float);
  else if (src_dtype == MIA_F32 && dst_dtype == MIA_BF16) RL(float, bf16_t);
  else if (src_dtype == MIA_BF16 && dst_dtype == MIA_F32) RL(bf16_t, float);
  else if (src_dtype == MIA_BF16 && dst_dtype == MIA_BF16) RL(bf16_t, bf16_t);
  else { mia_set_error("mia_relayout: bad dtype"); return MIA_EARG; }
#undef RL
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- per-channel column sum over P rows of an NHWC tensor: out[c] (+)= sum_p x[p][c]   (bias gradients)
#define COLSUM_BLOCKS 256
// vectorised: block = 64 channels (UPB 16-byte units) x pixel lanes; grid = (blocks, C/64)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int64_t p, int c, float* __restrict__ part) {
  constexpr int EPU = Elem<T>::EPU, UPB = 64 / EPU, LANES = 256 / UPB;
  __shared__ float sh[LANES][64 + 1];
  const int u = threadIdx.x % UPB, pl = threadIdx.x / UPB;
  const int ch0 = blockIdx.y * 64 + u * EPU;
  const int64_t per = (p + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < p ? r0 + per : p;
  float s1[EPU];
#pragma unroll
  for (int e = 0; e < EPU; ++e) s1[e] = 0.f;
  for (int64_t r = r0 + pl; r < r1; r += LANES) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + r * c + ch0);
#pragma unroll
    for (int e = 0; e < EPU; ++e) s1[e] += Elem<T>::ld(v + e);
  }
#pragma unroll
  for (int e = 0; e < EPU; ++e) sh[pl][u * EPU + e] = s1[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll 8
    for (int j = 0; j < LANES; ++j) t += sh[j][threadIdx.x];
    part[(size_t)blockIdx.x * c + blockIdx.y * 64 + threadIdx.x] = t;
  }
}

template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ x, int64_t p, int c, float* __restrict__ part) {
  extern __shared__ float sh[];
  const int cw = blockDim.x >= c ? c : blockDim.x;
  const int rows_par = blockDim.x / cw;
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const int64_t rows_per_blk = (p + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = (r0 + rows_per_blk < p) ? r0 + rows_per_blk : p;
  for (int cb = blockIdx.y * cw; cb < c; cb += gridDim.y * cw) {
    const int ch = cb + tc;
    float s = 0.f;
    if (ch < c && tr < rows_par)
      for (int64_t r = r0 + tr; r < r1; r += rows_par) s += Elem<T>::ld(x + r * c + ch);
    sh[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0 && ch < c) {
      float t = 0.f;
      for (int j = 0; j < rows_par; ++j) t += sh[j * cw + tc];
      part[(size_t)blockIdx.x * c + ch] = t;
    }
    __syncthreads();
  }
}

// block = 16 channels x 16 lanes sweeping the partial rows
__global__ void colsum_final_kernel(const float* __restrict__ part, int nblk, int c, float* __restrict__ out, int accumulate) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (ch < c)
    for (int b = tl; b < nblk; b += 16) s += part[(size_t)b * c + ch];
  sh[tl][cl] = s;
  __syncthreads();
  if (tl == 0 && ch < c) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += sh[j][cl];
    out[ch] = accumulate ? out[ch] + t : t;
  }
}

static int colsum_blocks(int64_t p) { return (int)(p / 64 < 1 ? 1 : (p / 64 > COLSUM_BLOCKS ? COLSUM_BLOCKS : p / 64)); }

extern "C" int mia_colsum_workspace(int64_t p, int c) { return colsum_blocks(p) * c; }

extern "C" int mia_colsum(const void* x, int dtype, int64_t p, int c, float* workspace, float* out, int accumulate,
                          void* stream) {
  MIA_CHECK_ARG(x && workspace && out && p > 0 && c > 0, "mia_colsum: bad arguments");
  MIA_CHECK_ARG(dtype == MIA_BF16 || dtype == MIA_F32, "mia_colsum: bad dtype");
  const int blocks = colsum_blocks(p);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c % 64 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    if (dtype == MIA_BF16) hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, dim3(blocks, c / 64), dim3(256), 0, st, static_cast<const bf16_t*>(x), p, c, workspace);
    else hipLaunchKernelGGL(colsum_vec_kernel<float>, dim3(blocks, c / 64), dim3(256), 0, st, static_cast<const float*>(x), p, c, workspace);
  } else if (dtype == MIA_BF16) {
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(blocks, ceil_div(c, 256)), dim3(256), 256 * sizeof(float), st,
                       static_cast<const bf16_t*>(x), p, c, workspace);
  } else {
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(blocks, ceil_div(c, 256)), dim3(256), 256 * sizeof(float), st,
                       static_cast<const float*>(x), p, c, workspace);
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, st, workspace, blocks, c, out, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- out = a + b (residual connection of ResidualBlock, reference blocks.py:164); 16-byte vectorised when aligned
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n) {
  constexpr int EPU = Elem<T>::EPU;
  const int64_t nu = n / EPU;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    alignas(16) T va[EPU]; alignas(16) T vb[EPU]; alignas(16) T vo[EPU];
    *reinterpret_cast<u32x4*>(va) = *reinterpret_cast<const u32x4*>(a + i * EPU);
    *reinterpret_cast<u32x4*>(vb) = *reinterpret_cast<const u32x4*>(b + i * EPU);
#pragma unroll
    for (int e = 0; e < EPU; ++e) vo[e] = Elem<T>::cvt(Elem<T>::ld(va + e) + Elem<T>::ld(vb + e));
    *reinterpret_cast<u32x4*>(out + i * EPU) = *reinterpret_cast<const u32x4*>(vo);
  }
  for (int64_t i = nu * EPU + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}

extern "C" int mia_add(const void* a, const void* b, void* out, int dtype, int64_t n, void* stream) {
  MIA_CHECK_ARG(a && b && out && n > 0, "mia_add: bad arguments");
  MIA_CHECK_ARG(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "mia_add: 16-byte alignment required");
  const int64_t w = n / (dtype == MIA_BF16 ? 8 : 4) + 1;
  const int blocks = (int)((w + 255) / 256 < 8192 ? (w + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MIA_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t*>(a), static_cast<const bf16_t*>(b), static_cast<bf16_t*>(out), n);
  else if (dtype == MIA_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(a), static_cast<const float*>(b), static_cast<float*>(out), n);
  else { mia_set_error("mia_add: bad dtype"); return MIA_EARG; }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}


// ---- small utilities that keep PyTorch kernels out of the training step (north_star: torch = containers / autograd glue only)
// Dropout2d channel masks (reference blocks.py:92-96: nn.Dropout2d(p, inplace) zeroes whole channels per sample and scales
// the survivors by 1/(1-p)): out[i] in {0, 1/keep}, one Philox4x32-10 counter per four elements, keyed by (seed, offset).
__device__ __forceinline__ void philox_round_u(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// dyn (device, u64[2]; nullptr = use the arguments): seed and BASE offset of a step replayed from a captured hipGraph; `offset` is
// then the launch's fixed distance from that base (mia_dropout_mask_dyn)
__global__ void dropout_mask_kernel(float* __restrict__ out, int64_t n, float keep, unsigned long long seed, unsigned long long offset,
                                    const unsigned long long* __restrict__ dyn) {
  if (dyn != nullptr) { seed = dyn[0]; offset += dyn[1]; }
  const int64_t quads = (n + 3) / 4;
  const float scale = 1.f / keep;
  for (int64_t qd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += (int64_t)gridDim.x * blockDim.x) {
    unsigned c0 = (unsigned)qd, c1 = (unsigned)(qd >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round_u(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const unsigned c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = qd * 4 + e;
      if (i < n) out[i] = ((c[e] >> 8) * (1.f / 16777216.f) < keep) ? scale : 0.f;  // U[0,1) with 24 bits, keep with probability `keep`
    }
  }
}
extern "C" int mia_dropout_mask(float* out, int64_t n, float keep, uint64_t seed, uint64_t offset, void* stream) {
  MIA_CHECK_ARG(out && n > 0 && keep > 0.f && keep <= 1.f, "mia_dropout_mask: bad arguments (keep=%f)", (double)keep);
  const int64_t quads = (n + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, keep,
                     (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long*)nullptr);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
extern "C" int mia_dropout_mask_dyn(float* out, int64_t n, float keep, const uint64_t* seed_base, uint64_t rel_offset, void* stream) {
  MIA_CHECK_ARG(out && n > 0 && keep > 0.f && keep <= 1.f && seed_base, "mia_dropout_mask_dyn: bad arguments (keep=%f)", (double)keep);
  const int64_t quads = (n + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 1024 ? (quads + 255) / 256 : 1024);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, keep, 0ull,
                     (unsigned long long)rel_offset, reinterpret_cast<const unsigned long long*>(seed_base));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// optimizer.zero_grad() on the flat gradient buffer (al_trainer.py:1375): 16-byte stores
__global__ void zero_kernel(u32x4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = u32x4{0u, 0u, 0u, 0u};
}
extern "C" int mia_zero(void* p, int64_t bytes, void* stream) {
  MIA_CHECK_ARG(p && bytes > 0 && bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "mia_zero: needs a 16-byte aligned buffer of a multiple of 16 bytes");
  const int64_t n16 = bytes / 16;
  const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u32x4*>(p), n16);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// dst[i] = src[i * stride] (fp32): picks the per-channel sums out of interleaved (sum, sum of squares) statistics -- the
// transposed conv's bias gradient delivered by the decoder block's input-gradient epilogue (ops.py)
__global__ void gather_f32_kernel(const float* __restrict__ src, int64_t stride, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(int64_t)i * stride];
}
extern "C" int mia_gather_f32(const float* src, int64_t stride, float* dst, int n, void* stream) {
  MIA_CHECK_ARG(src && dst && n > 0 && stride > 0, "mia_gather_f32: bad arguments");
  hipLaunchKernelGGL(gather_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), src, stride, dst, n);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- max |x| of an fp32 tensor as an fp32 bit pattern (the scale source of the split-f16 convs, common.h SplitF16).  Non-negative fp32
// values order like their bit patterns, so the maximum is an unsigned integer maximum and the cross-block fold is an atomicMax, whose
// result does not depend on the order of arrival (deterministic).  NaN patterns order above infinity: a NaN in the tensor stays
// visible in the result.  Blocks skip the atomic when the slot already holds at least their maximum.
__device__ __forceinline__ unsigned amax_block(unsigned m, unsigned* red /* >= 4 words of LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = m;
  __syncthreads();
  unsigned r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = red[i] > r ? red[i] : r;
  return r;
}
__device__ __forceinline__ unsigned amax_span(const float* __restrict__ x, int64_t n, int64_t first, int64_t stride) {
  // 16-byte loads over the aligned body, scalar head / tail (first / stride in threads)
  unsigned m = 0;
  const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4;
  const int64_t h = head < n ? head : n;
  const int64_t n4 = (n - h) / 4;
  const u32x4* v = reinterpret_cast<const u32x4*>(x + h);
  for (int64_t i = first; i < n4; i += stride) {
    const u32x4 q = v[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const unsigned b = q[e] & 0x7FFFFFFFu; m = b > m ? b : m; }
  }
  const unsigned* xu = reinterpret_cast<const unsigned*>(x);
  for (int64_t i = first; i < h; i += stride) { const unsigned b = xu[i] & 0x7FFFFFFFu; m = b > m ? b : m; }
  for (int64_t i = h + 4 * n4 + first; i < n; i += stride) { const unsigned b = xu[i] & 0x7FFFFFFFu; m = b > m ? b : m; }
  return m;
}
__global__ void amax_slots_zero_kernel(unsigned* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ slot) {
  __shared__ unsigned red[4];
  const unsigned m = amax_block(amax_span(x, n, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256), red);
  if (threadIdx.x == 0 && m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
}
extern "C" int mia_amax(const float* x, int64_t n, void* slot, int reset, void* stream) {
  MIA_CHECK_ARG(x && slot && n > 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0 && (reinterpret_cast<uintptr_t>(slot) & 3) == 0, "mia_amax: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (the slot is zeroed by a kernel, not hipMemsetAsync: a captured train step that held memset nodes on graph-pool memory replayed
  // wrongly once a second graph had been captured -- round 5, tools/probe/graph_diverge.py)
  if (reset) hipLaunchKernelGGL(amax_slots_zero_kernel, dim3(1), dim3(64), 0, st, static_cast<unsigned*>(slot), 1);
  const int64_t want = (n / 4 + 255) / 256 / 8;  // ~8 sixteen-byte loads per thread
  const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, st, x, n, static_cast<unsigned*>(slot));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// Batched form: the maxima of `count` tensors (device table of {pointer, element count}) into slots[0 .. count) in ONE launch --
// every weight of a model right after the optimizer step (ops.PackPlan).  The slots are zeroed first.
struct MiaAmaxDesc { const float* src; int64_t n; };
__global__ __launch_bounds__(256) void amax_batch_kernel(const MiaAmaxDesc* __restrict__ descs, unsigned* __restrict__ slots) {
  __shared__ unsigned red[4];
  const MiaAmaxDesc d = descs[blockIdx.y];
  if ((int64_t)blockIdx.x * 1024 >= d.n && blockIdx.x > 0) return;  // small tensors leave most of their row of blocks empty
  const unsigned m = amax_block(amax_span(d.src, d.n, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256), red);
  if (threadIdx.x == 0 && m > __hip_atomic_load(slots + blockIdx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slots + blockIdx.y, m);
}
extern "C" int mia_amax_desc_bytes(void) { return (int)sizeof(MiaAmaxDesc); }
extern "C" int mia_amax_batch(const void* descs_dev, int count, void* slots, void* stream) {
  MIA_CHECK_ARG(descs_dev && slots && count > 0 && count <= 65535, "mia_amax_batch: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(amax_slots_zero_kernel, dim3((count + 255) / 256), dim3(256), 0, st, static_cast<unsigned*>(slots), count);
  hipLaunchKernelGGL(amax_batch_kernel, dim3(64, count), dim3(256), 0, st, static_cast<const MiaAmaxDesc*>(descs_dev), static_cast<unsigned*>(slots));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- labels shipped over PCIe as bytes and widened on the device (training/feed.py): dst[i] = src[i], uint8 -> int64.  A thread
// turns two labels (one 16-bit load) into one 16-byte store, so a wave reads 128 contiguous bytes and writes 1 KB contiguous.
__global__ __launch_bounds__(256) void widen_u8_i64_kernel(const uint8_t* __restrict__ src, long long* __restrict__ dst, int64_t n) {
  const int64_t n2 = n / 2;
  const unsigned short* s2 = reinterpret_cast<const unsigned short*>(src);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned w = s2[i];
    const u32x4 d = {w & 0xFFu, 0u, w >> 8, 0u};
    store_data_fence();
    *reinterpret_cast<u32x4*>(dst + 2 * i) = d;
    store_data_pad();  // tools/check_store_hazard.py
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = (long long)src[n - 1];
}
extern "C" int mia_widen_u8_i64(const void* src, void* dst, int64_t n, void* stream) {
  MIA_CHECK_ARG(src && dst && n > 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "mia_widen_u8_i64: bad arguments");
  const int64_t want = (n / 2 + 255) / 256 / 4;  // ~4 stores per thread
  const int blocks = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
  hipLaunchKernelGGL(widen_u8_i64_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint8_t*>(src),
                     static_cast<long long*>(dst), n);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---- host side of the label feed: int64 -> uint8 in ONE pass over host memory, with the range check folded in (returns 1 when every
// label lies in 0 .. 255 and dst is valid, 0 otherwise -- the caller then ships the int64 tensor as it is).  Plain threads: the torch
// CPU ops this replaces (aminmax + a converting copy_) cost 5 .. 50 ms per 77 MB batch on the GPU box's host, depending on what the
// OpenMP pool had just been doing (tools/probe/feed_cfg4.py).
#include <atomic>
#include <thread>
#include <vector>
extern "C" int mia_host_narrow_labels(const int64_t* src, uint8_t* dst, int64_t n, int threads) {
  if (!src || !dst || n <= 0) { mia_set_error("mia_host_narrow_labels: bad arguments"); return MIA_EARG; }
  int nt = threads > 0 ? threads : 8;
  const unsigned hc = std::thread::hardware_concurrency();
  if (hc > 0 && (unsigned)nt > hc) nt = (int)hc;
  if (n < (int64_t)1 << 16) nt = 1;
  std::atomic<int> ok{1};
  auto work = [&](int64_t b, int64_t e) {
    uint64_t bad = 0;
    for (int64_t i = b; i < e; ++i) { const uint64_t v = (uint64_t)src[i]; bad |= v; dst[i] = (uint8_t)v; }
    if (bad >> 8) ok.store(0, std::memory_order_relaxed);
  };
  if (nt == 1) { work(0, n); return ok.load(); }
  std::vector<std::thread> th;
  const int64_t per = ((n + nt - 1) / nt + 63) & ~(int64_t)63;
  for (int t = 0; t < nt; ++t) {
    const int64_t b = t * per, e = b + per < n ? b + per : n;
    if (b < e) th.emplace_back(work, b, e);
  }
  for (auto& t : th) t.join();
  return ok.load();
}

// ---- host side of the image feed: pageable -> pinned copy on a few plain threads.  (torch's CPU copy_ runs on its OpenMP pool, which
// a GPU box sizes by the MACHINE's core count -- 128 threads on a 16-core share -- and a 38 MB copy then took 0.1 .. 35 ms depending on
// what the pool had just been doing; tools/probe/feed_cfg4.py.)
extern "C" int mia_host_copy(void* dst, const void* src, int64_t bytes, int threads) {
  if (!src || !dst || bytes <= 0) { mia_set_error("mia_host_copy: bad arguments"); return MIA_EARG; }
  int nt = threads > 0 ? threads : 4;
  const unsigned hc = std::thread::hardware_concurrency();
  if (hc > 0 && (unsigned)nt > hc) nt = (int)hc;
  if (bytes < ((int64_t)1 << 20)) nt = 1;
  if (nt == 1) { memcpy(dst, src, (size_t)bytes); return MIA_OK; }
  std::vector<std::thread> th;
  const int64_t per = ((bytes + nt - 1) / nt + 4095) & ~(int64_t)4095;
  for (int t = 0; t < nt; ++t) {
    const int64_t b = t * per, e = b + per < bytes ? b + per : bytes;
    if (b < e) th.emplace_back([=]() { memcpy(static_cast<char*>(dst) + b, static_cast<const char*>(src) + b, (size_t)(e - b)); });
  }
  for (auto& t : th) t.join();
  return MIA_OK;
}

// ---- packed fp32 weights -> (h | l << 16) fp16 words of w * 2^e (common.h SplitF16), e from the tensor's maximum slot: the split-f16
// convs then stage their weights as they are instead of splitting them once per tile (round 4 measured that at 7-10 % of a conv launch).
struct MiaSplitDesc { const float* src; unsigned* dst; int64_t n; const unsigned* amax; };
__device__ __forceinline__ void split_span(const MiaSplitDesc& d, int64_t first, int64_t stride) {
  const float s = SplitF16::pow2(SplitF16::exp_of(*d.amax & 0x7FFFFFFFu));
  const int64_t n4 = d.n / 4;  // (packed buffers are multiples of 64 x 16 elements, 16-byte aligned)
  const u32x4* src = reinterpret_cast<const u32x4*>(d.src);
  u32x4* dst = reinterpret_cast<u32x4*>(d.dst);
  for (int64_t i = first; i < n4; i += stride) {
    const u32x4 w = SplitF16::unit(src[i], s);
    store_data_fence();
    dst[i] = w;
    store_data_pad();
  }
}
__global__ __launch_bounds__(256) void split_f16_batch_kernel(const MiaSplitDesc* __restrict__ descs) {
  const MiaSplitDesc d = descs[blockIdx.y];
  if ((int64_t)blockIdx.x * 256 >= d.n / 4 && blockIdx.x > 0) return;
  split_span(d, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}
extern "C" int mia_split_desc_bytes(void) { return (int)sizeof(MiaSplitDesc); }
extern "C" int mia_split_f16_batch(const void* descs_dev, int count, void* stream) {
  MIA_CHECK_ARG(descs_dev && count > 0 && count <= 65535, "mia_split_f16_batch: bad arguments");
  hipLaunchKernelGGL(split_f16_batch_kernel, dim3(64, count), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const MiaSplitDesc*>(descs_dev));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
