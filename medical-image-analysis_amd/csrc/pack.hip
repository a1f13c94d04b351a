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

extern "C" int mia_pack_weight(const float* src, void* dst, int dtype, int d0, int d1, int taps, int npad, int kpad,
                               int n_from_d0, void* stream) {
  MIA_CHECK_ARG(src && dst && d0 > 0 && d1 > 0 && taps > 0, "mia_pack_weight: bad arguments");
  MIA_CHECK_ARG(npad >= (n_from_d0 ? d0 : d1) && kpad >= (n_from_d0 ? d1 : d0), "mia_pack_weight: padding too small");
  const int64_t total = (int64_t)taps * npad * kpad;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MIA_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, src, static_cast<bf16_t*>(dst), d0, d1,
                       taps, npad, kpad, n_from_d0);
  else if (dtype == MIA_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, st, src, static_cast<float*>(dst), d0, d1, taps,
                       npad, kpad, n_from_d0);
  else { mia_set_error("mia_pack_weight: bad dtype"); return MIA_EARG; }
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
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ x, int64_t p, int c, float* __restrict__ part) {
  // block = 256 threads: thread t -> channel (t % cw), row lane (t / cw); cw = min(c, 256) rounded to pow2 by host
  extern __shared__ float sh[];
  const int cw = blockDim.x >= c ? c : blockDim.x;  // channels covered per pass
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

__global__ void colsum_final_kernel(const float* __restrict__ part, int nblk, int c, float* __restrict__ out, int accumulate) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(size_t)b * c + ch];
  out[ch] = accumulate ? out[ch] + s : s;
}

extern "C" int mia_colsum_workspace(int64_t p, int c) {
  const int blocks = (int)(p / 64 < 1 ? 1 : (p / 64 > 1024 ? 1024 : p / 64));
  return blocks * c;  // floats
}

extern "C" int mia_colsum(const void* x, int dtype, int64_t p, int c, float* workspace, float* out, int accumulate,
                          void* stream) {
  MIA_CHECK_ARG(x && workspace && out && p > 0 && c > 0, "mia_colsum: bad arguments");
  const int blocks = (int)(p / 64 < 1 ? 1 : (p / 64 > 1024 ? 1024 : p / 64));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MIA_BF16)
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(blocks, ceil_div(c, 256)), dim3(256), 256 * sizeof(float), st,
                       static_cast<const bf16_t*>(x), p, c, workspace);
  else if (dtype == MIA_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(blocks, ceil_div(c, 256)), dim3(256), 256 * sizeof(float), st,
                       static_cast<const float*>(x), p, c, workspace);
  else { mia_set_error("mia_colsum: bad dtype"); return MIA_EARG; }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, workspace, blocks, c, out, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
