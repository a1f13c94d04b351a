// 1x1 segmentation head (C0 -> K1 <= 8 logits) and the fused Dice + cross-entropy loss, gfx950.
//
// Reference: seg_output = Conv2d(c0, K1, 1) (src/models/unet/unet.py:176); DiceLoss.forward
// (src/losses/dice_loss.py:32-76); DiceAndCELoss.forward (src/losses/compound_losses.py:33-49) with
// torch.nn.CrossEntropyLoss (mean over all pixels).  These are pure bandwidth: the head reads C0
// values and writes K1 per pixel; the loss reads K1 logits + one label per pixel ONCE (the reference
// materialises softmax, a long one-hot, a float one-hot and a product) and reduces I = sum p*t,
// sum p (or p^2), sum t per (image, class) plus the CE sum with wave shuffles, no float atomics.
#include "common.h"

#define MAXK 8

// ---------------------------------------------------------------- head forward
// 8 lanes per pixel, each lane strides over the pixel's 16-byte units; K1 accumulators; xor-shuffle reduce.
template <typename T>
__global__ void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ out, int64_t npix, int c0, int k1, int64_t osp, int64_t osk,
                                int64_t osn, int64_t hw) {
  extern __shared__ float wsh[];  // [k1][c0]
  for (int i = threadIdx.x; i < k1 * c0; i += blockDim.x) wsh[i] = w[i];
  __syncthreads();
  constexpr int EPU = Elem<T>::EPU;
  const int sub = threadIdx.x & 7;
  const bool vec = (c0 % EPU) == 0;
  for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; p < npix; p += ((int64_t)gridDim.x * blockDim.x) >> 3) {
    float acc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[k] = 0.f;
    const T* row = x + p * c0;
    if (vec) {
      for (int u = sub; u < c0 / EPU; u += 8) {
        alignas(16) T v[EPU];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(row + u * EPU);
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
          const float xv = Elem<T>::ld(v + e);
#pragma unroll
          for (int k = 0; k < MAXK; ++k)
            if (k < k1) acc[k] += xv * wsh[k * c0 + u * EPU + e];
        }
      }
    } else {
      for (int ch = sub; ch < c0; ch += 8) {
        const float xv = Elem<T>::ld(row + ch);
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
          if (k < k1) acc[k] += xv * wsh[k * c0 + ch];
      }
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      if (k < k1) {
        float v = acc[k];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (sub == 0) out[(osn == hw * osp ? p * osp : (p / hw) * osn + (p % hw) * osp) + k * osk] = v + b[k];
      }
    }
  }
}

// ---------------------------------------------------------------- specialised head kernels
// K1 (logits) and UPP (16-byte units per pixel = threads per pixel, 4 / 8 / 16) are compile-time: a thread owns one
// channel unit for its whole life, so its K1 x EPU weights sit in registers, there is no per-element index arithmetic
// (32-bit pixel counters, pointers advanced by a constant stride) and the cross-lane sums are DPP butterflies instead
// of LDS permutes.  Contract: c0 == UPP * EPU, pixel-linear logits (sn == hw * sp), n * hw * c0 < 2^31.
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
  const int iv = __builtin_bit_cast(int, v);
  int r;
  if (ctrl_sel == 0) r = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]: lane ^ 1
  else if (ctrl_sel == 1) r = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]: lane ^ 2
  else if (ctrl_sel == 2) r = __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false);  // row_half_mirror: other quad of 8
  else r = __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false);                     // row_mirror: other half of 16
  return v + __builtin_bit_cast(float, r);
}
template <int UPP> __device__ __forceinline__ float group_sum(float v) {  // every lane of the UPP group gets the sum
  v = dpp_add(v, 0); v = dpp_add(v, 1);
  if (UPP >= 8) v = dpp_add(v, 2);
  if (UPP >= 16) v = dpp_add(v, 3);
  return v;
}

template <typename T, int K1, int UPP>
__global__ __launch_bounds__(256) void head_fwd_fast_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ out, int npix,
                                                            int64_t osp, int64_t osk) {
  constexpr int EPU = Elem<T>::EPU, C0 = UPP * EPU, LANES = 256 / UPP;
  const int u = threadIdx.x % UPP, pl = threadIdx.x / UPP;
  float wr[K1][EPU];
#pragma unroll
  for (int k = 0; k < K1; ++k)
#pragma unroll
    for (int e = 0; e < EPU; ++e) wr[k][e] = w[k * C0 + u * EPU + e];
  float bu = 0.f;
#pragma unroll
  for (int k = 0; k < K1; ++k) bu = (u == k) ? b[k] : bu;
  const int stride = gridDim.x * LANES;
  auto body = [&](const u32x4& raw, int p) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = raw;
    float acc[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) acc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float xv = Elem<T>::ld(v + e);
#pragma unroll
      for (int k = 0; k < K1; ++k) acc[k] += xv * wr[k][e];
    }
    float mine = 0.f;  // lane u < K1 of the group writes logit u: one store instruction, K1 adjacent floats per pixel
#pragma unroll
    for (int k = 0; k < K1; ++k) { const float t = group_sum<UPP>(acc[k]); mine = (u == k) ? t : mine; }
    if (u < K1) out[(int64_t)p * osp + u * osk] = mine + bu;
  };
  int p = blockIdx.x * LANES + pl;
  for (; p + stride < npix; p += 2 * stride) {  // two independent 16-byte loads in flight
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(x + (size_t)p * C0 + u * EPU);
    const u32x4 r1 = *reinterpret_cast<const u32x4*>(x + (size_t)(p + stride) * C0 + u * EPU);
    body(r0, p); body(r1, p + stride);
  }
  if (p < npix) body(*reinterpret_cast<const u32x4*>(x + (size_t)p * C0 + u * EPU), p);
}

template <typename T, int K1, int UPP>
__global__ __launch_bounds__(256) void head_bwd_input_fast_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                  T* __restrict__ dx, int npix, int64_t gsp, int64_t gsk) {
  constexpr int EPU = Elem<T>::EPU, C0 = UPP * EPU, LANES = 256 / UPP;
  const int u = threadIdx.x % UPP, pl = threadIdx.x / UPP;
  float wr[K1][EPU];
#pragma unroll
  for (int k = 0; k < K1; ++k)
#pragma unroll
    for (int e = 0; e < EPU; ++e) wr[k][e] = w[k * C0 + u * EPU + e];
  const int stride = gridDim.x * LANES;
  for (int p = blockIdx.x * LANES + pl; p < npix; p += stride) {
    const float* g = dl + (int64_t)p * gsp;
    float gv[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) gv[k] = g[k * gsk];
    alignas(16) T o[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) a += gv[k] * wr[k][e];
      o[e] = Elem<T>::cvt(a);
    }
    *reinterpret_cast<u32x4*>(dx + (size_t)p * C0 + u * EPU) = *reinterpret_cast<const u32x4*>(o);
  }
}

// part: [gridDim.x][K1][C0 + 1] (last column = bias partial), same layout as the generic kernels
template <typename T, int K1, int UPP>
__global__ __launch_bounds__(256) void head_bwd_weight_fast_kernel(const float* __restrict__ dl, const T* __restrict__ x,
                                                                   float* __restrict__ part, int npix, int64_t gsp, int64_t gsk) {
  constexpr int EPU = Elem<T>::EPU, C0 = UPP * EPU, LANES = 256 / UPP, SHS = C0 + 1;
  __shared__ float shd[LANES * SHS];
  const int u = threadIdx.x % UPP, pl = threadIdx.x / UPP;
  const int per = (npix + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < npix ? r0 + per : npix;
  float acc[K1][EPU], bacc[K1];
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    bacc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[k][e] = 0.f;
  }
  auto body = [&](const u32x4& raw, const float* gv) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = raw;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      const float xv = Elem<T>::ld(v + e);
#pragma unroll
      for (int k = 0; k < K1; ++k) acc[k][e] += gv[k] * xv;
    }
#pragma unroll
    for (int k = 0; k < K1; ++k) bacc[k] += gv[k];
  };
  int p = r0 + pl;
  for (; p + LANES < r1; p += 2 * LANES) {
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(x + (size_t)p * C0 + u * EPU);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(x + (size_t)(p + LANES) * C0 + u * EPU);
    float g0[K1], g1[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) { g0[k] = dl[(int64_t)p * gsp + k * gsk]; g1[k] = dl[(int64_t)(p + LANES) * gsp + k * gsk]; }
    body(a0, g0); body(a1, g1);
  }
  if (p < r1) {
    float g0[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) g0[k] = dl[(int64_t)p * gsp + k * gsk];
    body(*reinterpret_cast<const u32x4*>(x + (size_t)p * C0 + u * EPU), g0);
  }
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPU; ++e) shd[pl * SHS + u * EPU + e] = acc[k][e];
    if (u == 0) shd[pl * SHS + C0] = bacc[k];
    __syncthreads();
    if (threadIdx.x <= C0) {
      float t = 0.f;
      for (int j = 0; j < LANES; ++j) t += shd[j * SHS + threadIdx.x];
      part[((size_t)blockIdx.x * K1 + k) * (C0 + 1) + threadIdx.x] = t;
    }
  }
}

// 0 = not eligible; otherwise UPP
static int head_fast_upp(int dtype, int c0, int k1, int64_t npix, int64_t sn, int64_t sp, int64_t hw, const void* x) {
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  if (dtype != MIA_BF16 && dtype != MIA_F32) return 0;
  if (c0 % epu != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return 0;
  const int upp = c0 / epu;
  if (upp != 4 && upp != 8 && upp != 16) return 0;
  if (k1 < 2 || k1 > 4) return 0;
  if (sn != hw * sp || npix * c0 >= ((int64_t)1 << 31)) return 0;
  return upp;
}
#define HEAD_DISPATCH(KERNEL, T, grid, ...)                                                                           \
  do {                                                                                                                \
    if (upp == 4) {                                                                                                   \
      if (k1 == 2) hipLaunchKernelGGL((KERNEL<T, 2, 4>), grid, dim3(256), 0, st, __VA_ARGS__);                        \
      else if (k1 == 3) hipLaunchKernelGGL((KERNEL<T, 3, 4>), grid, dim3(256), 0, st, __VA_ARGS__);                   \
      else hipLaunchKernelGGL((KERNEL<T, 4, 4>), grid, dim3(256), 0, st, __VA_ARGS__);                                \
    } else if (upp == 8) {                                                                                            \
      if (k1 == 2) hipLaunchKernelGGL((KERNEL<T, 2, 8>), grid, dim3(256), 0, st, __VA_ARGS__);                        \
      else if (k1 == 3) hipLaunchKernelGGL((KERNEL<T, 3, 8>), grid, dim3(256), 0, st, __VA_ARGS__);                   \
      else hipLaunchKernelGGL((KERNEL<T, 4, 8>), grid, dim3(256), 0, st, __VA_ARGS__);                                \
    } else {                                                                                                          \
      if (k1 == 2) hipLaunchKernelGGL((KERNEL<T, 2, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                       \
      else if (k1 == 3) hipLaunchKernelGGL((KERNEL<T, 3, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                  \
      else hipLaunchKernelGGL((KERNEL<T, 4, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                               \
    }                                                                                                                 \
  } while (0)

// the fused norm + head kernels also take twelve units per pixel (96 bf16 / 48 fp32 channels) on groups of sixteen lanes
#define HEAD_DISPATCH_N(KERNEL, T, grid, ...)                                                                         \
  do {                                                                                                                \
    if (upp == 12) {                                                                                                  \
      if (k1 == 2) hipLaunchKernelGGL((KERNEL<T, 2, 16, 12>), grid, dim3(256), 0, st, __VA_ARGS__);                   \
      else if (k1 == 3) hipLaunchKernelGGL((KERNEL<T, 3, 16, 12>), grid, dim3(256), 0, st, __VA_ARGS__);              \
      else hipLaunchKernelGGL((KERNEL<T, 4, 16, 12>), grid, dim3(256), 0, st, __VA_ARGS__);                           \
    } else HEAD_DISPATCH(KERNEL, T, grid, __VA_ARGS__);                                                               \
  } while (0)


// ---------------------------------------------------------------- head fused with the last block's norm + LeakyReLU
// The last decoder block's activated output z = lrelu(scale*y + shift) has exactly one consumer, the 1x1 head.  These
// kernels recompute it from the raw conv output y on load (per-(image, channel) scale / shift in registers: a block
// works inside one image), so z is never written or read: the forward apply pass and one activation round trip vanish.
// CU < UPP (round 5: 96 channels in bf16 = 12 units dealt to groups of 16 lanes): lanes CU .. UPP - 1 of a pixel group load nothing and
// hold zero coefficients, so the group sums and every address (C0 = CU units) stay right; 25 % idle lanes on a memory-bound pass.
template <typename T, int K1, int UPP, int CU = UPP>
__global__ __launch_bounds__(256) void head_norm_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float slope,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ out, int hw, int slabs, int64_t osn,
                                                            int64_t osp, int64_t osk) {
  constexpr int EPU = Elem<T>::EPU, C0 = CU * EPU, LANES = 256 / UPP;
  const int u = threadIdx.x % UPP, pl = threadIdx.x / UPP;
  const bool act = CU == UPP || u < CU;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float wr[K1][EPU], sc[EPU], sf[EPU];
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    sc[e] = act ? scale[(size_t)n * C0 + u * EPU + e] : 0.f;
    sf[e] = act ? shift[(size_t)n * C0 + u * EPU + e] : 0.f;
#pragma unroll
    for (int k = 0; k < K1; ++k) wr[k][e] = act ? w[k * C0 + u * EPU + e] : 0.f;
  }
  float bu = 0.f;
#pragma unroll
  for (int k = 0; k < K1; ++k) bu = (u == k) ? b[k] : bu;
  const T* yb = y + (size_t)n * hw * C0 + u * EPU;
  float* ob = out + (int64_t)n * osn + u * osk;
  auto body = [&](const u32x4& raw, int p) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = raw;
    float acc[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) acc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      float xv = sc[e] * Elem<T>::ld(v + e) + sf[e];
      xv = xv > 0.f ? xv : xv * slope;
#pragma unroll
      for (int k = 0; k < K1; ++k) acc[k] += xv * wr[k][e];
    }
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < K1; ++k) { const float t = group_sum<UPP>(acc[k]); mine = (u == k) ? t : mine; }
    if (u < K1) ob[(int64_t)p * osp] = mine + bu;
  };
  // Bulk: a wave takes 64 consecutive pixels per round -- UPP independent 16-byte loads per lane in flight (each load
  // instruction reads 64 / UPP whole pixels = 1 KB contiguous), the K1 sums of a pixel go through a wave-private LDS
  // transpose so that the logits leave as one contiguous 256-byte run per class plane (NCHW fp32 logits: the per-pixel
  // store of K1 lanes per pixel group wrote 32-byte pieces, 1.8 TB/s on the cfg3 launch).
  constexpr int GP = 64 / UPP;  // pixels per load instruction and wave
  __shared__ float tbuf[4][K1][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gl = lane / UPP;
  float* tb = &tbuf[wave][0][0];
  const int nbulk = (r1 - r0) / 256 * 256;
  for (int q0 = r0 + wave * 64; q0 < r0 + nbulk; q0 += 256) {
    u32x4 raw[UPP];
#pragma unroll
    for (int j = 0; j < UPP; ++j) raw[j] = act ? *reinterpret_cast<const u32x4*>(yb + (size_t)(q0 + j * GP + gl) * C0) : zero4;
#pragma unroll
    for (int j = 0; j < UPP; ++j) {
      alignas(16) T v[EPU];
      *reinterpret_cast<u32x4*>(v) = raw[j];
      float acc[K1];
#pragma unroll
      for (int k = 0; k < K1; ++k) acc[k] = 0.f;
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        float xv = sc[e] * Elem<T>::ld(v + e) + sf[e];
        xv = xv > 0.f ? xv : xv * slope;
#pragma unroll
        for (int k = 0; k < K1; ++k) acc[k] += xv * wr[k][e];
      }
      float mine = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) { const float t = group_sum<UPP>(acc[k]); mine = (u == k) ? t : mine; }
      if (u < K1) tb[u * 64 + j * GP + gl] = mine + bu;
    }
    // wave-private transpose: other lanes read what this lane wrote.  The wave barrier (free at run time) states the
    // write -> read and, for the next round, read -> write order in the program instead of leaving it to in-order DS issue.
    __builtin_amdgcn_wave_barrier();
    float* on = out + (int64_t)n * osn + (int64_t)(q0 + lane) * osp;
#pragma unroll
    for (int k = 0; k < K1; ++k) on[k * osk] = tb[k * 64 + lane];
    __builtin_amdgcn_wave_barrier();
  }
  for (int p = r0 + nbulk + pl; p < r1; p += LANES) body(act ? *reinterpret_cast<const u32x4*>(yb + (size_t)p * C0) : zero4, p);
}

// part: [gridDim.x][K1][C0 + 1] like head_bwd_weight_fast_kernel, with x = lrelu(scale*y + shift) recomputed
template <typename T, int K1, int UPP, int CU = UPP>  // (CU < UPP: see head_norm_fwd_kernel)
__global__ __launch_bounds__(256) void head_norm_wgrad_kernel(const float* __restrict__ dl, const T* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope, float* __restrict__ part, int hw, int slabs,
                                                              int64_t gsn, int64_t gsp, int64_t gsk) {
  constexpr int EPU = Elem<T>::EPU, C0 = CU * EPU, LANES = 256 / UPP, SHS = C0 + 1;
  __shared__ float shd[LANES * SHS];
  const int u = threadIdx.x % UPP, pl = threadIdx.x / UPP;
  const bool act = CU == UPP || u < CU;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float sc[EPU], sf[EPU], acc[K1][EPU], bacc[K1];
#pragma unroll
  for (int e = 0; e < EPU; ++e) {
    sc[e] = act ? scale[(size_t)n * C0 + u * EPU + e] : 0.f;
    sf[e] = act ? shift[(size_t)n * C0 + u * EPU + e] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    bacc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[k][e] = 0.f;
  }
  const T* yb = y + (size_t)n * hw * C0 + u * EPU;
  const float* gb = dl + (int64_t)n * gsn;
  auto body = [&](const u32x4& raw, const float (&gv)[K1]) {
    alignas(16) T v[EPU];
    *reinterpret_cast<u32x4*>(v) = raw;
#pragma unroll
    for (int k = 0; k < K1; ++k) bacc[k] += gv[k];
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      float xv = sc[e] * Elem<T>::ld(v + e) + sf[e];
      xv = xv > 0.f ? xv : xv * slope;
#pragma unroll
      for (int k = 0; k < K1; ++k) acc[k][e] += gv[k] * xv;
    }
  };
  int p = r0 + pl;
  for (; p + 3 * LANES < r1; p += 4 * LANES) {  // four pixels' loads in flight per thread
    u32x4 raw[4];
    float gv[4][K1];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = act ? *reinterpret_cast<const u32x4*>(yb + (size_t)(p + j * LANES) * C0) : zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < K1; ++k) gv[j][k] = gb[(int64_t)(p + j * LANES) * gsp + k * gsk];
#pragma unroll
    for (int j = 0; j < 4; ++j) body(raw[j], gv[j]);
  }
  for (; p < r1; p += LANES) {
    float gv[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) gv[k] = gb[(int64_t)p * gsp + k * gsk];
    body(act ? *reinterpret_cast<const u32x4*>(yb + (size_t)p * C0) : zero4, gv);
  }
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPU; ++e)
      if (act) shd[pl * SHS + u * EPU + e] = acc[k][e];
    if (u == 0) shd[pl * SHS + C0] = bacc[k];
    __syncthreads();
    if (threadIdx.x <= C0) {
      float t = 0.f;
      for (int j = 0; j < LANES; ++j) t += shd[j * SHS + threadIdx.x];
      part[((size_t)blockIdx.x * K1 + k) * (C0 + 1) + threadIdx.x] = t;
    }
  }
}

extern "C" int mia_head_fwd(const void* x, int dtype, const float* w, const float* b, float* logits, int n, int64_t hw,
                            int c0, int k1, int64_t osn, int64_t osk, int64_t osp, void* stream) {
  MIA_CHECK_ARG(x && w && b && logits && n > 0 && hw > 0 && c0 > 0, "mia_head_fwd: bad arguments");
  MIA_CHECK_ARG(k1 >= 1 && k1 <= MAXK, "mia_head_fwd: k1=%d not in [1,%d]", k1, MAXK);
  MIA_CHECK_ARG((size_t)k1 * c0 * 4 <= 60000, "mia_head_fwd: weight tile too large for LDS");
  const int64_t npix = (int64_t)n * hw;
  const int blocks = (int)((npix * 8 + 255) / 256 < 8192 ? (npix * 8 + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (const int upp = head_fast_upp(dtype, c0, k1, npix, osn, osp, hw, x)) {
    const int lanes = 256 / upp;
    const int fblocks = (int)((npix + 2 * lanes - 1) / (2 * lanes) < 16384 ? (npix + 2 * lanes - 1) / (2 * lanes) : 16384);
    if (dtype == MIA_BF16) HEAD_DISPATCH(head_fwd_fast_kernel, bf16_t, dim3(fblocks), static_cast<const bf16_t*>(x), w, b, logits, (int)npix, osp, osk);
    else HEAD_DISPATCH(head_fwd_fast_kernel, float, dim3(fblocks), static_cast<const float*>(x), w, b, logits, (int)npix, osp, osk);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  if (dtype == MIA_BF16)
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), k1 * c0 * 4, st, static_cast<const bf16_t*>(x), w, b, logits, npix, c0, k1, osp, osk, osn, hw);
  else if (dtype == MIA_F32)
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(blocks), dim3(256), k1 * c0 * 4, st, static_cast<const float*>(x), w, b, logits, npix, c0, k1, osp, osk, osn, hw);
  else { mia_set_error("mia_head_fwd: bad dtype"); return MIA_EARG; }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- head backward (input gradient)
template <typename T>
__global__ void head_bwd_input_kernel(const float* __restrict__ dl, const float* __restrict__ w, T* __restrict__ dx,
                                      int64_t npix, int c0, int k1, int64_t gsp, int64_t gsk, int64_t gsn, int64_t hw) {
  extern __shared__ float wsh[];
  for (int i = threadIdx.x; i < k1 * c0; i += blockDim.x) wsh[i] = w[i];
  __syncthreads();
  const int64_t total = npix * c0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c0;
    const int ch = (int)(i - p * c0);
    const float* g = dl + (gsn == hw * gsp ? p * gsp : (p / hw) * gsn + (p % hw) * gsp);
    float s = 0.f;
    for (int k = 0; k < k1; ++k) s += g[k * gsk] * wsh[k * c0 + ch];
    Elem<T>::st(dx + i, s);
  }
}

// weight / bias gradient partials: block handles a pixel slab; thread -> channel, loops pixels
template <typename T>
__global__ void head_bwd_weight_kernel(const float* __restrict__ dl, const T* __restrict__ x, float* __restrict__ part,
                                       int64_t npix, int c0, int k1, int64_t gsp, int64_t gsk, int64_t gsn, int64_t hw) {
  // part: [gridDim.x][k1][c0 + 1]  (last column = bias partial)
  extern __shared__ float sh[];  // [rows_par][cw] per k
  const int cw = blockDim.x >= c0 ? c0 : blockDim.x, rows_par = blockDim.x / cw;
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < npix ? r0 + per : npix;
  for (int cb = 0; cb < c0; cb += cw) {
    const int ch = cb + tc;
    float acc[MAXK], bacc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { acc[k] = 0.f; bacc[k] = 0.f; }
    if (ch < c0 && tr < rows_par)
      for (int64_t p = r0 + tr; p < r1; p += rows_par) {
        const float xv = Elem<T>::ld(x + p * c0 + ch);
        const float* g = dl + (gsn == hw * gsp ? p * gsp : (p / hw) * gsn + (p % hw) * gsp);
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
          if (k < k1) { const float gv = g[k * gsk]; acc[k] += gv * xv; bacc[k] += gv; }
      }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      if (k >= k1) break;
      __syncthreads();
      sh[threadIdx.x] = acc[k];
      sh[blockDim.x + threadIdx.x] = bacc[k];
      __syncthreads();
      if (tr == 0 && ch < c0) {
        float t = 0.f;
        for (int j = 0; j < rows_par; ++j) t += sh[j * cw + tc];
        part[((size_t)blockIdx.x * k1 + k) * (c0 + 1) + ch] = t;
        if (ch == 0) {
          float tb = 0.f;
          for (int j = 0; j < rows_par; ++j) tb += sh[blockDim.x + j * cw];
          part[((size_t)blockIdx.x * k1 + k) * (c0 + 1) + c0] = tb;
        }
      }
    }
    __syncthreads();
  }
}

// block = 16 outputs x 16 lanes sweeping the partial blocks
__global__ void head_bwd_final_kernel(const float* __restrict__ part, int nblk, int k1, int c0, float* __restrict__ dw,
                                      float* __restrict__ db, int accumulate) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  const int tot = k1 * (c0 + 1);
  float s = 0.f;
  if (i < tot)
    for (int b = tl; b < nblk; b += 16) s += part[(size_t)b * tot + i];
  sh[tl][cl] = s;
  __syncthreads();
  if (tl == 0 && i < tot) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += sh[j][cl];
    const int k = i / (c0 + 1), ch = i % (c0 + 1);
    if (ch < c0) dw[k * c0 + ch] = accumulate ? dw[k * c0 + ch] + t : t;
    else db[k] = accumulate ? db[k] + t : t;
  }
}

// vectorised variants (c0 % EPU == 0, 16-byte aligned): thread = (pixel lane, 16-byte unit)
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_input_vec_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                 T* __restrict__ dx, int64_t npix, int c0, int k1, int64_t gsp,
                                                                 int64_t gsk, int64_t gsn, int64_t hw) {
  extern __shared__ float wsh[];
  for (int i = threadIdx.x; i < k1 * c0; i += blockDim.x) wsh[i] = w[i];
  __syncthreads();
  constexpr int EPU = Elem<T>::EPU;
  const int upp = c0 / EPU;
  const int64_t total = npix * upp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / upp;
    const int ch = (int)(i - p * upp) * EPU;
    const float* g = dl + (gsn == hw * gsp ? p * gsp : (p / hw) * gsn + (p % hw) * gsp);
    float acc[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float gv = g[k * gsk];
#pragma unroll
        for (int e = 0; e < EPU; ++e) acc[e] += gv * wsh[k * c0 + ch + e];
      }
    alignas(16) T out[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) out[e] = Elem<T>::cvt(acc[e]);
    *reinterpret_cast<u32x4*>(dx + i * EPU) = *reinterpret_cast<const u32x4*>(out);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_weight_vec_kernel(const float* __restrict__ dl, const T* __restrict__ x,
                                                                  float* __restrict__ part, int64_t npix, int c0, int k1,
                                                                  int64_t gsp, int64_t gsk, int64_t gsn, int64_t hw) {
  // part: [gridDim.x][k1][c0 + 1]; requires c0 <= 256 and c0 % EPU == 0
  constexpr int EPU = Elem<T>::EPU;
  extern __shared__ float shd[];  // [lanes][c0 + 1]
  const int shs = c0 + 1;
  const int upp = c0 / EPU, lanes = 256 / upp;
  const int u = threadIdx.x % upp, pl = threadIdx.x / upp;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < npix ? r0 + per : npix;
  float acc[MAXK][EPU], bacc[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    bacc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[k][e] = 0.f;
  }
  if (pl < lanes)
    for (int64_t p = r0 + pl; p < r1; p += lanes) {
      alignas(16) T v[EPU];
      *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + p * c0 + u * EPU);
      const float* g = dl + (gsn == hw * gsp ? p * gsp : (p / hw) * gsn + (p % hw) * gsp);
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < k1) {
          const float gv = g[k * gsk];
          bacc[k] += gv;
#pragma unroll
          for (int e = 0; e < EPU; ++e) acc[k][e] += gv * Elem<T>::ld(v + e);
        }
    }
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {  // static indexing: a runtime k would push acc[][] to scratch memory
    if (k < k1) {
      __syncthreads();
      if (pl < lanes) {
#pragma unroll
        for (int e = 0; e < EPU; ++e) shd[pl * shs + u * EPU + e] = acc[k][e];
        if (u == 0) shd[pl * shs + c0] = bacc[k];
      }
      __syncthreads();
      if (threadIdx.x <= c0) {
        float t = 0.f;
        for (int j = 0; j < lanes; ++j) t += shd[j * shs + threadIdx.x];
        part[((size_t)blockIdx.x * k1 + k) * (c0 + 1) + threadIdx.x] = t;
      }
    }
  }
}

#define HEAD_BWD_BLOCKS 2048
extern "C" int mia_head_bwd_workspace(int c0, int k1) { return HEAD_BWD_BLOCKS * k1 * (c0 + 1); }

extern "C" int mia_head_bwd(const float* dlogits, const void* x, int dtype, const float* w, void* dx, float* dw, float* db,
                            float* workspace, int n, int64_t hw, int c0, int k1, int64_t gsn, int64_t gsk, int64_t gsp,
                            int accumulate, void* stream) {
  MIA_CHECK_ARG(dlogits && x && w && dw && db && workspace && n > 0 && hw > 0 && c0 > 0, "mia_head_bwd: bad arguments");
  MIA_CHECK_ARG(k1 >= 1 && k1 <= MAXK, "mia_head_bwd: k1=%d not in [1,%d]", k1, MAXK);
  const int64_t npix = (int64_t)n * hw;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t total = npix * c0;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const int wblocks = (int)(npix / 64 < 1 ? 1 : (npix / 64 > HEAD_BWD_BLOCKS ? HEAD_BWD_BLOCKS : npix / 64));
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  const bool vec = (c0 % epu == 0) && c0 <= 255 && c0 / epu <= 128 && (256 / (c0 / epu)) <= 128 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  const int upp = ((reinterpret_cast<uintptr_t>(dx) & 15) == 0) ? head_fast_upp(dtype, c0, k1, npix, gsn, gsp, hw, x) : 0;
  if (upp) {
    const int lanes = 256 / upp;
    const int iblocks = (int)((npix + lanes - 1) / lanes < 16384 ? (npix + lanes - 1) / lanes : 16384);
    if (dtype == MIA_BF16) {
      if (dx) HEAD_DISPATCH(head_bwd_input_fast_kernel, bf16_t, dim3(iblocks), dlogits, w, static_cast<bf16_t*>(dx), (int)npix, gsp, gsk);
      HEAD_DISPATCH(head_bwd_weight_fast_kernel, bf16_t, dim3(wblocks), dlogits, static_cast<const bf16_t*>(x), workspace, (int)npix, gsp, gsk);
    } else {
      if (dx) HEAD_DISPATCH(head_bwd_input_fast_kernel, float, dim3(iblocks), dlogits, w, static_cast<float*>(dx), (int)npix, gsp, gsk);
      HEAD_DISPATCH(head_bwd_weight_fast_kernel, float, dim3(wblocks), dlogits, static_cast<const float*>(x), workspace, (int)npix, gsp, gsk);
    }
  } else if (vec && (dtype == MIA_BF16 || dtype == MIA_F32)) {
    const int64_t units = npix * (c0 / epu);
    const int vblocks = (int)((units + 255) / 256 < 16384 ? (units + 255) / 256 : 16384);
    if (dtype == MIA_BF16) {
      if (dx) hipLaunchKernelGGL(head_bwd_input_vec_kernel<bf16_t>, dim3(vblocks), dim3(256), k1 * c0 * 4, st, dlogits, w, static_cast<bf16_t*>(dx), npix, c0, k1, gsp, gsk, gsn, hw);
      hipLaunchKernelGGL(head_bwd_weight_vec_kernel<bf16_t>, dim3(wblocks), dim3(256), (256 / (c0 / epu)) * (c0 + 1) * 4, st, dlogits, static_cast<const bf16_t*>(x), workspace, npix, c0, k1, gsp, gsk, gsn, hw);
    } else {
      if (dx) hipLaunchKernelGGL(head_bwd_input_vec_kernel<float>, dim3(vblocks), dim3(256), k1 * c0 * 4, st, dlogits, w, static_cast<float*>(dx), npix, c0, k1, gsp, gsk, gsn, hw);
      hipLaunchKernelGGL(head_bwd_weight_vec_kernel<float>, dim3(wblocks), dim3(256), (256 / (c0 / epu)) * (c0 + 1) * 4, st, dlogits, static_cast<const float*>(x), workspace, npix, c0, k1, gsp, gsk, gsn, hw);
    }
  } else if (dtype == MIA_BF16) {
    if (dx) hipLaunchKernelGGL(head_bwd_input_kernel<bf16_t>, dim3(blocks), dim3(256), k1 * c0 * 4, st, dlogits, w, static_cast<bf16_t*>(dx), npix, c0, k1, gsp, gsk, gsn, hw);
    hipLaunchKernelGGL(head_bwd_weight_kernel<bf16_t>, dim3(wblocks), dim3(256), 512 * 4, st, dlogits, static_cast<const bf16_t*>(x), workspace, npix, c0, k1, gsp, gsk, gsn, hw);
  } else if (dtype == MIA_F32) {
    if (dx) hipLaunchKernelGGL(head_bwd_input_kernel<float>, dim3(blocks), dim3(256), k1 * c0 * 4, st, dlogits, w, static_cast<float*>(dx), npix, c0, k1, gsp, gsk, gsn, hw);
    hipLaunchKernelGGL(head_bwd_weight_kernel<float>, dim3(wblocks), dim3(256), 512 * 4, st, dlogits, static_cast<const float*>(x), workspace, npix, c0, k1, gsp, gsk, gsn, hw);
  } else { mia_set_error("mia_head_bwd: bad dtype"); return MIA_EARG; }
  hipLaunchKernelGGL(head_bwd_final_kernel, dim3(ceil_div(k1 * (c0 + 1), 16)), dim3(256), 0, st, workspace, wblocks, k1, c0, dw, db, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// 0 = not eligible for the fused (norm + LeakyReLU + head) kernels, else units per pixel
extern "C" int mia_head_norm_eligible(int dtype, int n, int64_t hw, int c0, int k1) {
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  if ((dtype != MIA_BF16 && dtype != MIA_F32) || c0 % epu != 0) return 0;
  const int upp = c0 / epu;
  if ((upp != 4 && upp != 8 && upp != 12 && upp != 16) || k1 < 2 || k1 > 4) return 0;
  if (hw >= ((int64_t)1 << 31) || n > HEAD_BWD_BLOCKS) return 0;
  return upp;
}
static int head_norm_slabs(int n, int64_t hw) {
  int64_t sl = HEAD_BWD_BLOCKS / n;
  const int64_t maxsl = hw / 512 > 0 ? hw / 512 : 1;
  if (sl > maxsl) sl = maxsl;
  return (int)(sl < 1 ? 1 : sl);
}

// logits = W * lrelu(scale*y + shift) + b   (y = raw conv output of the last decoder block, NHWC; scale / shift [N][C0])
extern "C" int mia_head_norm_fwd(const void* y, int dtype, const float* scale, const float* shift, float slope, const float* w,
                                 const float* b, float* logits, int n, int64_t hw, int c0, int k1, int64_t osn, int64_t osk,
                                 int64_t osp, void* stream) {
  MIA_CHECK_ARG(y && scale && shift && w && b && logits && n > 0 && hw > 0, "mia_head_norm_fwd: bad arguments");
  const int upp = mia_head_norm_eligible(dtype, n, hw, c0, k1);
  MIA_CHECK_ARG(upp > 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "mia_head_norm_fwd: shape not eligible (c0=%d k1=%d)", c0, k1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int slabs = head_norm_slabs(n, hw);
  if (dtype == MIA_BF16) HEAD_DISPATCH_N(head_norm_fwd_kernel, bf16_t, dim3(n * slabs), static_cast<const bf16_t*>(y), scale, shift, slope, w, b, logits, (int)hw, slabs, osn, osp, osk);
  else HEAD_DISPATCH_N(head_norm_fwd_kernel, float, dim3(n * slabs), static_cast<const float*>(y), scale, shift, slope, w, b, logits, (int)hw, slabs, osn, osp, osk);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// dW, db of the head with its input recomputed from y (workspace: mia_head_bwd_workspace floats)
extern "C" int mia_head_norm_wgrad(const float* dlogits, const void* y, int dtype, const float* scale, const float* shift,
                                   float slope, float* dw, float* db, float* workspace, int n, int64_t hw, int c0, int k1,
                                   int64_t gsn, int64_t gsk, int64_t gsp, int accumulate, void* stream) {
  MIA_CHECK_ARG(dlogits && y && scale && shift && dw && db && workspace && n > 0 && hw > 0, "mia_head_norm_wgrad: bad arguments");
  const int upp = mia_head_norm_eligible(dtype, n, hw, c0, k1);
  MIA_CHECK_ARG(upp > 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "mia_head_norm_wgrad: shape not eligible (c0=%d k1=%d)", c0, k1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int slabs = head_norm_slabs(n, hw);
  if (dtype == MIA_BF16) HEAD_DISPATCH_N(head_norm_wgrad_kernel, bf16_t, dim3(n * slabs), dlogits, static_cast<const bf16_t*>(y), scale, shift, slope, workspace, (int)hw, slabs, gsn, gsp, gsk);
  else HEAD_DISPATCH_N(head_norm_wgrad_kernel, float, dim3(n * slabs), dlogits, static_cast<const float*>(y), scale, shift, slope, workspace, (int)hw, slabs, gsn, gsp, gsk);
  hipLaunchKernelGGL(head_bwd_final_kernel, dim3(ceil_div(k1 * (c0 + 1), 16)), dim3(256), 0, st, workspace, n * slabs, k1, c0, dw, db, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- fused Dice + CE
// flags
#define LF_SOFTMAX 1
#define LF_DO_BG 2
#define LF_BATCH 4
#define LF_SQUARED 8
#define LF_DENSE 16  // `labels` points at a contiguous fp32 [B][K1][HW] target (already one-hot / soft; dice_loss.py:40-41 skips the encoder)

struct LossGeom { int64_t sn, sk, sp; };  // element strides of the logits tensor: image, class, pixel

// forward partials: part[b][slab][k][3] (I, sum_p, sum_t) and cepart[b][slab]
__global__ void dice_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int64_t hw, int k1,
                                   LossGeom g, int flags, int slabs, float* __restrict__ part, float* __restrict__ cepart,
                                   int* __restrict__ bad_label) {
  __shared__ float red[16];
  const int b = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float si[MAXK], sp[MAXK], st[MAXK], ce = 0.f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) { si[k] = 0.f; sp[k] = 0.f; st[k] = 0.f; }
  const float* base = logits + b * g.sn;
  for (int64_t p = r0 + threadIdx.x; p < r1; p += blockDim.x) {
    float v[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { v[k] = base[p * g.sp + k * g.sk]; mx = fmaxf(mx, v[k]); }
    const float* dense = reinterpret_cast<const float*>(labels) + (int64_t)b * k1 * hw + p;
    const long long lab = (flags & LF_DENSE) ? 0 : labels[(int64_t)b * hw + p];
    if (lab < 0 || lab >= k1) { *bad_label = 1; continue; }  // finalize poisons the loss with NaN (see there)
    float pr[MAXK];
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { pr[k] = __expf(v[k] - mx); se += pr[k]; }
    const float inv = 1.f / se;
    const float lse = mx + __logf(se);
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float pk = (flags & LF_SOFTMAX) ? pr[k] * inv : v[k];
        const float t = (flags & LF_DENSE) ? dense[k * hw] : ((k == (int)lab) ? 1.f : 0.f);
        si[k] += pk * t;
        sp[k] += (flags & LF_SQUARED) ? pk * pk : pk;
        st[k] += (flags & LF_SQUARED) ? t * t : t;
        ce += t * (lse - v[k]);
      }
  }
  for (int k = 0; k < k1; ++k) {
    float* dst = part + (((size_t)b * slabs + s) * k1 + k) * 3;
    float r;
    r = block_sum(si[k], red); if (threadIdx.x == 0) dst[0] = r;
    r = block_sum(sp[k], red); if (threadIdx.x == 0) dst[1] = r;
    r = block_sum(st[k], red); if (threadIdx.x == 0) dst[2] = r;
  }
  const float r = block_sum(ce, red);
  if (threadIdx.x == 0) cepart[(size_t)b * slabs + s] = r;
}


// Fast path: channels-last logits (class stride 1, pixel stride K1), int64 index labels, K1 in {2,3,4}, hw % 4 == 0.
// A thread owns FOUR consecutive pixels per step: K1 16-byte loads of logits + two 16-byte loads of labels (the generic
// kernel issues K1 + 2 four-byte loads per pixel), two steps in flight.  One block = one slab of one image; all 3*K1+1
// block sums share ONE barrier (per-wave DPP sums -> LDS -> 3*K1+1 threads add four waves).
template <int K1>
__global__ __launch_bounds__(256) void dice_ce_fwd_fast_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                               int hw, int flags, int slabs, float* __restrict__ part,
                                                               float* __restrict__ cepart, int* __restrict__ bad_label) {
  constexpr int NV = 3 * K1 + 1;
  __shared__ float red[4][NV];
  const int b = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int quads = hw >> 2;
  const int per = (quads + slabs - 1) / slabs, q0 = s * per, q1 = q0 + per < quads ? q0 + per : quads;
  const f32x4* lg = reinterpret_cast<const f32x4*>(logits + (size_t)b * hw * K1);
  const u32x4* lb = reinterpret_cast<const u32x4*>(labels + (size_t)b * hw);
  float si[K1], sp[K1], st[K1], ce = 0.f;
#pragma unroll
  for (int k = 0; k < K1; ++k) { si[k] = 0.f; sp[k] = 0.f; st[k] = 0.f; }
  bool bad = false;
  auto one = [&](const f32x4* f, const u32x4& l0, const u32x4& l1) {
    const unsigned lo[4] = {l0[0], l0[2], l1[0], l1[2]}, hi[4] = {l0[1], l0[3], l1[1], l1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[K1];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < K1; ++k) { v[k] = f[(j * K1 + k) >> 2][(j * K1 + k) & 3]; mx = fmaxf(mx, v[k]); }
      const bool ok = hi[j] == 0u && lo[j] < (unsigned)K1;
      bad |= !ok;
      float pr[K1], se = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) { pr[k] = __expf(v[k] - mx); se += pr[k]; }
      const float inv = ok ? 1.f / se : 0.f;   // an out-of-range label drops the pixel (and poisons the loss in finalize)
      const float lse = mx + __logf(se);
#pragma unroll
      for (int k = 0; k < K1; ++k) {
        const float pk = (flags & LF_SOFTMAX) ? pr[k] * inv : (ok ? v[k] : 0.f);
        const float t = (ok && lo[j] == (unsigned)k) ? 1.f : 0.f;
        si[k] += pk * t;
        sp[k] += (flags & LF_SQUARED) ? pk * pk : pk;
        st[k] += t;
        ce += t * (lse - v[k]);
      }
    }
  };
  int q = q0 + threadIdx.x;
  for (; q + 256 < q1; q += 512) {
    f32x4 fa[K1], fb[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) { fa[k] = lg[(size_t)q * K1 + k]; fb[k] = lg[(size_t)(q + 256) * K1 + k]; }
    const u32x4 a0 = lb[2 * (size_t)q], a1 = lb[2 * (size_t)q + 1], b0 = lb[2 * (size_t)(q + 256)], b1 = lb[2 * (size_t)(q + 256) + 1];
    one(fa, a0, a1);
    one(fb, b0, b1);
  }
  if (q < q1) {
    f32x4 fa[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) fa[k] = lg[(size_t)q * K1 + k];
    const u32x4 a0 = lb[2 * (size_t)q], a1 = lb[2 * (size_t)q + 1];
    one(fa, a0, a1);
  }
  if (bad) *bad_label = 1;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    const float a = wave_sum(si[k]), c = wave_sum(sp[k]), d = wave_sum(st[k]);
    if (l == 0) { red[w][3 * k] = a; red[w][3 * k + 1] = c; red[w][3 * k + 2] = d; }
  }
  ce = wave_sum(ce);
  if (l == 0) red[w][3 * K1] = ce;
  __syncthreads();
  if (threadIdx.x < NV) {
    const float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x < 3 * K1) part[((size_t)b * slabs + s) * K1 * 3 + threadIdx.x] = r;
    else cepart[(size_t)b * slabs + s] = r;
  }
}

template <int K1>
__global__ __launch_bounds__(256) void dice_ce_bwd_fast_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                               const float* __restrict__ coef, const float* __restrict__ gout,
                                                               float* __restrict__ dl, int nb, int hw, int flags, float dice_w,
                                                               float ce_w) {
  const int b = blockIdx.y;
  const int quads = hw >> 2;
  const float go_s = gout ? gout[0] : 1.f;
  const float cew = go_s * ce_w / (float)((double)nb * (double)hw);
  float al[K1], be[K1];
#pragma unroll
  for (int k = 0; k < K1; ++k) { al[k] = go_s * dice_w * coef[((size_t)b * K1 + k) * 2]; be[k] = go_s * dice_w * coef[((size_t)b * K1 + k) * 2 + 1]; }
  const f32x4* lg = reinterpret_cast<const f32x4*>(logits + (size_t)b * hw * K1);
  const u32x4* lb = reinterpret_cast<const u32x4*>(labels + (size_t)b * hw);
  f32x4* dst = reinterpret_cast<f32x4*>(dl + (size_t)b * hw * K1);
  for (int q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) {
    f32x4 f[K1], o[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k) f[k] = lg[(size_t)q * K1 + k];
    const u32x4 l0 = lb[2 * (size_t)q], l1 = lb[2 * (size_t)q + 1];
    const unsigned lo[4] = {l0[0], l0[2], l1[0], l1[2]}, hi[4] = {l0[1], l0[3], l1[1], l1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[K1], pr[K1], gk[K1];
      float mx = -INFINITY, se = 0.f, dot = 0.f;
#pragma unroll
      for (int k = 0; k < K1; ++k) { v[k] = f[(j * K1 + k) >> 2][(j * K1 + k) & 3]; mx = fmaxf(mx, v[k]); }
#pragma unroll
      for (int k = 0; k < K1; ++k) { pr[k] = __expf(v[k] - mx); se += pr[k]; }
      const bool ok = hi[j] == 0u && lo[j] < (unsigned)K1;
      const float inv = 1.f / se;
#pragma unroll
      for (int k = 0; k < K1; ++k) {
        pr[k] *= inv;
        const float pk = (flags & LF_SOFTMAX) ? pr[k] : v[k];
        const float t = (lo[j] == (unsigned)k) ? 1.f : 0.f;
        gk[k] = al[k] * t + be[k] * ((flags & LF_SQUARED) ? 2.f * pk : 1.f);
        dot += gk[k] * pr[k];
      }
#pragma unroll
      for (int k = 0; k < K1; ++k) {
        const float t = (lo[j] == (unsigned)k) ? 1.f : 0.f;
        const float dd = (flags & LF_SOFTMAX) ? pr[k] * (gk[k] - dot) : gk[k];
        o[(j * K1 + k) >> 2][(j * K1 + k) & 3] = ok ? dd + cew * (pr[k] - t) : 0.f;  // dropped pixel: same as the forward
      }
    }
#pragma unroll
    for (int k = 0; k < K1; ++k) dst[(size_t)q * K1 + k] = o[k];
  }
}

static bool dice_ce_fast_ok(const void* logits, const void* labels, int64_t hw, int k1, int64_t sn, int64_t sk, int64_t sp, int flags) {
  return !(flags & LF_DENSE) && k1 >= 2 && k1 <= 4 && sk == 1 && sp == k1 && sn == hw * k1 && (hw & 3) == 0 && hw < ((int64_t)1 << 30) &&
         (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(labels) & 15) == 0;
}

// finalize: sums[b][k][3]; coef[b][k][2] = (alpha, beta) with dDice/dp_k(pixel) = alpha*t (+2p*... if squared) + beta
// out[0] = total loss, out[1] = ce, out[2] = dice
__global__ void dice_ce_finalize_kernel(const float* __restrict__ part, const float* __restrict__ cepart, int nb, int slabs,
                                        int k1, int64_t hw, int flags, float smooth, float dice_w, float ce_w,
                                        float* __restrict__ sums, float* __restrict__ coef, float* __restrict__ out,
                                        int* __restrict__ bad_label) {
  // single block; thread -> (b,k)
  __shared__ double dsum[256];
  __shared__ double cesum[256];
  const int kb = (flags & LF_DO_BG) ? 0 : 1;
  const int nk = k1 - kb;
  const int total = nb * k1;
  double mydice = 0.0, myce = 0.0;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int b = i / k1, k = i % k1;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int s = 0; s < slabs; ++s) {
      const float* p = part + (((size_t)b * slabs + s) * k1 + k) * 3;
      a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    sums[i * 3 + 0] = (float)a0; sums[i * 3 + 1] = (float)a1; sums[i * 3 + 2] = (float)a2;
  }
  for (int i = threadIdx.x; i < nb * slabs; i += blockDim.x) myce += cepart[i];
  __syncthreads();
  // dice terms (thread per (b,k) for !batch, per k for batch)
  if (flags & LF_BATCH) {
    for (int k = kb + threadIdx.x; k < k1; k += blockDim.x) {
      double I = 0, P = 0, Tt = 0;
      for (int b = 0; b < nb; ++b) { I += sums[(b * k1 + k) * 3]; P += sums[(b * k1 + k) * 3 + 1]; Tt += sums[(b * k1 + k) * 3 + 2]; }
      I /= nb; P /= nb; Tt /= nb;
      const double num = 2 * I + smooth, den = P + Tt + smooth;
      mydice += (1.0 - num / den) / nk;
      // d(dice_k)/dI_b = -(2/den)/nb ; d/dP_b = (num/den^2)/nb ; loss = mean_k
      for (int b = 0; b < nb; ++b) {
        coef[(b * k1 + k) * 2 + 0] = (float)(-(2.0 / den) / nb / nk);
        coef[(b * k1 + k) * 2 + 1] = (float)((num / (den * den)) / nb / nk);
      }
    }
  } else {
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int k = i % k1;
      if (k < kb) continue;
      const double I = sums[i * 3], P = sums[i * 3 + 1], Tt = sums[i * 3 + 2];
      const double num = 2 * I + smooth, den = P + Tt + smooth;
      mydice += (1.0 - num / den) / ((double)nb * nk);
      coef[i * 2 + 0] = (float)(-(2.0 / den) / ((double)nb * nk));
      coef[i * 2 + 1] = (float)((num / (den * den)) / ((double)nb * nk));
    }
  }
  if (!(flags & LF_DO_BG))
    for (int b = threadIdx.x; b < nb; b += blockDim.x) { coef[(b * k1) * 2] = 0.f; coef[(b * k1) * 2 + 1] = 0.f; }
  dsum[threadIdx.x] = mydice; cesum[threadIdx.x] = myce;
  __syncthreads();
  if (threadIdx.x == 0) {
    double d = 0, c = 0;
    for (int i = 0; i < blockDim.x; ++i) { d += dsum[i]; c += cesum[i]; }
    c /= ((double)nb * (double)hw);
    out[1] = (float)c; out[2] = (float)d;
    out[0] = (float)(ce_w * c + dice_w * d);
  }
  // A label outside [0, K1): the reference raises (scatter index error in DiceLoss, dice_loss.py:25-30; target bound check
  // in CrossEntropyLoss).  A device kernel cannot raise, so the result is made unusable instead of silently training on
  // such masks: loss values and the backward coefficients (hence every gradient) become NaN; ops.DiceCEFn.check_labels()
  // turns the flag into an exception at the caller's next host sync.
  // bad_label[0] is the working flag the pixel kernels raise; it is OR-ed into bad_label[1] -- the STICKY verdict that
  // check_labels reads and clears, so a clean forward (validation, a second loss term, a deep-supervision head) between the
  // offending call and the check cannot erase it -- and re-armed here, so the caller never has to clear it between calls.
  __syncthreads();
  const int bad = bad_label[0];
  __syncthreads();
  if (threadIdx.x == 0) { if (bad) bad_label[1] = 1; bad_label[0] = 0; }
  if (bad) {
    const float qn = __builtin_nanf("");
    if (threadIdx.x < 3) out[threadIdx.x] = qn;
    for (int i = threadIdx.x; i < total * 2; i += blockDim.x) coef[i] = qn;
  }
}

// backward: dlogits[b,p,k] = gout * ( ce_w/(B*HW) * (softmax_k - t_k) + dice_w * dDice/dlogit_k )
__global__ void dice_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                   const float* __restrict__ coef, const float* __restrict__ gout, float* __restrict__ dl,
                                   int nb, int64_t hw, int k1, LossGeom g, LossGeom go, int flags, float dice_w, float ce_w) {
  const int64_t total = (int64_t)nb * hw;
  const float go_s = gout ? gout[0] : 1.f;
  const float cew = ce_w / (float)((double)nb * (double)hw);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / hw);
    const int64_t p = i - (int64_t)b * hw;
    const float* src = logits + b * g.sn + p * g.sp;
    float v[MAXK], pr[MAXK];
    float mx = -INFINITY, se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { v[k] = src[k * g.sk]; mx = fmaxf(mx, v[k]); }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { pr[k] = __expf(v[k] - mx); se += pr[k]; }
    const float inv = 1.f / se;
    const float* dense = reinterpret_cast<const float*>(labels) + (int64_t)b * k1 * hw + p;
    const int lab = (flags & LF_DENSE) ? 0 : (int)labels[i];
    // dL/dp_k for the dice part
    float gk[MAXK], tk[MAXK], dot = 0.f, tsum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float sm = pr[k] * inv;
        const float pk = (flags & LF_SOFTMAX) ? sm : v[k];
        const float al = coef[((size_t)b * k1 + k) * 2], be = coef[((size_t)b * k1 + k) * 2 + 1];
        tk[k] = (flags & LF_DENSE) ? dense[k * hw] : (k == lab ? 1.f : 0.f);
        tsum += tk[k];
        float gg = al * tk[k] + be * ((flags & LF_SQUARED) ? 2.f * pk : 1.f);
        gk[k] = gg * dice_w;
        pr[k] = sm;
        dot += gk[k] * sm;
      }
    float* dst = dl + b * go.sn + p * go.sp;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float dd = (flags & LF_SOFTMAX) ? pr[k] * (gk[k] - dot) : gk[k];
        const float dc = cew * (tsum * pr[k] - tk[k]);  // d/dv_k of sum_j t_j (lse - v_j)
        dst[k * go.sk] = go_s * (dd + dc);
      }
  }
}

extern "C" int mia_dice_ce_workspace(int nb, int k1, int slabs) { return nb * slabs * (k1 * 3 + 1); }

// sums: [B][K1][3], coef: [B][K1][2], out: [3] (loss, ce, dice), bad_label: int flag (device)
extern "C" int mia_dice_ce_fwd(const float* logits, const long long* labels, int nb, int64_t hw, int k1, int64_t sn, int64_t sk,
                               int64_t sp, int flags, float smooth, float dice_w, float ce_w, int slabs, float* workspace,
                               float* sums, float* coef, float* out, int* bad_label, void* stream) {
  MIA_CHECK_ARG(logits && labels && workspace && sums && coef && out && bad_label, "mia_dice_ce_fwd: null pointer");
  MIA_CHECK_ARG(nb > 0 && hw > 0 && slabs > 0, "mia_dice_ce_fwd: bad shape");
  MIA_CHECK_ARG(k1 >= 1 && k1 <= MAXK, "mia_dice_ce_fwd: k1=%d not in [1,%d]", k1, MAXK);
  hipStream_t st = static_cast<hipStream_t>(stream);
  LossGeom g{sn, sk, sp};
  float* part = workspace;
  float* cepart = workspace + (size_t)nb * slabs * k1 * 3;
  if (dice_ce_fast_ok(logits, labels, hw, k1, sn, sk, sp, flags)) {
    if (k1 == 2) hipLaunchKernelGGL(dice_ce_fwd_fast_kernel<2>, dim3(nb * slabs), dim3(256), 0, st, logits, labels, (int)hw, flags, slabs, part, cepart, bad_label);
    else if (k1 == 3) hipLaunchKernelGGL(dice_ce_fwd_fast_kernel<3>, dim3(nb * slabs), dim3(256), 0, st, logits, labels, (int)hw, flags, slabs, part, cepart, bad_label);
    else hipLaunchKernelGGL(dice_ce_fwd_fast_kernel<4>, dim3(nb * slabs), dim3(256), 0, st, logits, labels, (int)hw, flags, slabs, part, cepart, bad_label);
  } else {
    hipLaunchKernelGGL(dice_ce_fwd_kernel, dim3(nb * slabs), dim3(256), 0, st, logits, labels, hw, k1, g, flags, slabs, part, cepart, bad_label);
  }
  hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(256), 0, st, part, cepart, nb, slabs, k1, hw, flags, smooth, dice_w, ce_w, sums, coef, out, bad_label);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_dice_ce_bwd(const float* logits, const long long* labels, const float* coef, const float* grad_out,
                               float* dlogits, int nb, int64_t hw, int k1, int64_t sn, int64_t sk, int64_t sp, int64_t gsn,
                               int64_t gsk, int64_t gsp, int flags, float dice_w, float ce_w, void* stream) {
  MIA_CHECK_ARG(logits && labels && coef && dlogits && nb > 0 && hw > 0, "mia_dice_ce_bwd: bad arguments");
  MIA_CHECK_ARG(k1 >= 1 && k1 <= MAXK, "mia_dice_ce_bwd: k1=%d not in [1,%d]", k1, MAXK);
  const int64_t total = (int64_t)nb * hw;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  LossGeom g{sn, sk, sp}, go{gsn, gsk, gsp};
  if (dice_ce_fast_ok(logits, labels, hw, k1, sn, sk, sp, flags) && gsn == sn && gsk == sk && gsp == sp &&
      (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int quads = (int)(hw >> 2);
    const dim3 grid((unsigned)(quads + 255) / 256 < 512u ? (unsigned)(quads + 255) / 256 : 512u, (unsigned)nb);
    if (k1 == 2) hipLaunchKernelGGL(dice_ce_bwd_fast_kernel<2>, grid, dim3(256), 0, st, logits, labels, coef, grad_out, dlogits, nb, (int)hw, flags, dice_w, ce_w);
    else if (k1 == 3) hipLaunchKernelGGL(dice_ce_bwd_fast_kernel<3>, grid, dim3(256), 0, st, logits, labels, coef, grad_out, dlogits, nb, (int)hw, flags, dice_w, ce_w);
    else hipLaunchKernelGGL(dice_ce_bwd_fast_kernel<4>, grid, dim3(256), 0, st, logits, labels, coef, grad_out, dlogits, nb, (int)hw, flags, dice_w, ce_w);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels, coef,
                     grad_out, dlogits, nb, hw, k1, g, go, flags, dice_w, ce_w);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
