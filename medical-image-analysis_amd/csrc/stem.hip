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
#include "options.h"
#include <stdlib.h>

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

// ---- stem forward on the fp32 matrix cores.  The VALU kernel above needs 72 FMAs (+ 9 image loads in each of the 8 threads
// that share a pixel) per 16-byte unit it writes and runs at 2.6 TB/s; v_mfma_f32_16x16x4_f32 is EXACT fp32 (an fmaf chain,
// MI355X_MICROARCH.md) at the vector rate but leaves the VALU free, so the kernel becomes a pure stream:
//   out[co][px] = sum_tap w[co][tap] * img[px + tap]  =  A(co x tap) * B(tap x px),  K = 9 taps padded to 12 = three MFMAs
// per 16-pixel run and 16-channel tile.  A = weights (lane: row = channel, k = tap), resident in 3 registers per channel tile;
// B = one image value per lane (lane: k = tap, column = pixel), fetched straight from global / L1 with a per-lane-group tap
// offset (image border -> buffer-load zero); C starts as the bias.  D has a lane's four registers = four consecutive
// channels of one pixel, so bf16 output goes out in 16-byte stores (v_permlane16_swap pairs the channel tiles), fp32 output
// directly.  No LDS, no barrier in the main loop: a wave walks its own (row, 16-pixel strip) units of the workgroup's band.
typedef __amdgpu_buffer_rsrc_t srsrc_t;
template <typename T, int MT>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w /*[C0][9]*/,
                                                            const float* __restrict__ bias, T* __restrict__ y, int h, int wd,
                                                            int slabs, float* __restrict__ stats /*[N][slabs][C0][2]*/) {
  constexpr int C0 = 16 * MT;
  __shared__ float red[4][2][C0];
  __shared__ __attribute__((aligned(16))) unsigned char tbuf[4 * 16 * (2 * C0 + 16)];  // per-wave store transpose (bf16 path)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int band = (h + slabs - 1) / slabs, rb = s * band, re = rb + band < h ? rb + band : h;
  const int strips = (wd + 15) / 16;
  const size_t hw = (size_t)h * wd;
  const srsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)n * hw), 0, (int)(hw * 4), 0x00020000);
  // A fragments: wa[m][g] = w[16m + c16][4g + q] (taps >= 9 are zero columns)
  float wa[MT][3];
  f32x4 bv[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int g = 0; g < 3; ++g) { const int t = 4 * g + q; wa[m][g] = t < 9 ? w[(16 * m + c16) * 9 + t] : 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[m][r] = bias ? bias[16 * m + 4 * q + r] : 0.f;
  }
  // tap of this lane in K-group g: (dy, dx) in -1..1
  int tdy[3], tdx[3];
  bool tok[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) { const int t = 4 * g + q; tok[g] = t < 9; tdy[g] = (t < 9 ? t / 3 : 1) - 1; tdx[g] = (t < 9 ? t % 3 : 1) - 1; }
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.f; s2[m][r] = 0.f; }
  const int units = (re > rb ? re - rb : 0) * strips;
  T* yimg = y + (size_t)n * hw * C0;
  auto load_b = [&](int u, float* b) {
    const int row = rb + u / strips, x0 = (u % strips) * 16;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int yy = row + tdy[g], xx = x0 + c16 + tdx[g];
      const bool ok = u < units && tok[g] && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd;
      b[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (yy * wd + xx) * 4 : (int)0xFFFFFFF0u, 0, 0));
    }
  };
  float bnext[3];
  load_b(wave, bnext);
  for (int u = wave; u < units; u += 4) {
    const int row = rb + u / strips, x0 = (u % strips) * 16;
    float b[3] = {bnext[0], bnext[1], bnext[2]};
    load_b(u + 4, bnext);  // the next unit's image values travel behind this unit's MFMAs and stores
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][0], b[0], bv[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][1], b[1], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][2], b[2], acc[m], 0, 0, 0);
    }
    const bool colok = x0 + c16 < wd;
    const float msk = colok ? 1.f : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = acc[m][r] * msk; s1[m][r] += v; s2[m][r] += v * acc[m][r]; }
    const size_t pix = (size_t)row * wd + x0 + c16;
    if constexpr (sizeof(T) == 4) {
      if (colok)
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(yimg + pix * C0 + 16 * m + 4 * q) = acc[m];
    } else {
      // bf16: a lane holds 8 bytes (4 channels) per channel tile.  Even channel-tile counts go through a wave-private LDS
      // transpose (16 pixels x C0 bf16 at a 16-byte-padded pitch) so that every global store covers WHOLE 2*C0-byte pixel
      // lines (8 pixels x 128 B contiguous per instruction at C0 = 64) instead of 64-byte halves; no barrier: one wave's LDS
      // operations execute in order.
      typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
      typedef float f2_t __attribute__((ext_vector_type(2)));
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      auto pk = [](float a, float c) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{a, c}, bf2_t)); };
      if constexpr ((MT & 1) == 0) {
        constexpr int PITCH = 2 * C0 + 16;  // bytes per pixel row in LDS
        unsigned char* t = tbuf + wave * (16 * PITCH);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          *reinterpret_cast<u32x2*>(t + c16 * PITCH + 32 * m + 8 * q) = u32x2{pk(acc[m][0], acc[m][1]), pk(acc[m][2], acc[m][3])};
        __builtin_amdgcn_wave_barrier();  // wave-private transpose: write -> cross-lane read order stated in the program
        constexpr int UPP = 2 * MT;  // 16-byte units per pixel
#pragma unroll
        for (int it = 0; it < UPP * 16 / 64; ++it) {
          const int idx = lane + 64 * it, pxl = idx / UPP, ch = idx - pxl * UPP;
          const u32x4 d = *reinterpret_cast<const u32x4*>(t + pxl * PITCH + 16 * ch);
          if (x0 + pxl < wd) *reinterpret_cast<u32x4*>(yimg + ((size_t)row * wd + x0 + pxl) * C0 + 8 * ch) = d;
        }
        __builtin_amdgcn_wave_barrier();  // ... and read -> next round's write
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const u32x2 d = {pk(acc[m][0], acc[m][1]), pk(acc[m][2], acc[m][3])};
          if (colok) *reinterpret_cast<u32x2*>(yimg + pix * C0 + 16 * m + 4 * q) = d;
        }
      }
    }
  }
  if (stats != nullptr) {
    // sums over the 16 pixel columns of a lane row (DPP), then over the four waves (LDS)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[m][r], c = s2[m][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        if (c16 == 0) { red[wave][0][16 * m + 4 * q + r] = a; red[wave][1][16 * m + 4 * q + r] = c; }
      }
    __syncthreads();
    for (int i = tid; i < 2 * C0; i += 256) {
      const int k = i / C0, ch = i % C0;
      stats[(((size_t)n * slabs + s) * C0 + ch) * 2 + k] = red[0][k][ch] + red[1][k][ch] + red[2][k][ch] + red[3][k][ch];
    }
  }
}

// partial[blk][9][c0] = sum over the block's pixels of x[p + t] * dy[p][co]
// FUSED (mia_stem_wgrad_fused): `dy` holds dz, the gradient w.r.t. the block's ACTIVATED output, and the kernel forms the conv
// output gradient on load from (dz, y, coefficients) with norm_bwd_dy -- rounded to T exactly as the streaming apply pass would
// have stored it -- so that pass (read dz + y, write dy) and this kernel's read of dy collapse into one read of dz + y.
struct StemFuse {
  const void* y; const float* scale; const float* shift; const float* xa; const float* xb; const float* c1; const float* c2; float slope;
};

template <typename T, typename TI, bool FUSED = false>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const TI* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                         int slabs, int h, int wd, int c0, int win_rows, StemFuse fz) {
  constexpr int EPU = Elem<T>::EPU;
  extern __shared__ float red[];  // [lanes][c0 + 1], then (win_rows > 0) the slab's image window [win_rows][wd + 2]
  const int upp = c0 / EPU, lanes = 256 / upp;
  const int u = threadIdx.x % upp, pl = threadIdx.x / upp;
  const int n = blockIdx.x / slabs, s = blockIdx.x % slabs;  // block = one pixel slab of one image
  const int hw = h * wd, per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float acc[9][EPU];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < EPU; ++e) acc[t][e] = 0.f;
  float fsc[EPU], fsf[EPU], fka[EPU], fkb[EPU];  // FUSED: this thread's channel unit of image n
  const T* yrow = nullptr;
  if constexpr (FUSED) {
    if (pl < lanes) {
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        const size_t o = (size_t)n * c0 + u * EPU + e;
        fsc[e] = fz.scale[o]; fsf[e] = fz.shift[o];
        fka[e] = -fsc[e] * fz.c2[o] * fz.xa[o];
        fkb[e] = -fsc[e] * (fz.c1[o] + fz.c2[o] * fz.xb[o]);
      }
    }
    yrow = static_cast<const T*>(fz.y) + ((size_t)n * hw) * c0 + u * EPU;
  }
  // g (the block's conv-output gradient for this thread's EPU channels of one pixel) from the loaded unit(s)
  auto grad_of = [&](const u32x4& raw, const u32x4& yraw, float* gf) {
    alignas(16) T g[EPU]; alignas(16) T yv[EPU];
    *reinterpret_cast<u32x4*>(g) = raw;
    *reinterpret_cast<u32x4*>(yv) = yraw;
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
      if constexpr (FUSED) {  // dy rounded to T exactly as the apply pass would have stored it
        const T r = Elem<T>::cvt(norm_bwd_dy(Elem<T>::ld(g + e), Elem<T>::ld(yv + e), fsc[e], fsf[e], fka[e], fkb[e], fz.slope));
        gf[e] = Elem<T>::ld(&r);
      } else gf[e] = Elem<T>::ld(g + e);
    }
  };
  if (win_rows > 0 && r0 < r1) {
    // The nine image taps of a pixel come from LDS: the slab's rows (one above, one below, a zero column either side) are
    // staged once per block, so a tap is a broadcast ds_read with no bounds test.  As nine global loads per pixel group
    // (eight lanes, same address) the kernel was bound by load-instruction issue: 0.30 ms for the 570 MB it streams.
    float* win = red + lanes * (c0 + 1);
    const int wp = wd + 2, py0 = r0 / wd;
    const TI* img = x + (size_t)n * hw;
    for (int i = threadIdx.x; i < win_rows * wp; i += 256) {
      const int wy = i / wp, wx = i - wy * wp;
      const int yy = py0 - 1 + wy, xx = wx - 1;
      win[i] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) ? Elem<TI>::ld(img + (size_t)yy * wd + xx) : 0.f;
    }
    __syncthreads();
    if (pl < lanes) {
      const T* grow = dy + ((size_t)n * hw) * c0 + u * EPU;
      int p = r0 + pl;
      int py = p / wd, px = p - py * wd;
      auto body = [&](const u32x4& raw, const u32x4& yraw, int wy, int wx) {  // (wy, wx): window coordinates of the pixel's top-left tap
        float gf[EPU];
        grad_of(raw, yraw, gf);
        const float* wrow = win + wy * wp + wx;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float xv = wrow[(t / 3) * wp + t % 3];
#pragma unroll
          for (int e = 0; e < EPU; ++e) acc[t][e] += xv * gf[e];
        }
      };
      const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
      auto ldy = [&](int pp) -> u32x4 { if constexpr (FUSED) return *reinterpret_cast<const u32x4*>(yrow + (size_t)pp * c0); else return zero; };
      for (; p + lanes < r1; p += 2 * lanes) {  // two dy (FUSED: dz + y) loads in flight
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(grow + (size_t)p * c0);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(grow + (size_t)(p + lanes) * c0);
        const u32x4 y0 = ldy(p), y1 = ldy(p + lanes);
        body(a0, y0, py - py0, px);
        px += lanes;
        while (px >= wd) { px -= wd; ++py; }
        body(a1, y1, py - py0, px);
        px += lanes;
        while (px >= wd) { px -= wd; ++py; }
      }
      if (p < r1) body(*reinterpret_cast<const u32x4*>(grow + (size_t)p * c0), ldy(p), py - py0, px);
    }
  } else if (pl < lanes && r0 + pl < r1) {
    const TI* img = x + (size_t)n * hw;
    const T* grow = dy + ((size_t)n * hw) * c0 + u * EPU;
    int p = r0 + pl;
    int py = p / wd, px = p - py * wd;  // one division per thread; the walk below is incremental
    for (; p < r1; p += lanes) {
      float gf[EPU];
      u32x4 yraw = u32x4{0u, 0u, 0u, 0u};
      if constexpr (FUSED) yraw = *reinterpret_cast<const u32x4*>(yrow + (size_t)p * c0);
      grad_of(*reinterpret_cast<const u32x4*>(grow + (size_t)p * c0), yraw, gf);
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

#define STEM_SLABS 128
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
  // fp32 image, channel count a multiple of 16: the matrix-core kernel (exact fp32 arithmetic, HBM-bound)
  const int use_mfma = mia_options().stem_mfma;
  const bool al16 = (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  if (use_mfma && x_dtype == MIA_F32 && al16 && (c0 == 16 || c0 == 32 || c0 == 64 || c0 == 96 || c0 == 128) && (int64_t)h * wd * 4 < ((int64_t)1 << 31)) {
#define SM(T, MT) hipLaunchKernelGGL((stem_fwd_mfma_kernel<T, MT>), grid, dim3(256), 0, st, static_cast<const float*>(x), w, bias, static_cast<T*>(y), h, wd, STEM_SLABS, stat_partials)
#define SMT(T) do { if (c0 == 16) SM(T, 1); else if (c0 == 32) SM(T, 2); else if (c0 == 64) SM(T, 4); else if (c0 == 96) SM(T, 6); else SM(T, 8); } while (0)
    if (dtype == MIA_BF16) SMT(bf16_t); else SMT(float);
#undef SMT
#undef SM
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
#define SF(T, TI) hipLaunchKernelGGL((stem_fwd_kernel<T, TI>), grid, dim3(256), shb, st, static_cast<const TI*>(x), w, bias, static_cast<T*>(y), h, wd, c0, STEM_SLABS, stat_partials)
  if (dtype == MIA_BF16 && x_dtype == MIA_F32) SF(bf16_t, float);
  else if (dtype == MIA_BF16) SF(bf16_t, bf16_t);
  else if (x_dtype == MIA_F32) SF(float, float);
  else SF(float, bf16_t);
#undef SF
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

static int stem_wgrad_run(const void* x, int x_dtype, const void* dy, int dtype, float* workspace, float* grad, int n, int h, int wd,
                          int c0, int accumulate, void* stream, const StemFuse* fuse) {
  const int epu = dtype == MIA_BF16 ? 8 : 4, lanes = 256 / (c0 / epu);
  // one block = one pixel slab of one image (n * slabs <= STEM_WBLOCKS partial rows in the workspace)
  const int64_t hw = (int64_t)h * wd;
  int slabs = STEM_WBLOCKS / n;
  if (slabs > (int)((hw + 255) / 256)) slabs = (int)((hw + 255) / 256);
  if (slabs < 1) slabs = 1;
  const int blocks = n * slabs;
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t shb = (size_t)lanes * (c0 + 1) * 4;
  // image window in LDS when a slab's rows (+ halo) fit next to the reduction buffer
  const int64_t per = (hw + slabs - 1) / slabs;
  int win_rows = (int)((per + wd - 1) / wd) + 3;  // rows a slab can touch (it may start mid-row) + one above and below
  if ((size_t)win_rows * (wd + 2) * 4 + shb <= 60 * 1024) shb += (size_t)win_rows * (wd + 2) * 4;
  else win_rows = 0;
  const StemFuse fz = fuse ? *fuse : StemFuse{};
#define SW(T, TI, F) hipLaunchKernelGGL((stem_wgrad_kernel<T, TI, F>), dim3(blocks), dim3(256), shb, st, static_cast<const TI*>(x), static_cast<const T*>(dy), workspace, slabs, h, wd, c0, win_rows, fz)
#define SWF(T, TI) do { if (fuse) SW(T, TI, true); else SW(T, TI, false); } while (0)
  if (dtype == MIA_BF16 && x_dtype == MIA_F32) SWF(bf16_t, float);
  else if (dtype == MIA_BF16) SWF(bf16_t, bf16_t);
  else if (x_dtype == MIA_F32) SWF(float, float);
  else SWF(float, bf16_t);
#undef SWF
#undef SW
  hipLaunchKernelGGL(stem_wgrad_final_kernel, dim3(ceil_div(9 * c0, 16)), dim3(256), 0, st, workspace, blocks, c0, grad, accumulate);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_stem_wgrad(const void* x, int x_dtype, const void* dy, int dtype, float* workspace, float* grad, int n, int h, int wd,
                              int c0, int accumulate, void* stream) {
  MIA_CHECK_ARG(x && dy && workspace && grad && n > 0 && h > 0 && wd > 0, "mia_stem_wgrad: bad arguments");
  MIA_CHECK_ARG((dtype == MIA_BF16 || dtype == MIA_F32) && (x_dtype == MIA_BF16 || x_dtype == MIA_F32), "mia_stem_wgrad: bad dtype");
  MIA_CHECK_ARG(stem_ok(dtype, c0), "mia_stem_wgrad: c0=%d must be a multiple of the 16-byte unit and <= 256", c0);
  MIA_CHECK_ARG((int64_t)h * wd < ((int64_t)1 << 31) && n <= STEM_WBLOCKS, "mia_stem_wgrad: image or batch too large");
  return stem_wgrad_run(x, x_dtype, dy, dtype, workspace, grad, n, h, wd, c0, accumulate, stream, nullptr);
}

// Weight gradient of the stem with the block's norm + LeakyReLU backward folded in (see stem_wgrad_kernel<.., FUSED>).
extern "C" int mia_stem_wgrad_fused(const void* x, int x_dtype, const void* dz, const void* y, int dtype, const float* scale,
                                    const float* shift, const float* xa, const float* xb, const float* c1, const float* c2, float slope,
                                    float* workspace, float* grad, int n, int h, int wd, int c0, int accumulate, void* stream) {
  MIA_CHECK_ARG(x && dz && y && scale && shift && xa && xb && c1 && c2 && workspace && grad && n > 0 && h > 0 && wd > 0,
                "mia_stem_wgrad_fused: bad arguments");
  MIA_CHECK_ARG((dtype == MIA_BF16 || dtype == MIA_F32) && (x_dtype == MIA_BF16 || x_dtype == MIA_F32), "mia_stem_wgrad_fused: bad dtype");
  MIA_CHECK_ARG(stem_ok(dtype, c0), "mia_stem_wgrad_fused: c0=%d must be a multiple of the 16-byte unit and <= 256", c0);
  MIA_CHECK_ARG((int64_t)h * wd < ((int64_t)1 << 31) && n <= STEM_WBLOCKS, "mia_stem_wgrad_fused: image or batch too large");
  MIA_CHECK_ARG(((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "mia_stem_wgrad_fused: dz / y must be 16-byte aligned");
  const StemFuse fz{y, scale, shift, xa, xb, c1, c2, slope};
  return stem_wgrad_run(x, x_dtype, dz, dtype, workspace, grad, n, h, wd, c0, accumulate, stream, &fz);
}
