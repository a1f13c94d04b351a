// Implicit-GEMM convolution on MFMA for gfx950 (NHWC activations, packed [tap][n][k] weights).
//
// One kernel template covers every dense conv-shaped product of the UNet hot path
// (reference: src/models/unet/blocks.py:83-90 Conv2d 3x3, unet.py:142 ConvTranspose2d 2x2/s2):
//   MODE_G3S1  gather 3x3 stride 1 pad 1   -> conv fwd (s=1), conv dgrad (s=1, flipped taps)
//   MODE_G3S2  gather 3x3 stride 2 pad 1   -> conv fwd (s=2)
//   MODE_G2S2  gather 2x2 stride 2 pad 0   -> ConvTranspose2d dgrad
//   MODE_T3S2  transposed 3x3 stride 2     -> conv dgrad (s=2), one output-parity class per blockIdx.y
//   MODE_T2S2  transposed 2x2 stride 2     -> ConvTranspose2d fwd (pixel-shuffle store)
//   MODE_G1    1x1                          -> pointwise conv (ResidualBlock skip / wide heads)
//
// Data path: a 256-thread workgroup (4 waves) owns TH x 16 output pixels x BN output channels.  Per
// 64-byte channel chunk the input halo tile and the weight slice are staged global -> VGPR -> LDS as
// 16-byte units laid out [channel-group plane][pixel] so that every MFMA operand fragment is ONE
// ds_read_b128.  The M-row <-> pixel map is permuted (pi()) and plane pitches are == 2 (mod 16)
// units, which makes both the b128 fragment reads and the b128 staging writes bank-conflict free
// (MI355X LDS: 16-lane groups {0-3,12-15,20-27}... for ds_read_b128, 8x8 contiguous lanes for writes).
// bf16: v_mfma_f32_16x16x32_bf16, one per 32-channel block; fp32: 4 x v_mfma_f32_16x16x4_f32 per
// 16-channel block (exact fp32 fma chains).  Epilogue: +bias, optional per-(image,tile,channel)
// sum / sum-of-squares partials for the norm layer, tile transposed through LDS, 16-byte stores.
#include <stdlib.h>

#include "conv_common.h"
#include "options.h"

template <typename T, int MODE, int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_mma_kernel(const ConvArgs a) {
  using G = Geo<MODE, MT>;
  constexpr int TH = G::TH, BN = 16 * NT, EPU = Elem<T>::EPU, KB = 4 * EPU;
  constexpr int NPA = G::NPA, NPB = BN + 2, PITCH = G::PITCH, S = G::S, IW = G::IW, IH = G::IH;
  constexpr int OSTR = BN + EPU;  // out-tile row stride (elements)
  constexpr int A_UNITS = 4 * NPA, B_UNITS = G::MAXTAPS * 4 * NPB;
  constexpr int STAGE_BYTES = (A_UNITS + B_UNITS) * 16;
  constexpr int OUT_BYTES = TH * 16 * OSTR * (int)sizeof(T);
  constexpr int LDS_BYTES = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + 2 * 4 * BN * 4];
  u32x4* ldsA = reinterpret_cast<u32x4*>(smem);
  u32x4* ldsB = ldsA + A_UNITS;
  T* ldsO = reinterpret_cast<T*>(smem);
  float* ldsR = reinterpret_cast<float*>(smem + LDS_BYTES);  // [2][4 waves][BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r16 = lane & 15;
  const int pr = pi16(r16);

  // ---- block -> (image, tile, n-block, parity)
  int bid = blockIdx.x;
  const int nb = bid % a.nblk_n; bid /= a.nblk_n;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid;
  const int n0 = nb * BN;
  const int ph = (MODE == MODE_T3S2 || MODE == MODE_T2S2) ? (int)(blockIdx.y >> 1) : 0;
  const int pw = (MODE == MODE_T3S2 || MODE == MODE_T2S2) ? (int)(blockIdx.y & 1) : 0;
  const int oy0 = ty * TH, ox0 = tx * 16;  // tile origin in the tile domain
  const int iy0 = MODE == MODE_G3S1 ? oy0 - 1 : MODE == MODE_G3S2 ? 2 * oy0 - 1 : MODE == MODE_G2S2 ? 2 * oy0 : oy0;
  const int ix0 = MODE == MODE_G3S1 ? ox0 - 1 : MODE == MODE_G3S2 ? 2 * ox0 - 1 : MODE == MODE_G2S2 ? 2 * ox0 : ox0;

  // tap enumeration (uniform per block)
  const int nth = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : MODE == MODE_T3S2 ? (ph ? 2 : 1) : 1;
  const int ntw = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : MODE == MODE_T3S2 ? (pw ? 2 : 1) : 1;
  auto tap_w = [&](int ta, int tb) -> int {  // index into the packed weight's tap dimension
    if (MODE == MODE_G3S1) { const int t = ta * 3 + tb; return a.flip ? 8 - t : t; }
    if (MODE == MODE_G3S2) return ta * 3 + tb;
    if (MODE == MODE_G2S2) return ta * 2 + tb;
    if (MODE == MODE_T3S2) { const int kh = ph ? (ta == 0 ? 0 : 2) : 1; const int kw = pw ? (tb == 0 ? 0 : 2) : 1; return kh * 3 + kw; }
    if (MODE == MODE_T2S2) return ph * 2 + pw;
    return 0;
  };
  auto tap_off = [&](int ta, int tb) -> int {  // LDS pixel offset of the tap
    if (MODE == MODE_G3S1) return ta * PITCH + tb;
    if (MODE == MODE_G3S2 || MODE == MODE_G2S2) return ta * PITCH + (tb & 1) * G::IWH + (tb >> 1);
    if (MODE == MODE_T3S2) { const int dh = ph ? (ta == 0 ? 1 : 0) : 0; const int dw = pw ? (tb == 0 ? 1 : 0) : 0; return dh * PITCH + dw; }
    return 0;
  };

  const int ctot = a.c1 + a.c2;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* wp = static_cast<const T*>(a.wp);

  // ---- software pipeline: chunk k+1 is fetched global -> VGPR while chunk k's MFMAs run from LDS
  constexpr int A_N = IH * IW * 4, A_IT = (A_N + 255) / 256;
  constexpr int B_NMAX = G::MAXTAPS * 4 * BN, B_IT = (B_NMAX + 255) / 256;
  const int ntaps = nth * ntw;
  const int b_n = ntaps * 4 * BN;
  // chunk-invariant addressing of this thread's staging units (32-bit element offsets; LDS slots are recomputed)
  constexpr unsigned OOB = 0xFFFFFFFFu;
  unsigned a_pix[A_IT];  // global pixel index, OOB = outside the image / not owned (zero fill)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int u = tid + i * 256;
    a_pix[i] = OOB;
    if (u < A_N) {
      const int pix = u >> 2;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) a_pix[i] = (unsigned)(((int64_t)img * a.Hin + gy) * a.Win + gx);
    }
  }
  auto b_src = [&](int u) -> size_t {  // element offset into wp of (tap, n0+n, 0) + g*EPU (recomputed: saves VGPRs)
    const int g = u & 3, n = (u >> 2) % BN, tl = u / (4 * BN);
    const int ta = tl / ntw, tb = tl - ta * ntw;
    return ((size_t)tap_w(ta, tb) * a.npad + n0 + n) * a.kpad + g * EPU;
  };
  u32x4 pa[A_IT], pb[B_IT];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int g = (tid + i * 256) & 3;
      const int c = c0 + g * EPU;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (a_pix[i] != OOB && c < ctot) {
        const size_t p = a_pix[i];
        if (a.vec_in) {
          const T* src = (c < a.c1) ? in1 + p * a.c1 + c : in2 + p * a.c2 + (c - a.c1);
          v = *reinterpret_cast<const u32x4*>(src);
        } else {
          alignas(16) T tmp[EPU];
#pragma unroll
          for (int e = 0; e < EPU; ++e) {
            const int ce = c + e;
            T val = (T)0;
            if (ce < a.c1) val = in1[p * a.c1 + ce];
            else if (ce < ctot) val = in2[p * a.c2 + (ce - a.c1)];
            tmp[e] = val;
          }
          v = *reinterpret_cast<const u32x4*>(tmp);
        }
      }
      pa[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (tid + i * 256 < b_n) pb[i] = *reinterpret_cast<const u32x4*>(wp + b_src(tid + i * 256) + c0);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int u = tid + i * 256;
      if (u < A_N) {
        const int pix = u >> 2;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int lidx = (S == 2) ? iy * PITCH + (ix & 1) * G::IWH + (ix >> 1) : iy * PITCH + ix;
        ldsA[(u & 3) * NPA + lidx] = pa[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int u = tid + i * 256;
      if (u < b_n) ldsB[((u / (4 * BN)) * 4 + (u & 3)) * NPB + ((u >> 2) % BN)] = pb[i];
    }
  };

  fetch(0);
  for (int c0 = 0; c0 < ctot; c0 += KB) {
    commit();
    __syncthreads();
    if (c0 + KB < ctot) fetch(c0 + KB);

    // ---- MFMA over taps
    for (int ta = 0; ta < nth; ++ta) {
      for (int tb = 0; tb < ntw; ++tb) {
        const int tl = ta * ntw + tb;
        const int toff = tap_off(ta, tb);
        u32x4 bf[NT], af[MT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = ldsB[(tl * 4 + q) * NPB + n * 16 + pr];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = ldsA[q * NPA + S * (wave * MT + m) * PITCH + toff + pr];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = Mma<T>::run(af[m], bf[n], acc[m][n]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, stats partials, transpose through LDS, coalesced store
  const bool tmode = (MODE == MODE_T3S2 || MODE == MODE_T2S2);
  const int hd = tmode ? (a.Hout - ph + 1) / 2 : a.Hout;  // tile-domain extent
  const int wd = tmode ? (a.Wout - pw + 1) / 2 : a.Wout;
  const int nout = a.o1 + a.o2;
  float s1[NT], s2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { s1[n] = 0.f; s2[n] = 0.f; }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int nn = n * 16 + pr;  // B column r16 <-> out channel pi(r16)
    const float b = (a.bias != nullptr && n0 + nn < nout) ? a.bias[n0 + nn] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int y = wave * MT + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = pi16(4 * q + r);
        const float v = acc[m][n][r] + b;
        if (oy0 + y < hd && ox0 + px < wd) { s1[n] += v; s2[n] += v * v; }
        ldsO[(y * 16 + px) * OSTR + nn] = Elem<T>::cvt(v);
      }
    }
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float t1 = s1[n], t2 = s2[n];
      t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) { ldsR[(0 * 4 + wave) * BN + n * 16 + pr] = t1; ldsR[(1 * 4 + wave) * BN + n * 16 + pr] = t2; }
    }
  }
  __syncthreads();
  if (a.stats != nullptr && tid < BN && n0 + tid < nout) {
    const float t1 = ldsR[0 * BN + tid] + ldsR[1 * BN + tid] + ldsR[2 * BN + tid] + ldsR[3 * BN + tid];
    const float t2 = ldsR[4 * BN + tid] + ldsR[5 * BN + tid] + ldsR[6 * BN + tid] + ldsR[7 * BN + tid];
    const size_t tile = (size_t)img * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx;
    float* dst = a.stats + (tile * nout + n0 + tid) * 2;
    dst[0] = t1; dst[1] = t2;
  }
  T* out1 = static_cast<T*>(a.out1);
  T* out2 = static_cast<T*>(a.out2);
  constexpr int UPP = BN / EPU;  // 16-byte units per pixel in the out tile
  for (int u = tid; u < TH * 16 * UPP; u += 256) {
    const int cu = u % UPP, pl = u / UPP;
    const int y = pl >> 4, px = pl & 15;
    if (oy0 + y >= hd || ox0 + px >= wd) continue;
    const int oy = tmode ? 2 * (oy0 + y) + ph : oy0 + y;
    const int ox = tmode ? 2 * (ox0 + px) + pw : ox0 + px;
    const size_t p = ((size_t)img * a.Hout + oy) * a.Wout + ox;
    const int n = n0 + cu * EPU;
    if (n >= nout) continue;
    const T* srow = ldsO + pl * OSTR + cu * EPU;
    if (a.vec_out) {
      T* dst = (n < a.o1) ? out1 + p * a.o1 + n : out2 + p * a.o2 + (n - a.o1);
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(srow);
    } else {
#pragma unroll
      for (int e = 0; e < EPU; ++e) {
        const int ne = n + e;
        if (ne < a.o1) out1[p * a.o1 + ne] = srow[e];
        else if (ne < nout) out2[p * a.o2 + (ne - a.o1)] = srow[e];
      }
    }
  }
}

template <typename T, int MODE, int MT, int NT>
static void launch(const ConvArgs& a, int grid_y, hipStream_t st) {
  const int grid_x = a.N * a.tiles_x * a.tiles_y * a.nblk_n;
  hipLaunchKernelGGL((conv_mma_kernel<T, MODE, MT, NT>), dim3(grid_x, grid_y), dim3(256), 0, st, a);
}

template <typename T, int MODE, int MT>
static void launch_nt(const ConvArgs& a, int nt, int grid_y, hipStream_t st) {
  if (nt == 4) launch<T, MODE, MT, 4>(a, grid_y, st);
  else if (nt == 2) launch<T, MODE, MT, 2>(a, grid_y, st);
  else launch<T, MODE, MT, 1>(a, grid_y, st);
}

template <typename T, int MODE>
static void launch_mt(const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st) {
  if constexpr (MODE == MODE_G3S2 || MODE == MODE_G2S2) {
    launch_nt<T, MODE, 2>(a, nt, grid_y, st);
  } else {
    if (mt >= 4) launch_nt<T, MODE, 4>(a, nt, grid_y, st);
    else launch_nt<T, MODE, 2>(a, nt, grid_y, st);
  }
}

template <typename T>
static int dispatch(int mode, const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st) {
  switch (mode) {
    case MODE_G3S1: launch_mt<T, MODE_G3S1>(a, mt, nt, grid_y, st); break;
    case MODE_G3S2: launch_mt<T, MODE_G3S2>(a, mt, nt, grid_y, st); break;
    case MODE_G2S2: launch_mt<T, MODE_G2S2>(a, mt, nt, grid_y, st); break;
    case MODE_T3S2: launch_mt<T, MODE_T3S2>(a, mt, nt, grid_y, st); break;
    case MODE_T2S2: launch_mt<T, MODE_T2S2>(a, mt, nt, grid_y, st); break;
    case MODE_G1: launch_mt<T, MODE_G1>(a, mt, nt, grid_y, st); break;
    default: return MIA_EARG;
  }
  return MIA_OK;
}

extern "C" int mia_amax(const float* x, int64_t n, void* slot, int reset, void* stream);

// Tile-domain geometry shared with the host (stats buffer sizing): see include/mia_hip.h.
static void conv_tiles(const MiaOptions& opt, int mode, int hout, int wout, int* tiles_y, int* tiles_x, int* tile_h) {
  const bool tmode = (mode == MODE_T3S2 || mode == MODE_T2S2);
  const int hd = tmode ? (hout + 1) / 2 : hout, wd = tmode ? (wout + 1) / 2 : wout;
  int mt = (mode == MODE_G3S2 || mode == MODE_G2S2) ? 2 : (hd > 8 ? 4 : 2);
  const int th = 4 * mt;
  if (tiles_y) *tiles_y = ceil_div(hd, th);
  if (tiles_x) *tiles_x = ceil_div(wd, 16);
  if (tile_h) *tile_h = th;
}

extern "C" int mia_conv_mma_tiles(int mode, int hout, int wout, int* tiles_y, int* tiles_x, int* tile_h) {
  const MiaOptions opt = mia_options();
  conv_tiles(opt, mode, hout, wout, tiles_y, tiles_x, tile_h);
  return MIA_OK;
}

static int conv_mma_run(int mode, int dtype, const void* in1, int c1, const void* in2, int c2, const void* wpack,
                        int npad, int kpad, int flip_taps, const float* bias, void* out1, int o1, void* out2,
                        int o2, float* stat_partials, int n, int hin, int win, int hout, int wout, void* stream,
                        const float* nl_scale, const float* nl_shift, float nl_slope, const void* cr_y = nullptr,
                        const float* const* cr_coef = nullptr, float cr_slope = 0.f, int acc_out = 0, const void* amax_in1 = nullptr,
                        const void* amax_in2 = nullptr, const void* amax_w = nullptr, void* amax_out1 = nullptr, void* amax_out2 = nullptr,
                        const void* wpack_split = nullptr) {
  MIA_CHECK_ARG(mode >= 0 && mode <= MODE_G1, "mia_conv_mma: bad mode %d", mode);
  MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_conv_mma: bad dtype %d", dtype);
  MIA_CHECK_ARG(in1 && wpack && out1 && c1 > 0 && o1 > 0 && c2 >= 0 && o2 >= 0, "mia_conv_mma: null/empty operand");
  MIA_CHECK_ARG((c2 == 0) == (in2 == nullptr) && (o2 == 0) == (out2 == nullptr), "mia_conv_mma: split operand mismatch");
  MIA_CHECK_ARG(n > 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, "mia_conv_mma: bad shape");
  const int epu = dtype == MIA_BF16 ? 8 : 4, kb = 4 * epu;
  MIA_CHECK_ARG(npad % 64 == 0 && npad >= o1 + o2, "mia_conv_mma: npad=%d must be a multiple of 64 >= %d", npad, o1 + o2);
  MIA_CHECK_ARG(kpad % kb == 0 && kpad >= c1 + c2, "mia_conv_mma: kpad=%d must be a multiple of %d >= %d", kpad, kb, c1 + c2);
  // shape contract per mode
  bool ok = true;
  switch (mode) {
    case MODE_G3S1: case MODE_G1: ok = (hout == hin && wout == win); break;
    case MODE_G3S2: ok = (hout == (hin + 1) / 2 && wout == (win + 1) / 2); break;
    case MODE_G2S2: ok = (hin == 2 * hout && win == 2 * wout); break;
    case MODE_T3S2: ok = (hin == (hout + 1) / 2 && win == (wout + 1) / 2); break;
    case MODE_T2S2: ok = (hout == 2 * hin && wout == 2 * win); break;
  }
  MIA_CHECK_ARG(ok, "mia_conv_mma: mode %d shape mismatch in %dx%d out %dx%d", mode, hin, win, hout, wout);
  const MiaOptions opt = mia_options();  // one snapshot per call
  ConvArgs a;
  a.in1 = in1; a.in2 = in2; a.c1 = c1; a.c2 = c2; a.wp = wpack; a.bias = bias;
  a.out1 = out1; a.out2 = out2; a.o1 = o1; a.o2 = o2; a.stats = stat_partials;
  a.N = n; a.Hin = hin; a.Win = win; a.Hout = hout; a.Wout = wout;
  a.npad = npad; a.kpad = kpad; a.flip = flip_taps;
  a.nl_scale = nl_scale; a.nl_shift = nl_shift; a.nl_slope = nl_slope;
  a.acc_out = acc_out;
  // fp32: split f16 products when the caller knows every operand's maximum (otherwise, or with the option off, exact fp32 MFMAs)
  a.split = (dtype == MIA_F32 && opt.f32_split && amax_in1 != nullptr && amax_w != nullptr && (c2 == 0 || amax_in2 != nullptr)) ? opt.f32_split : 0;  // 1: four products on interleaved words, 2: three on planes
  a.amax_in1 = static_cast<const unsigned*>(amax_in1); a.amax_in2 = static_cast<const unsigned*>(amax_in2);
  a.amax_w = static_cast<const unsigned*>(amax_w);
  if (dtype != MIA_F32) amax_out1 = amax_out2 = nullptr;
  if (cr_y != nullptr) {
    a.cr_y = cr_y; a.cr_scale = cr_coef[0]; a.cr_shift = cr_coef[1]; a.cr_xa = cr_coef[2]; a.cr_xb = cr_coef[3]; a.cr_slope = cr_slope;
  }
  int th;
  conv_tiles(opt, mode, hout, wout, &a.tiles_y, &a.tiles_x, &th);  // same snapshot as the launch below
  int mt = th / 4;
  const int nout = o1 + o2;
  const int nt = nout > 32 ? 4 : (nout > 16 ? 2 : 1);
  a.nblk_n = ceil_div(nout, 16 * nt);
  a.st_tiles_y = a.tiles_y;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec_in = (c1 % epu == 0) && (c2 % epu == 0) && al16(in1) && (in2 == nullptr || al16(in2));
  a.vec_out = (o1 % epu == 0) && (o2 % epu == 0) && al16(out1) && (out2 == nullptr || al16(out2));
  const bool tmode = (mode == MODE_T3S2 || mode == MODE_T2S2);
  a.xcd = opt.conv_xcd;
  const int grid_y = tmode ? 4 : 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  const bool fast = conv_mma_fast_eligible(dtype, a, nt);
  // stride-2 3x3 forward, bf16, 128-multiples of output channels: 512-thread workgroups on 16-row tiles (two statistics tiles each)
  // (measured per level, tools/s2_levels.py: 14-20 % faster from 128 input channels on; at 64 the two-chunk K loop leaves
  // the lone workgroup's prologue / epilogue exposed: 0.63 vs 0.62 ms, so the two-workgroups-per-CU shape keeps that launch)
  if (opt.conv_s2_wide && fast && mode == MODE_G3S2 && dtype == MIA_BF16 && nout % 128 == 0 && o1 % 128 == 0 && hout > 8 &&
      (c1 + c2 >= 128 || opt.conv_s2_wide >= 2)) {
    mt = 4;
    a.tiles_y = ceil_div(hout, 16);
    a.nblk_n = nout / 128;
  }
  // conv64_dma: 1 = the one-pass two-destination input gradient only (measured faster there), 2 = every 64 -> 64 launch
  if (acc_out) {  // out += result: the tile kernel's epilogue reads the previous values
    if (!fast) { mia_set_error("mia_conv_mma_acc: shape outside the accumulating kernel's contract"); return MIA_EUNSUPPORTED; }
    if (a.split && wpack_split != nullptr) { a.wp = wpack_split; a.wsplit = 1; }
    rc = conv_mma_fast_launch(mode, dtype, a, mt, nt, grid_y, st);
  }
#ifdef MIA_EXPERIMENTS
  else if (cr_y != nullptr) {  // column-reduce epilogue: the 64-channel register kernel (its epilogue overlaps the co-resident workgroup)
    if (!(mt == 4 && conv64_eligible(mode, dtype, a))) {
      mia_set_error("mia_conv_mma_cr: shape outside the column-reduce kernel's contract (ask mia_conv_cr_supported first)");
      return MIA_EUNSUPPORTED;
    }
    rc = conv64_launch(a, 0, opt.reserve_cus, st);
  }
#endif
  else if (nl_scale != nullptr) {  // normalise-on-load: the register-staged 64-channel kernel is the one consumer that transforms
    if (!(mt == 4 && conv64_eligible(mode, dtype, a))) {
      mia_set_error("mia_conv_mma_nl: shape outside the normalise-on-load kernel's contract (ask mia_conv_nl_supported first)");
      return MIA_EUNSUPPORTED;
    }
    rc = conv64_launch(a, 0, opt.reserve_cus, st);
  }
  else if (opt.conv64 && opt.conv64_dma && (opt.conv64_dma >= 2 || a.o2 != 0) && mt == 4 && conv64_dma_eligible(mode, dtype, a)) rc = conv64_dma_launch(a, opt.reserve_cus, st);
  else if (opt.conv64 && mt == 4 && conv64_eligible(mode, dtype, a)) rc = conv64_launch(a, 0, opt.reserve_cus, st);
  else if (opt.conv_bt && mt == 4 && conv_bt_eligible(mode, dtype, a)) rc = conv_bt_launch(a, opt.conv_bt_order, opt.reserve_cus, st);
  // (strided 3x3 forward as a tap-gathered GEMM: measured 0.62 -> 0.51, 0.44 -> 0.40, 0.37 -> 0.35 ms at 64 / 128 / 256 input channels,
  // 0.30 -> 0.30 at 512 in isolation (tools/s2_levels.py); step-time A/Bs could not resolve it (+-0.1 ms box noise), the per-kernel sums of
  // two interleaved rocprofv3 pairs inside the cfg3 step can: 36.90 / 36.91 -> 36.77 / 36.77 ms of kernels, so it is on;
  // conv_pw_s2 = 1 stops at 256 input channels, = 2 always)
  else if (opt.conv_pw && (mode != MODE_G3S2 || opt.conv_pw_s2 >= 2 || (opt.conv_pw_s2 == 1 && c1 <= 256)) && conv_pw_eligible(mode, dtype, a))
    rc = conv_pw_launch(mode, a, opt.reserve_cus, st);
  else if (fast) {
    if (a.split && wpack_split != nullptr) { a.wp = wpack_split; a.wsplit = 1; }
    if (dtype == MIA_F32) { a.amax_out1 = static_cast<unsigned*>(amax_out1); a.amax_out2 = static_cast<unsigned*>(amax_out2); amax_out1 = amax_out2 = nullptr; }
    rc = conv_mma_fast_launch(mode, dtype, a, mt, nt, grid_y, st);
  }
  else rc = dtype == MIA_BF16 ? dispatch<bf16_t>(mode, a, mt, nt, grid_y, st) : dispatch<float>(mode, a, mt, nt, grid_y, st);
  if (rc != MIA_OK) return rc;
  MIA_LAUNCH_CHECK();
  // output maxima the launched kernel did not fold in its epilogue (generic shapes): a separate pass each
  if (amax_out1 != nullptr) { rc = mia_amax(static_cast<const float*>(out1), (int64_t)n * hout * wout * o1, amax_out1, 0, stream); if (rc) return rc; }
  if (amax_out2 != nullptr && out2 != nullptr) { rc = mia_amax(static_cast<const float*>(out2), (int64_t)n * hout * wout * o2, amax_out2, 0, stream); if (rc) return rc; }
  return MIA_OK;
}

extern "C" int mia_conv_mma(int mode, int dtype, const void* in1, int c1, const void* in2, int c2, const void* wpack,
                            int npad, int kpad, int flip_taps, const float* bias, void* out1, int o1, void* out2,
                            int o2, float* stat_partials, int n, int hin, int win, int hout, int wout, const void* amax_in1,
                            const void* amax_in2, const void* amax_w, const void* wpack_split, void* amax_out1, void* amax_out2,
                            void* stream) {
  return conv_mma_run(mode, dtype, in1, c1, in2, c2, wpack, npad, kpad, flip_taps, bias, out1, o1, out2, o2, stat_partials, n, hin,
                      win, hout, wout, stream, nullptr, nullptr, 0.f, nullptr, nullptr, 0.f, 0, amax_in1, amax_in2, amax_w, amax_out1,
                      amax_out2, wpack_split);
}

// Normalise-on-load forward conv (the fused PlainBlock, SURVEY 8b export list "conv3x3_nhwc ... optional fused normalise +
// LeakyReLU on load taking per-(n,c) scale / shift"): see include/mia_hip.h.
extern "C" int mia_conv_nl_supported(int mode, int dtype, int c1, int nout, int hout, int wout) {
  if (mode != MODE_G3S1) return 0;
  return (dtype == MIA_BF16 && c1 == 64 && nout == 64 && hout > 8) ? 1 : 0;
}

extern "C" int mia_conv_mma_nl(int mode, int dtype, const void* y_in, int c1, const float* in_scale, const float* in_shift,
                               float slope, const void* wpack, int npad, int kpad, const float* bias, void* out, int nout,
                               float* stat_partials, int n, int hin, int win, int hout, int wout, void* stream) {
  MIA_CHECK_ARG(in_scale != nullptr && in_shift != nullptr, "mia_conv_mma_nl: null coefficient arrays");
  MIA_CHECK_ARG(slope >= 0.f && slope <= 1.f, "mia_conv_mma_nl: slope %g outside [0, 1]", (double)slope);
  MIA_CHECK_ARG(mia_conv_nl_supported(mode, dtype, c1, nout, hout, wout), "mia_conv_mma_nl: unsupported shape (mode %d dtype %d %d -> %d)",
                mode, dtype, c1, nout);
  return conv_mma_run(mode, dtype, y_in, c1, nullptr, 0, wpack, npad, kpad, 0, bias, out, nout, nullptr, 0, stat_partials, n, hin, win,
                      hout, wout, stream, in_scale, in_shift, slope);
}

#ifdef MIA_EXPERIMENTS  // probe builds only (tools/r5_store_hazard.sh): include/mia_hip_experiments.h
// Input gradient with the producing block's norm-backward reduction in its epilogue: see include/mia_hip.h.
extern "C" int mia_conv_cr_supported(int mode, int dtype, int c1, int nout, int hout, int wout) {
  (void)wout;
  return (mode == MODE_G3S1 && dtype == MIA_BF16 && c1 == 64 && nout == 64 && hout > 8) ? 1 : 0;
}

extern "C" int mia_conv_mma_cr(int mode, int dtype, const void* in1, int c1, const void* wpack, int npad, int kpad, int flip_taps,
                               void* out, int nout, const void* y_prod, const float* scale, const float* shift, const float* xa,
                               const float* xb, float slope, float* partials, int n, int hin, int win, int hout, int wout,
                               void* stream) {
  MIA_CHECK_ARG(y_prod && scale && shift && xa && xb && partials, "mia_conv_mma_cr: null pointer");
  MIA_CHECK_ARG(mia_conv_cr_supported(mode, dtype, c1, nout, hout, wout), "mia_conv_mma_cr: unsupported shape (mode %d dtype %d %d -> %d)",
                mode, dtype, c1, nout);
  MIA_CHECK_ARG((reinterpret_cast<uintptr_t>(y_prod) & 15) == 0 && (reinterpret_cast<uintptr_t>(scale) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(shift) & 15) == 0 && (reinterpret_cast<uintptr_t>(xa) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(xb) & 15) == 0, "mia_conv_mma_cr: 16-byte aligned tensors / coefficient rows required");
  const float* coef[4] = {scale, shift, xa, xb};
  return conv_mma_run(mode, dtype, in1, c1, nullptr, 0, wpack, npad, kpad, flip_taps, nullptr, out, nout, nullptr, 0, partials, n, hin,
                      win, hout, wout, stream, nullptr, nullptr, 0.f, y_prod, coef, slope);
}

#endif

// out += conv(in): see include/mia_hip.h.
extern "C" int mia_conv_acc_supported(int mode, int dtype, int c1, int nout) {
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  return ((mode == MODE_T3S2 || mode == MODE_G3S1) && (dtype == MIA_BF16 || dtype == MIA_F32) && c1 % (4 * epu) == 0 && nout % epu == 0) ? 1 : 0;
}

extern "C" int mia_conv_mma_acc(int mode, int dtype, const void* in1, int c1, const void* wpack, int npad, int kpad, int flip_taps,
                                void* out_inout, int nout, int n, int hin, int win, int hout, int wout, const void* amax_in,
                                const void* amax_w, const void* wpack_split, void* stream) {
  MIA_CHECK_ARG(mode == MODE_T3S2 || mode == MODE_G3S1, "mia_conv_mma_acc: mode %d not served", mode);
  return conv_mma_run(mode, dtype, in1, c1, nullptr, 0, wpack, npad, kpad, flip_taps, nullptr, out_inout, nout, nullptr, 0, nullptr, n, hin,
                      win, hout, wout, stream, nullptr, nullptr, 0.f, nullptr, nullptr, 0.f, 1, amax_in, nullptr, amax_w, nullptr, nullptr, wpack_split);
}
