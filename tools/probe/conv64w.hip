// Winograd F(2x2, 3x3) form of the 64 -> 64 channel 3x3 / stride-1 bf16 conv (the canonical block of the benchmark, reference
// src/models/unet/blocks.py:83-90) -- EXPERIMENT (option conv64_wino, off by default).
//
// Why: the board's power frontier (profiles/r04_mfma_power_frontier.txt) caps the fused 64-channel block near 0.43-0.52 of the
// HBM peak while the launch issues 618 GFLOP of MFMAs; F(2x2, 3x3) issues 275: per 2 x 2 output tile and channel pair, 16 products
// instead of 36.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        g: 3x3 filter, d: 4x4 input patch (stride 2 between patches), Y: 2x2 outputs
//
// One 512-thread workgroup per CU walks 16 x 16 output tiles in two passes of 8 rows (= 4 x 8 = 32 Winograd tiles):
//   stage   the (8 + 2) x (16 + 2) pixel patch, 64 channels bf16, global -> registers -> LDS (normalise-on-load here when the
//           input is the previous block's raw output, as conv64_persist_kernel<NL> does)
//   V       thread (tile, 4-channel group): 16 ds_read_b64 of its 4 x 4 patch, B^T d B in fp32 (32 adds per channel), 16 bf16
//           frequency planes back to LDS in MFMA-fragment order
//   MFMA    wave (tile group of 16, 16 output channels): U = G g G^T of its channels lives in 128 VGPRs for the whole launch
//           (computed once from the packed 3x3 weights); 16 frequencies x 2 channel halves = 32 MFMAs into 16 accumulators
//   output  A^T M A per lane (24 adds per accumulator register), + bias, statistics, bf16 store
// bf16 rounding of U and V costs accuracy: emulated rms error 3.6e-3 against 1.6e-3 for the direct bf16 conv.
#include "conv_common.h"
#include <type_traits>

typedef __amdgpu_buffer_rsrc_t w_rsrc_t;
#define WSENT64 0xFFFFFFF0u

namespace {

__device__ __forceinline__ w_rsrc_t w_make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned w_pack(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(wf32x2{a, b}, wbf16x2)); }
__device__ __forceinline__ float w_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float w_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

constexpr int WC = 64;                      // channels in == out
constexpr int WTH = 16, WTW = 16, WPH = 8;  // output tile; rows per pass
constexpr int RH = WPH + 2, RW = WTW + 2;   // raw patch of a pass
constexpr int RAW_BYTES = RH * RW * 128;    // [pixel][64 ch bf16], 16-byte chunks XOR-swizzled by (pixel & 7)
constexpr int NTILE = (WPH / 2) * (WTW / 2);  // 32 Winograd tiles per pass
constexpr int NPV = NTILE + 2;              // V plane pitch in 16-byte units (== 2 mod 16: conflict-free b128 fragment reads)
constexpr int V_BYTES = 16 * 8 * NPV * 16;  // [16 freq][8 channel chunks][tile]
constexpr int W_LDS = 2 * RAW_BYTES + V_BYTES + 2 * (2 * WC * 4) + 2 * WC * 2 * 4;  // two raw patches, V, two NL coefficient tables, statistics exchange
constexpr int RAW_UNITS = RH * RW * 8;      // 1440 16-byte units
constexpr int R_IT = (RAW_UNITS + 511) / 512;

}  // namespace

template <bool NL>
__global__ __launch_bounds__(512, 1) void conv64_wino_kernel(const ConvArgs a, int total_tiles, int tiles_per_img) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[W_LDS];
  unsigned char* raw0 = smem;                                           // two raw patches (the next step's lands while this one is transformed)
  unsigned char* vim = smem + 2 * RAW_BYTES;
  float* cf0 = reinterpret_cast<float*>(smem + 2 * RAW_BYTES + V_BYTES);  // NL: two tables of [0, 64) scale, [64, 128) shift, by step parity
  float* red = cf0 + 2 * 2 * WC;                                          // [2 tile groups][64 channels][2] statistics of a finished tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mg = wave >> 2, cw = wave & 3;  // tile group (16 Winograd tiles); 16 output channels
  const int q = lane >> 4, c16 = lane & 15;

  // ---- U = G g G^T for this wave's 16 output channels: lane (row = channel 16 cw + c16, k group q), both 32-channel halves
  const w_rsrc_t rsw = w_make_rsrc(a.wp, (unsigned)(9 * a.npad * WC * 2));
  u32x4 U[16][2];
  {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      u32x4 g[3][3];
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          const int t = ta * 3 + tb, tw = a.flip ? 8 - t : t;
          g[ta][tb] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((16 * cw + c16) * WC + 32 * kh + 8 * q) * 2, tw * a.npad * WC * 2, 0);
        }
#pragma unroll
      for (int d = 0; d < 4; ++d) {      // dword d = channels 2d, 2d + 1 of the lane's 8
        float u[2][4][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float gg[3][3], t[4][3];
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) gg[i][j] = e ? w_hi(g[i][j][d]) : w_lo(g[i][j][d]);
#pragma unroll
          for (int j = 0; j < 3; ++j) {  // G g: rows (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
            t[0][j] = gg[0][j];
            t[1][j] = 0.5f * (gg[0][j] + gg[1][j] + gg[2][j]);
            t[2][j] = 0.5f * (gg[0][j] - gg[1][j] + gg[2][j]);
            t[3][j] = gg[2][j];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // (G g) G^T
            u[e][i][0] = t[i][0];
            u[e][i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            u[e][i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            u[e][i][3] = t[i][2];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 0; m < 4; ++m) U[4 * i + m][kh][d] = w_pack(u[0][i][m], u[1][i][m]);
      }
    }
  }
  f32x4 bv;
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[16 * cw + 4 * q + r] : 0.f;

  const size_t ipix = (size_t)a.Hin * a.Win;
  const unsigned img_bytes = (unsigned)(ipix * WC * 2);
  const bf16_t* in = static_cast<const bf16_t*>(a.in1);
  bf16_t* out = static_cast<bf16_t*>(a.out1);

  // ---- a step = (tile, pass): 8 output rows of a 16 x 16 tile.  Staging: unit u = tid + 512 i -> (pixel = u >> 3, chunk = u & 7)
  const int nsteps = 2 * ((total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);  // this workgroup's tiles x 2 passes
  struct Step { int img, oy, ox, tile; };
  auto step_of = [&](int sidx) -> Step {
    const int t = (int)blockIdx.x + (sidx >> 1) * (int)gridDim.x;
    const int img = t / tiles_per_img, rem = t - img * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    return Step{img, ty * WTH + WPH * (sidx & 1), tx * WTW, t};
  };
  u32x4 pf[R_IT];
  auto fetch = [&](const Step& st) {
    const w_rsrc_t rs = w_make_rsrc(in + (size_t)st.img * ipix * WC, img_bytes);
#pragma unroll
    for (int i = 0; i < R_IT; ++i) {
      const int u = tid + 512 * i, pix = u >> 3, ch = u & 7;
      const int ry = pix / RW, rx = pix - ry * RW;
      const int gy = st.oy - 1 + ry, gx = st.ox - 1 + rx;
      const bool ok = u < RAW_UNITS && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
      pf[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (int)(((unsigned)(gy * a.Win + gx) * WC + ch * 8) * 2) : (int)WSENT64, 0, 0);
    }
  };
  auto commit = [&](const Step& st, unsigned char* raw, const float* cf) {
#pragma unroll
    for (int i = 0; i < R_IT; ++i) {
      const int u = tid + 512 * i, pix = u >> 3, ch = u & 7;
      if (u >= RAW_UNITS) continue;
      u32x4 v = pf[i];
      if constexpr (NL) {  // z = bf16(lrelu(scale * y + shift)), zero outside the image (padding of z, not of y)
        const int ry = pix / RW, rx = pix - ry * RW;
        const int gy = st.oy - 1 + ry, gx = st.ox - 1 + rx;
        const bool ok = (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cf + 8 * ch), s1 = *reinterpret_cast<const f32x4*>(cf + 8 * ch + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(cf + 64 + 8 * ch), h1 = *reinterpret_cast<const f32x4*>(cf + 64 + 8 * ch + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float sa = d < 2 ? s0[2 * d] : s1[2 * d - 4], sb = d < 2 ? s0[2 * d + 1] : s1[2 * d - 3];
          const float ha = d < 2 ? h0[2 * d] : h1[2 * d - 4], hb = d < 2 ? h0[2 * d + 1] : h1[2 * d - 3];
          const float va = __builtin_fmaf(sa, w_lo(v[d]), ha), vb = __builtin_fmaf(sb, w_hi(v[d]), hb);
          const unsigned r = w_pack(__builtin_fmaxf(va, va * a.nl_slope), __builtin_fmaxf(vb, vb * a.nl_slope));
          v[d] = ok ? r : 0u;
        }
      }
      *reinterpret_cast<u32x4*>(raw + pix * 128 + ((ch ^ (pix & 7)) * 16)) = v;
    }
  };
  auto load_table = [&](const Step& st, float* cf) {  // threads 0 .. 127: the image's scale / shift rows
    if constexpr (NL) {
      if (tid < 2 * WC) cf[tid] = (tid < WC ? a.nl_scale : a.nl_shift)[(size_t)st.img * WC + (tid & 63)];
    }
  };

  // ---- the V transform of this thread: Winograd tile tl = tid >> 4 (row tl >> 3, column tl & 7), channels 4 cg .. 4 cg + 3
  const int tl = tid >> 4, cg = tid & 15;
  const int tyy = tl >> 3, txx = tl & 7;
  auto transform = [&](const unsigned char* raw) {
    // one channel PAIR (one dword of the thread's 8-byte unit) at a time: 16 dword reads, B^T d B for two channels, 16 dword writes --
    // the 128 registers of U stay resident, so the transform has to live in the other half of the register file
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      unsigned w[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pix = (2 * tyy + i) * RW + 2 * txx + j;
          w[i][j] = *reinterpret_cast<const unsigned*>(raw + pix * 128 + (((cg >> 1) ^ (pix & 7)) * 16) + (cg & 1) * 8 + cp * 4);
        }
      float v[2][4][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // B^T d
          const float d0 = e ? w_hi(w[0][j]) : w_lo(w[0][j]), d1 = e ? w_hi(w[1][j]) : w_lo(w[1][j]);
          const float d2 = e ? w_hi(w[2][j]) : w_lo(w[2][j]), d3 = e ? w_hi(w[3][j]) : w_lo(w[3][j]);
          t[0][j] = d0 - d2;
          t[1][j] = d1 + d2;
          t[2][j] = d2 - d1;
          t[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B
          v[e][i][0] = t[i][0] - t[i][2];
          v[e][i][1] = t[i][1] + t[i][2];
          v[e][i][2] = t[i][2] - t[i][1];
          v[e][i][3] = t[i][1] - t[i][3];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          *reinterpret_cast<unsigned*>(vim + (((4 * i + m) * 8 + (cg >> 1)) * NPV + tl) * 16 + (cg & 1) * 8 + cp * 4) = w_pack(v[0][i][m], v[1][i][m]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const bool want_stats = a.stats != nullptr;
  if (nsteps <= 0) return;  // uniform per workgroup
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  Step cur = step_of(0);
  load_table(cur, cf0);
  fetch(cur);
  __syncthreads();
  commit(cur, raw0, cf0);
  __syncthreads();
  int flush_tile = -1;  // a finished tile whose statistics sit in `red`
#pragma unroll 1
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    const bool more = sidx + 1 < nsteps;  // uniform
    const int par = sidx & 1;
    Step nxt = cur;
    if (more) { nxt = step_of(sidx + 1); fetch(nxt); load_table(nxt, cf0 + (par ^ 1) * 2 * WC); }
    if (want_stats && flush_tile >= 0 && tid < 2 * WC) {  // the tile finished in the previous step (its `red` is complete: barrier below us)
      const int ch = tid >> 1, k = tid & 1;
      a.stats[((size_t)flush_tile * WC + ch) * 2 + k] = red[(0 * WC + ch) * 2 + k] + red[(1 * WC + ch) * 2 + k];
    }
    flush_tile = -1;
    transform(raw0 + par * RAW_BYTES);
    __syncthreads();              // V complete; the table of the next step is visible; `red` has been read
    if (more) commit(nxt, raw0 + (par ^ 1) * RAW_BYTES, cf0 + (par ^ 1) * 2 * WC);
    // ---- 32 MFMAs: A = U (row = output channel), B = V fragment (column = Winograd tile 16 mg + c16, k group q)
    f32x4 acc[16];
    u32x4 bq[2][4];  // fragments of two frequencies (x 2 channel halves), double buffered: the next pair's reads behind this pair's MFMAs
    auto load_pair = [&](int f0, u32x4* b) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b[2 * j] = *reinterpret_cast<const u32x4*>(vim + (((f0 + j) * 8 + q) * NPV + 16 * mg + c16) * 16);
        b[2 * j + 1] = *reinterpret_cast<const u32x4*>(vim + (((f0 + j) * 8 + 4 + q) * NPV + 16 * mg + c16) * 16);
      }
    };
    load_pair(0, bq[0]);
#pragma unroll
    for (int f0 = 0; f0 < 16; f0 += 2) {
      const int cb = (f0 >> 1) & 1;
      if (f0 + 2 < 16) load_pair(f0 + 2, bq[cb ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, U[f0 + j][0]), __builtin_bit_cast(bf16x8, bq[cb][2 * j]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, U[f0 + j][1]), __builtin_bit_cast(bf16x8, bq[cb][2 * j + 1]), acc[f0 + j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- output transform A^T M A: this lane holds Winograd tile tlo = 16 mg + c16, output channels 16 cw + 4 q + r
    const int tlo = 16 * mg + c16, oty = tlo >> 3, otx = tlo & 7;
    const int py = cur.oy + 2 * oty, px = cur.ox + 2 * otx;
    const w_rsrc_t rso = w_make_rsrc(out + (size_t)cur.img * ipix * WC, img_bytes);
    float y[2][2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        s[0][m] = acc[0 + m][r] + acc[4 + m][r] + acc[8 + m][r];
        s[1][m] = acc[4 + m][r] - acc[8 + m][r] - acc[12 + m][r];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        y[i][0][r] = s[i][0] + s[i][1] + s[i][2] + bv[r];
        y[i][1][r] = s[i][1] - s[i][2] - s[i][3] + bv[r];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = py + i < a.Hout && px + j < a.Wout;
        if (want_stats) {
          const float wgt = ok ? 1.f : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = y[i][j][r] * wgt; s1[r] += v; s2[r] += v * y[i][j][r]; }
        }
        const wu32x2 o = {w_pack(y[i][j][0], y[i][j][1]), w_pack(y[i][j][2], y[i][j][3])};
        const unsigned voff = ok ? (unsigned)((((py + i) * a.Wout + px + j) * WC + 16 * cw + 4 * q) * 2) : WSENT64;
        __builtin_amdgcn_raw_buffer_store_b64(o, rso, (int)voff, 0, 0);
      }
    if (want_stats && par == 1) {  // second pass of a tile: per-channel sums over the wave's 16 tiles -> `red`, flushed next step
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = s1[r], t2 = s2[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (c16 == 0) { red[(mg * WC + 16 * cw + 4 * q + r) * 2] = t1; red[(mg * WC + 16 * cw + 4 * q + r) * 2 + 1] = t2; }
        s1[r] = 0.f; s2[r] = 0.f;
      }
      flush_tile = cur.tile;
    }
    __syncthreads();              // V and this step's raw patch are free; the next raw patch and `red` are complete
    cur = nxt;
  }
  if (want_stats && flush_tile >= 0 && tid < 2 * WC) {
    const int ch = tid >> 1, k = tid & 1;
    a.stats[((size_t)flush_tile * WC + ch) * 2 + k] = red[(0 * WC + ch) * 2 + k] + red[(1 * WC + ch) * 2 + k];
  }
}

bool conv64_wino_eligible(int mode, int dtype, const ConvArgs& a) {
  if (mode != MODE_G3S1 || dtype != MIA_BF16) return false;
  if (a.c1 != WC || a.c2 != 0 || a.o1 != WC || a.o2 != 0 || a.npad != WC || a.kpad != WC) return false;
  if (a.cr_y != nullptr || a.acc_out) return false;
  if (a.nl_scale != nullptr && (a.nl_shift == nullptr || !(a.nl_slope >= 0.f && a.nl_slope <= 1.f))) return false;
  if (!a.vec_in || !a.vec_out) return false;
  if ((size_t)a.Hin * a.Win * WC * 2 >= ((size_t)1 << 31)) return false;
  if (a.Hout <= 8) return false;  // statistics of small maps use 8-row tiles
  return true;
}

int conv64_wino_launch(const ConvArgs& a, int reserve, hipStream_t st) {
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int total = a.N * tiles_per_img;
  const int cap = persistent_cus(256, reserve);  // one 512-thread workgroup per CU
  const int nblk = total < cap ? total : cap;
  if (a.nl_scale != nullptr) hipLaunchKernelGGL((conv64_wino_kernel<true>), dim3(nblk), dim3(512), 0, st, a, total, tiles_per_img);
  else hipLaunchKernelGGL((conv64_wino_kernel<false>), dim3(nblk), dim3(512), 0, st, a, total, tiles_per_img);
  return MIA_OK;
}
