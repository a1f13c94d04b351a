// Weight-gradient kernels on MFMA for gfx950 (NHWC activations).
//
//   dW[tap][n][k] = sum over output pixels p of  dy[p][n] * x[p*stride + tap - pad][k]
//
// (reference: autograd of nn.Conv2d at src/models/unet/blocks.py:83-90 and of
// nn.ConvTranspose2d at unet.py:142).  The contraction index is the PIXEL, which is the slow
// dimension of both NHWC operands, so:
//   * bf16: tiles are staged to LDS in their natural [pixel][channel] order (XOR-swizzled 8-row x
//     32-column subtiles) and both MFMA operands are fetched with ds_read_b64_tr_b16 (hardware
//     transposing read) -> v_mfma_f32_16x16x32_bf16, K = 32 pixels per instruction.
//   * fp32: v_mfma_f32_16x16x4_f32 takes one scalar per lane; [pixel][channel] LDS rows padded to
//     80 dwords make the b32 fragment reads conflict free.
// A workgroup (4 waves) owns a 64(n) x 64(k) block of dW for ALL taps; wave w owns k-tile w and keeps
// taps x 4 accumulators in registers while it walks its share of the pixel tiles (split-K over
// gridDim.y).  Partials go to per-split slabs; mia_wgrad_reduce sums the slabs in a fixed order
// (bitwise reproducible, no float atomics) straight into the parameter's native OIHW / IOHW layout.
//   MODE_W3S1: 3x3 stride 1 pad 1;  MODE_W3S2: 3x3 stride 2 pad 1;  MODE_W2S2: 2x2 stride 2 pad 0
//   (MODE_W2S2 is ConvTranspose2d's wgrad with x := grad_output (fine grid), dy := input (coarse)).
#include "common.h"
#include "options.h"
#include <stdlib.h>
#include <type_traits>

enum { MODE_W3S1 = 0, MODE_W3S2 = 1, MODE_W2S2 = 2 };

struct WgArgs {
  const void* x1; const void* x2; int c1; int c2;
  const void* dy; int cdy;
  float* slabs;
  int N, Hx, Wx, Hy, Wy;
  int npad, kpad, ksplit;
  int tiles_x, tiles_y;
  int vec_x, vec_dy;
  int opt;  // bit 0: table-driven staging of interior tiles (wgrad_bf16_2wg_kernel)
  // normalise-on-load (mia_conv_wgrad_nl): x1 is the RAW conv output y of the producing PlainBlock; the kernel stages
  // lrelu(nl_scale[n][k] * y + nl_shift[n][k]) (zero outside the image); nullptr = x1 is an ordinary activation
  const float* nl_scale = nullptr; const float* nl_shift = nullptr; float nl_slope = 0.f;
  // fp32 split mode (common.h SplitF16): max |x| of x1 / x2 / dy as fp32 bit patterns in device memory
  const unsigned* amax_x1 = nullptr; const unsigned* amax_x2 = nullptr; const unsigned* amax_dy = nullptr;
};

template <int MODE> struct WGeo {
  static constexpr int KS = MODE == MODE_W2S2 ? 2 : 3;
  static constexpr int S = MODE == MODE_W3S1 ? 1 : 2;
  static constexpr int PAD = MODE == MODE_W2S2 ? 0 : 1;
  static constexpr int TAPS = KS * KS;
};

// ---------------------------------------------------------------- bf16 (tr16 reads)
__device__ __forceinline__ int swz_off(int row, int ch) {
  // byte offset of 16-byte chunk `ch` (0..7) of pixel-row `row` in a [rows][64 x bf16] tile, stored as
  // 8-row x 32-column subtiles of 512 B with the chunk index XOR-swizzled by (row>>2)&3
  return 512 * ((row >> 3) * 2 + (ch >> 2)) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr_read(const unsigned char* base, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
}

// transposing read at an absolute LDS byte address (the workgroup's LDS base folded into the lane-constant part once, instead of
// a v_add per read)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ s16x4 tr_read_at(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_u8*)(size_t)addr);
}

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgArgs a) {
  using G = WGeo<MODE>;
  constexpr int KS = G::KS, S = G::S, PAD = G::PAD, TAPS = G::TAPS;
  constexpr int TH = (S == 1) ? 8 : 4;
  constexpr int XH = (TH - 1) * S + KS, XW = 15 * S + KS;
  constexpr int XROWS = ((XH * XW + 7) / 8) * 8;
  constexpr int X_BYTES = XROWS * 128, D_BYTES = TH * 16 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[X_BYTES + D_BYTES];
  unsigned char* xs = smem;
  unsigned char* ds = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int nkb = a.kpad / 64;
  const int kblk = blockIdx.x % nkb, nblk = blockIdx.x / nkb;
  const int n0 = nblk * 64, k0 = kblk * 64;
  const int kin = a.c1 + a.c2;
  const bf16_t* x1 = static_cast<const bf16_t*>(a.x1);
  const bf16_t* x2 = static_cast<const bf16_t*>(a.x2);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_active = (k0 + wave * 16) < kin;

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  for (int tile = blockIdx.y; tile < ntiles; tile += a.ksplit) {
    int tt = tile;
    const int tx = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty = tt % a.tiles_y; tt /= a.tiles_y;
    const int img = tt;
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    __syncthreads();  // previous tile's reads done
    for (int u = tid; u < XH * XW * 8; u += 256) {
      const int ch = u & 7, pix = u >> 3;
      const int iy = pix / XW, ix = pix - iy * XW;
      const int gy = iy0 + iy, gx = ix0 + ix, c = k0 + ch * 8;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (gy >= 0 && gy < a.Hx && gx >= 0 && gx < a.Wx && c < kin) {
        const size_t p = ((size_t)img * a.Hx + gy) * a.Wx + gx;
        if (a.vec_x) {
          const bf16_t* src = (c < a.c1) ? x1 + p * a.c1 + c : x2 + p * a.c2 + (c - a.c1);
          v = *reinterpret_cast<const u32x4*>(src);
        } else {
          alignas(16) bf16_t tmp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ce = c + e;
            tmp[e] = ce < a.c1 ? x1[p * a.c1 + ce] : (ce < kin ? x2[p * a.c2 + (ce - a.c1)] : (bf16_t)0);
          }
          v = *reinterpret_cast<const u32x4*>(tmp);
        }
      }
      *reinterpret_cast<u32x4*>(xs + swz_off(pix, ch)) = v;
    }
    for (int u = tid; u < TH * 16 * 8; u += 256) {
      const int ch = u & 7, pix = u >> 3;
      const int y = pix >> 4, xx = pix & 15;
      const int gy = oy0 + y, gx = ox0 + xx, c = n0 + ch * 8;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (gy < a.Hy && gx < a.Wy && c < a.cdy) {
        const size_t p = ((size_t)img * a.Hy + gy) * a.Wy + gx;
        if (a.vec_dy) {
          v = *reinterpret_cast<const u32x4*>(dy + p * a.cdy + c);
        } else {
          alignas(16) bf16_t tmp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp[e] = (c + e < a.cdy) ? dy[p * a.cdy + c + e] : (bf16_t)0;
          v = *reinterpret_cast<const u32x4*>(tmp);
        }
      }
      *reinterpret_cast<u32x4*>(ds + swz_off(pix, ch)) = v;
    }
    __syncthreads();
    if (wave_active) {
#pragma unroll
      for (int kb = 0; kb < TH / 2; ++kb) {
        // lane group `grp` covers k = 8*grp .. 8*grp+7 of this 32-pixel block; two 4-row tr reads each
        const int yy = 2 * kb + (grp >> 1), xb = 8 * (grp & 1) + qp;
        u32x4 af[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 2 * c + (pp >> 1);
          const int r0 = yy * 16 + xb;
          const s16x4 lo = tr_read(ds, swz_off(r0, ch) + 8 * (pp & 1));
          const s16x4 hi = tr_read(ds, swz_off(r0 + 4, ch) + 8 * (pp & 1));
          af[c] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const int ch = 2 * wave + (pp >> 1);
            const int r0 = (yy * S + kh) * XW + xb * S + kw;
            const s16x4 lo = tr_read(xs, swz_off(r0, ch) + 8 * (pp & 1));
            const s16x4 hi = tr_read(xs, swz_off(r0 + 4 * S, ch) + 8 * (pp & 1));
            const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int c = 0; c < 4; ++c)
              acc[kh * KS + kw][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]), b,
                                                                               acc[kh * KS + kw][c], 0, 0, 0);
          }
        }
      }
    }
  }
  // slab[z][tap][n][k]: C rows = n (A side), cols = k (B side)
  float* slab = a.slabs + (size_t)blockIdx.y * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + c * 16 + 4 * grp + r, k = k0 + wave * 16 + i16;
        slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16 fast path
// Same tiling / LDS image / MFMA schedule as wgrad_bf16_kernel, but (a) the NEXT pixel tile is fetched
// global -> VGPR while the current tile's MFMAs run (the generic kernel idles the matrix pipe for the
// whole staging phase: rocprofv3 SQ_WAIT_ANY = 63 % of wave cycles), and (b) every access is a raw
// buffer load against a per-image descriptor, so borders are out-of-range offsets -> zeros, no branches.
// Contract: channel counts multiples of the 16-byte unit, a two-source input split on a 64-channel boundary,
// 16-byte aligned, per-image tensors < 2 GiB (channel tails are zero-filled like the image border).
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
#define WSENT 0xFFFFFFF0u
__device__ __forceinline__ wrsrc_t wmake_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// W8 (round 5): 512 threads -- eight waves = 4 input-channel tiles x 2 halves of the block's output channels, 72 accumulator registers per
// wave.  The 256-thread form compiles to 436-464 registers (accumulators parked in AGPRs): ONE wave per SIMD, staging and MFMAs never
// overlapping (0.37 PFLOP/s on cfg5's 96 -> 192 stride-2 and 192 -> 96 transposed gradients, the only launches that still use it).
template <int MODE, int TH, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void wgrad_bf16_fast_kernel(const WgArgs a) {
  using G = WGeo<MODE>;
  constexpr int KS = G::KS, S = G::S, PAD = G::PAD, TAPS = G::TAPS;
  constexpr int XH = (TH - 1) * S + KS, XW = 15 * S + KS;
  constexpr int PPI = W8 ? 64 : 32;  // pixels x 8 chunks per staging iteration
  constexpr int X_IT = (XH * XW + PPI - 1) / PPI, D_IT = (TH * 16 + PPI - 1) / PPI;
  constexpr int X_BYTES = X_IT * PPI * 128, D_BYTES = D_IT * PPI * 128;
  constexpr int NC = W8 ? 2 : 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[X_BYTES + D_BYTES];
  unsigned char* xs = smem;
  unsigned char* ds = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int ch8 = tid & 7, p8 = tid >> 3;
  const int kq = W8 ? (wave & 3) : wave, nh = W8 ? 2 * (wave >> 2) : 0;  // input-channel tile; first output-channel tile of this wave
  // 64-channel input blocks are cut per SOURCE (ceil(c1/64) + ceil(c2/64) of them), so a block never straddles the
  // two tensors of a concatenated input whatever c1 is; a source's last block may be partial (lanes beyond cs read
  // zeros and do not store)
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {  // XCD-aware order (1-D grid): the column blocks of one split index share their tiles -> one XCD, one L2
    const int ncol = nkb * (a.npad / 64), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 64, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  // tile-invariant unit geometry
  int x_iy[X_IT], x_ix[X_IT];
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int pix = p8 + PPI * i;
    x_iy[i] = pix < XH * XW ? pix / XW : -100000;
    x_ix[i] = pix - (pix / XW) * XW;
  }
  const int lds_x0 = swz_off(p8, ch8), lds_d0 = swz_off(p8, ch8);  // + 128 PPI per iteration (PPI rows)

  f32x4 acc[TAPS][NC];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 px[X_IT], pd[D_IT];
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  auto fetch = [&](int tile) {
    int tt = tile;
    const int tx = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty = tt % a.tiles_y; tt /= a.tiles_y;
    const int img = tt;
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const wrsrc_t rx = wmake_rsrc(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
    const wrsrc_t rd = wmake_rsrc(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int gy = iy0 + x_iy[i], gx = ix0 + x_ix[i];
      const bool ok = gy >= 0 && gy < a.Hx && gx >= 0 && gx < a.Wx;
      const unsigned voff = (ok && kloc + ch8 * 8 < cs) ? (unsigned)(((gy * a.Wx + gx) * cs + kloc + ch8 * 8) * 2) : WSENT;
      px[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int pix = p8 + PPI * i;
      const int gy = oy0 + (pix >> 4), gx = ox0 + (pix & 15);
      const bool ok = pix < TH * 16 && gy < a.Hy && gx < a.Wy;
      const unsigned voff = (ok && n0 + ch8 * 8 < a.cdy) ? (unsigned)(((gy * a.Wy + gx) * a.cdy + n0 + ch8 * 8) * 2) : WSENT;
      pd[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)voff, 0, 0);
    }
  };

  int tile = by;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += a.ksplit) {
    __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < X_IT; ++i) *reinterpret_cast<u32x4*>(xs + lds_x0 + 128 * PPI * i) = px[i];
#pragma unroll
    for (int i = 0; i < D_IT; ++i) *reinterpret_cast<u32x4*>(ds + lds_d0 + 128 * PPI * i) = pd[i];
    __syncthreads();
    if (tile + a.ksplit < ntiles) fetch(tile + a.ksplit);
#pragma unroll(TH <= 8 ? TH / 2 : 2)
    for (int kb = 0; kb < TH / 2; ++kb) {
      const int yy = 2 * kb + (grp >> 1), xb = 8 * (grp & 1) + qp;
      u32x4 af[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = 2 * (nh + c) + (pp >> 1);
        const int r0 = yy * 16 + xb;
        const s16x4 lo = tr_read(ds, swz_off(r0, ch) + 8 * (pp & 1));
        const s16x4 hi = tr_read(ds, swz_off(r0 + 4, ch) + 8 * (pp & 1));
        af[c] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const int ch = 2 * kq + (pp >> 1);
          const int r0 = (yy * S + kh) * XW + xb * S + kw;
          const s16x4 lo = tr_read(xs, swz_off(r0, ch) + 8 * (pp & 1));
          const s16x4 hi = tr_read(xs, swz_off(r0 + 4 * S, ch) + 8 * (pp & 1));
          const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int c = 0; c < NC; ++c)
            acc[kh * KS + kw][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]), b,
                                                                             acc[kh * KS + kw][c], 0, 0, 0);
        }
      }
    }
  }
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (nh + c) * 16 + 4 * grp + r, k = k0 + kq * 16 + i16;
        if (kloc + kq * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16, two workgroups per CU
// wgrad_bf16_fast_kernel ends up at 464 registers = ONE workgroup of four waves per CU, one wave per SIMD.  Measured on the
// 64-channel launch (rocprofv3 PMC, profiles/r02_pmc_wgrad64.csv): matrix pipe 41 % busy, 3.3 us per 128-pixel tile against
// 1.2 us of MFMA issue -- with a single tile (39 KB) of loads in flight per CU the walk waits on memory latency, and nothing
// covers the staging writes and the two barriers of a tile.  This kernel keeps the same 64(n) x 64(k) x all-taps block per
// workgroup and the same swizzled LDS image, but fits in 256 registers so that TWO workgroups share a CU (two tiles in
// flight, one workgroup's staging / barriers behind the other's MFMAs):
//   * the row-block loop stays ROLLED and its fragment reads are "lane-constant base + immediate": the x image lives at a
//     pitch of 32 pixels (18 used), so a tap / row step is a multiple of 32 LDS rows = 4096 bytes and leaves the swizzle
//     bits alone (unrolled, hipcc hoists one computed address per (row block, tap) and the fragment reads of all row
//     blocks: 464 registers);
//   * staging addresses are branch free and recomputed per tile (a hoisted table is spilled and reloaded behind vmcnt(0)),
//     the tile walk carries (image, row, column) digits instead of dividing.
// Stride-1 3x3 only (the 32-pixel pitch); the stride-2 / transposed shapes stay on wgrad_bf16_fast_kernel.
// Diagnostic build only (-DCONV64_STAMPS, tools/conv64_stamps.py wgrad): per-wave cycle sums of a tile's phases.
#ifdef CONV64_STAMPS
__device__ unsigned long long wgrad_dbg[512 * 4 * 8];
#define WSTAMP(var)                                                                  \
  do {                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                               \
  } while (0)
extern "C" int mia_wgrad_debug_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wgrad_dbg), sizeof(wgrad_dbg));
}
#else
#define WSTAMP(var) do { } while (0)
#endif

// NL = normalise-on-load of the x operand (see conv64.hip: the consumer-side half of the fused PlainBlock): x1 holds the
// producer's raw conv output and commit() turns each staged 16-byte unit into bf16(lrelu(scale * y + shift)) -- bit for bit the
// activation mia_norm_act_fwd would have written -- with the coefficients of the tile's image in a 512-byte LDS table
// (threads 0..127 fetch one entry each with the tile), and halo units outside the image forced back to zero.
template <int TH, bool NL = false>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_2wg_kernel(const WgArgs a) {
  constexpr int KS = 3, PAD = 1, TAPS = 9;
  constexpr int XH = TH - 1 + KS, XW = 15 + KS, XP = 32;
  constexpr int X_IT = (XH * XW + 31) / 32, D_IT = TH * 16 / 32;  // 32 pixels x 8 chunks per staging iteration
  constexpr int X_BYTES = XH * XP * 128, D_BYTES = TH * 16 * 128;
  static_assert(X_IT == 6 && D_IT == 4, "the staging table below is laid out for TH = 8");
  __shared__ __attribute__((aligned(16))) unsigned char smem[X_BYTES + D_BYTES + 4 * 256 * 16 + (NL ? 512 : 0)];
  u32x4* tab = reinterpret_cast<u32x4*>(smem + X_BYTES + D_BYTES);  // [4][256]: per-thread staging constants, see below
  float* cft = reinterpret_cast<float*>(smem + X_BYTES + D_BYTES + 4 * 256 * 16);  // NL: [0, 64) scale, [64, 128) shift (this k block, committed tile's image)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int ch8 = tid & 7, p8 = tid >> 3;
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {  // XCD-aware order (1-D grid): the column blocks of one split index share their tiles -> one XCD, one L2
    const int ncol = nkb * (a.npad / 64), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 64, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;
  const bool x_chan_ok = kloc + ch8 * 8 < cs, d_chan_ok = n0 + ch8 * 8 < a.cdy;

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }

  // Per-thread staging constants live in LDS, not in registers (there are none to spare) and not in VALU work per tile
  // (measured with the stamps build: ~200 address instructions per tile, issued at half rate beside the other workgroup's
  // MFMAs, made the fetch phase 1800 cycles of an 8000-cycle tile): for a tile whose halo lies inside the image the ten
  // global offsets are "tile origin (folded into the buffer descriptor) + constant", and the six LDS offsets are constant.
  //   tab[0] = x offsets 0..3, tab[1] = x offsets 4..5 | dy offsets 0..1, tab[2] = dy offsets 2..3 | LDS offsets 0..1,
  //   tab[3] = LDS offsets 2..5
  {
    unsigned xo[X_IT], xl[X_IT], dofs[D_IT];
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int pix = p8 + 32 * i, iy = pix / XW, ix = pix - iy * XW;
      xo[i] = (pix < XH * XW && x_chan_ok) ? (unsigned)(((iy * a.Wx + ix) * cs + kloc + ch8 * 8) * 2) : WSENT;
      xl[i] = (unsigned)swz_off(iy * XP + ix, ch8);
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int pix = p8 + 32 * i;
      dofs[i] = d_chan_ok ? (unsigned)((((pix >> 4) * a.Wy + (pix & 15)) * a.cdy + n0 + ch8 * 8) * 2) : WSENT;
    }
    tab[tid] = u32x4{xo[0], xo[1], xo[2], xo[3]};
    tab[256 + tid] = u32x4{xo[4], xo[5], dofs[0], dofs[1]};
    tab[512 + tid] = u32x4{dofs[2], dofs[3], xl[0], xl[1]};
    tab[768 + tid] = u32x4{xl[2], xl[3], xl[4], xl[5]};
  }

  u32x4 px[X_IT], pd[D_IT];
  float cpf = 0.f;  // NL: this thread's entry of the fetched tile's coefficient table
  auto fetch = [&](int img, int ty, int tx) {
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 - PAD, ix0 = ox0 - PAD;
    if constexpr (NL) {
      if (wave < 2) {  // wave 0 fetches the 64 scales, wave 1 the 64 shifts: the array pointer stays scalar
        const float* cp = wave == 0 ? a.nl_scale : a.nl_shift;
        int lv = lane;
        asm volatile("" : "+v"(lv));  // recomputed per tile: a hoisted 64-bit lane address would be spilled around the tile loop
        const int ch = kloc + lv;
        cpf = ch < cs ? cp[(unsigned)(img * cs + ch)] : 0.f;  // scalar base + 32-bit lane offset
      }
    }
    if ((a.opt & 1) && iy0 >= 0 && ix0 >= 0 && iy0 + XH <= a.Hx && ix0 + XW <= a.Wx && oy0 + TH <= a.Hy && ox0 + 16 <= a.Wy) {
      const size_t xorg = (size_t)iy0 * a.Wx + ix0, dorg = (size_t)oy0 * a.Wy + ox0;
      const wrsrc_t rx = wmake_rsrc(xsrc + ((size_t)img * xpix + xorg) * cs, (unsigned)((xpix - xorg) * cs * 2));
      const wrsrc_t rd = wmake_rsrc(dy + ((size_t)img * ypix + dorg) * a.cdy, (unsigned)((ypix - dorg) * a.cdy * 2));
      const u32x4 t0 = tab[tid], t1 = tab[256 + tid], t2 = tab[512 + tid];
      px[0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t0.x, 0, 0);
      px[1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t0.y, 0, 0);
      px[2] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t0.z, 0, 0);
      px[3] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t0.w, 0, 0);
      px[4] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t1.x, 0, 0);
      px[5] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)t1.y, 0, 0);
      pd[0] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)t1.z, 0, 0);
      pd[1] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)t1.w, 0, 0);
      pd[2] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)t2.x, 0, 0);
      pd[3] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)t2.y, 0, 0);
      return;
    }
    const wrsrc_t rx = wmake_rsrc(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
    const wrsrc_t rd = wmake_rsrc(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
    int p8v = p8;
    asm volatile("" : "+v"(p8v));
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int pix = p8v + 32 * i, iy = pix / XW, ix = pix - iy * XW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const int okm = -(int)(((unsigned)gy < (unsigned)a.Hx) & ((unsigned)gx < (unsigned)a.Wx) & (pix < XH * XW) & x_chan_ok);
      const unsigned off = (unsigned)(((gy * a.Wx + gx) * cs + kloc + ch8 * 8) * 2);
      px[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((off & (unsigned)okm) | (WSENT & ~(unsigned)okm)), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int pix = p8v + 32 * i;
      const int gy = oy0 + (pix >> 4), gx = ox0 + (pix & 15);
      const int okm = -(int)((gy < a.Hy) & (gx < a.Wy) & d_chan_ok);
      const unsigned off = (unsigned)(((gy * a.Wy + gx) * a.cdy + n0 + ch8 * 8) * 2);
      pd[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)((off & (unsigned)okm) | (WSENT & ~(unsigned)okm)), 0, 0);
    }
  };
  auto commit = [&](int ty, int tx) {
    int p8v = p8;
    asm volatile("" : "+v"(p8v));
    if constexpr (NL) {
      const f32x4* cf4 = reinterpret_cast<const f32x4*>(cft);
      const int iy0 = ty * TH - PAD, ix0 = tx * 16 - PAD;
      const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + XH <= a.Hx && ix0 + XW <= a.Wx;  // uniform
      typedef float nl_f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 nl_bf16x2 __attribute__((ext_vector_type(2)));
      const nl_f32x2 sl2 = {a.nl_slope, a.nl_slope};
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // dwords 0,1 then 2,3 of every unit: 8 coefficient registers live at a time
        const f32x4 sc = cf4[2 * ch8 + hf], sh = cf4[16 + 2 * ch8 + hf];
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {  // packed fp32 math: one issue slot per channel pair (v_pk_fma_f32, v_pk_mul_f32)
            const unsigned w = px[i][2 * hf + d];
            const nl_f32x2 x = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
            const nl_f32x2 v = __builtin_elementwise_fma(nl_f32x2{sc[2 * d], sc[2 * d + 1]}, x, nl_f32x2{sh[2 * d], sh[2 * d + 1]});
            const nl_f32x2 m = v * sl2;
            // (channels past `cs` carry scale = shift = 0 in the table: lrelu(0) = 0)
            px[i][2 * hf + d] = __builtin_bit_cast(unsigned, __builtin_convertvector(nl_f32x2{__builtin_fmaxf(v[0], m[0]), __builtin_fmaxf(v[1], m[1])}, nl_bf16x2));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!interior) {  // border tile: halo units outside the image go back to zero
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
          const int pix = p8v + 32 * i, iy = pix / XW, ix = pix - iy * XW;
          const unsigned keep = 0u - (unsigned)(((unsigned)(iy0 + iy) < (unsigned)a.Hx) & ((unsigned)(ix0 + ix) < (unsigned)a.Wx));
#pragma unroll
          for (int d = 0; d < 4; ++d) px[i][d] &= keep;
        }
      }
    }
    const u32x4 t2 = tab[512 + tid], t3 = tab[768 + tid];
    *reinterpret_cast<u32x4*>(smem + t2.z) = px[0];
    *reinterpret_cast<u32x4*>(smem + t2.w) = px[1];
    *reinterpret_cast<u32x4*>(smem + t3.x) = px[2];
    *reinterpret_cast<u32x4*>(smem + t3.y) = px[3];
    *reinterpret_cast<u32x4*>(smem + t3.z) = px[4];
    if (p8v + 32 * 5 < XH * XW) *reinterpret_cast<u32x4*>(smem + t3.w) = px[5];
    const int d0 = swz_off(p8v, ch8);  // + 4096 per iteration (32 rows)
#pragma unroll
    for (int i = 0; i < D_IT; ++i) *reinterpret_cast<u32x4*>(smem + X_BYTES + d0 + 4096 * i) = pd[i];
  };

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant read bases (bytes).  Lane group `grp` covers pixels 8*(grp&1) .. +7 of output row 2*kb + (grp>>1); a
  // transposing read fetches 4 consecutive pixel rows, the pair (lo, hi) = rows r0 .. r0+3 and r0+4 .. r0+7.
  const int g1 = grp >> 1, xb = 8 * (grp & 1) + qp, sub = 8 * (pp & 1);
  const int lds0 = (int)(unsigned)(size_t)(lds_u8*)smem;  // absolute LDS address of the tile image
  int dbase[4][2], xbase[KS][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dbase[c][0] = lds0 + X_BYTES + swz_off(g1 * 16 + xb, 2 * c + (pp >> 1)) + sub;
    dbase[c][1] = lds0 + X_BYTES + swz_off(g1 * 16 + xb + 4, 2 * c + (pp >> 1)) + sub;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xbase[kw][0] = lds0 + swz_off(g1 * XP + xb + kw, 2 * wave + (pp >> 1)) + sub;
    xbase[kw][1] = lds0 + swz_off(g1 * XP + xb + kw + 4, 2 * wave + (pp >> 1)) + sub;
  }

  if (tile < ntiles) fetch(t_img, t_ty, t_tx);
#ifdef CONV64_STAMPS
  unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, a_b1 = 0, a_c = 0, a_b2 = 0, a_f = 0, a_m = 0, a_n = 0;
#endif
  for (; tile < ntiles; tile += a.ksplit) {
    WSTAMP(w0);
    if constexpr (NL) {  // the table is read in commit() only, i.e. between the two barriers below
      if (tid < 128) cft[tid] = cpf;
    }
    __syncthreads();  // previous tile's fragment reads are done
    WSTAMP(w1);
    commit(t_ty, t_tx);
    WSTAMP(w2);
    __syncthreads();
    WSTAMP(w3);
    if (tile + a.ksplit < ntiles) {
      t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
      t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
      t_img += d_img;
      fetch(t_img, t_ty, t_tx);
    }
    WSTAMP(w4);
#pragma unroll 1
    for (int kb = 0; kb < TH / 2; ++kb) {
      const int koff = 4096 * kb;  // 32 dy rows per row block; the x image advances two 32-pixel rows
      u32x4 af[4], bf[2];
      auto load_b = [&](int t) -> u32x4 {
        const int kh = t / KS, kw = t % KS;
        const s16x4 lo = tr_read_at((unsigned)(xbase[kw][0] + 2 * koff + 4096 * kh));
        const s16x4 hi = tr_read_at((unsigned)(xbase[kw][1] + 2 * koff + 4096 * kh));
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const s16x4 lo = tr_read_at((unsigned)(dbase[c][0] + koff));
        const s16x4 hi = tr_read_at((unsigned)(dbase[c][1] + koff));
        af[c] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      bf[0] = load_b(0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t + 1 < TAPS) bf[(t + 1) & 1] = load_b(t + 1);  // next tap's fragment ahead of this tap's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]), __builtin_bit_cast(bf16x8, bf[t & 1]),
                                                              acc[t][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WSTAMP(w5);
#ifdef CONV64_STAMPS
    a_b1 += w1 - w0; a_c += w2 - w1; a_b2 += w3 - w2; a_f += w4 - w3; a_m += w5 - w4; a_n += 1;
#endif
  }
#ifdef CONV64_STAMPS
  if (lane == 0 && bx == 0 && by < 512) {
    unsigned long long* d = wgrad_dbg + ((size_t)by * 4 + wave) * 8;
    d[0] = a_b1; d[1] = a_c; d[2] = a_b2; d[3] = a_f; d[4] = a_m; d[5] = a_n;
  }
#endif
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + c * 16 + 4 * grp + r, k = k0 + wave * 16 + i16;
        if (kloc + wave * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16, LDS-DMA ring (stride-1 3x3)
// Same block per workgroup (64 n x 64 k x 9 taps), same swizzled LDS image and fragment reads as wgrad_bf16_2wg_kernel, but
// the tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... offen lds`: no staging registers, no ds_write pass, no second
// barrier) into a ring of THREE tile images, so a tile's loads have two tile times to land:
//   * a tile is 4 output rows x 16 pixels: x image [6 rows][24 pixels (18 used)][64 ch] = 18 KB, dy image [64 px][64 ch] =
//     8 KB; 3 x 26 KB = 78 KB per workgroup, two workgroups per CU (156 of 160 KB);
//   * one DMA instruction writes 1 KB = one 8-pixel x 128-byte row block of the image, lane L at byte 16 L; the chunk
//     swizzle is applied on the SOURCE side (lane L fetches chunk (L&3) ^ swizzle(row)), out-of-image / out-of-channel lanes
//     point past the descriptor and are zero filled.  26 pieces per tile, dealt round-robin to the 4 waves;
//   * per tile: issue tile t+2 -> MFMAs of tile t -> s_waitcnt vmcnt(own pieces of t+2) [= own pieces of t+1 landed] ->
//     s_barrier [everyone's pieces of t+1 landed, everyone done reading t].  The DMA is issued and counted in inline asm
//     (hipcc would drain it with vmcnt(0) at every barrier / LDS read it can see);
//   * the 40 staging registers of the 2wg kernel pay for a second set of dy fragments and a three-deep x fragment ring, so
//     the fragment reads run two taps ahead of their MFMAs.
typedef int wi32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wi32x4 wmake_rsrc_i(const void* p, unsigned bytes) {
  const unsigned long long addr = (unsigned long long)p;
  wi32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)addr);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(addr >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16(wi32x4 rsrc, unsigned voff, unsigned lds_dst) {
  // M0 = wave-uniform LDS byte address of the 1 KB piece.  M0 is written and read inside this one statement and not restored:
  // hipcc uses M0 for nothing else in these kernels (checked in the ISA: no other reference to m0), and a save / restore
  // pair per piece is two more scalar instructions in the phase that has to hide behind the other workgroup's MFMAs.
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}

// NARROW = the launch has blocks with fewer than 64 valid channels (c or cdy not a multiple of 64: cfg5's 96-channel level): those
// blocks skip their empty 16-channel tiles and re-deal the waves (below); full blocks of such a launch run their four n tiles as two
// passes of two over the staged tile.  NARROW = false is the unchanged round-2 body.
template <bool NARROW>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_dma_kernel(const WgArgs a) {
  constexpr int KS = 3, TAPS = 9, TH = 4;
  constexpr int XH = TH + 2, XROW = 3072;  // 24 pixels x 128 B per image row of the x tile
  constexpr int X_BYTES = XH * XROW, D_BYTES = TH * 16 * 128, STAGE = X_BYTES + D_BYTES, NSTAGE = 3;
  constexpr int XPIECES = XH * 3, PIECES = XPIECES + TH * 2;  // 18 + 8
  constexpr int MAXOWN = (PIECES + 3) / 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {  // XCD-aware order (1-D grid): the column blocks of one split index share their tiles -> one XCD, one L2
    const int ncol = nkb * (a.npad / 64), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 64, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  // DMA lane constants: lane L of a piece is 16-byte chunk (L&3) of half (L>>5) of pixel row r = (L>>2)&7 of the 8-row block;
  // the source chunk is un-swizzled by the block's parity p (rows 8 blk + r: (row>>2)&3 = (2 p + (r>>2)) & 3)
  const int dr = (lane >> 2) & 7;
  // (plain scalars, not arrays: a wave-uniform but run-time index sends an array to scratch, whose reload waits vmcnt(0))
  const int ch8_0 = 4 * (lane >> 5) + ((lane & 3) ^ ((dr >> 2) & 3)), ch8_1 = 4 * (lane >> 5) + ((lane & 3) ^ ((2 + (dr >> 2)) & 3));
  const unsigned xlane0 = (unsigned)((dr * cs + kloc + ch8_0 * 8) * 2), xlane1 = (unsigned)((dr * cs + kloc + ch8_1 * 8) * 2);
  const unsigned dlane0 = (unsigned)((dr * a.cdy + n0 + ch8_0 * 8) * 2), dlane1 = (unsigned)((dr * a.cdy + n0 + ch8_1 * 8) * 2);
  const bool xok0 = kloc + ch8_0 * 8 < cs, xok1 = kloc + ch8_1 * 8 < cs;
  const bool dok0 = n0 + ch8_0 * 8 < a.cdy, dok1 = n0 + ch8_1 * 8 < a.cdy;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  // Narrow blocks (round 4; cfg5's 96-channel level: 96 = 64 + 32 in both dimensions): a block with <= 32 valid input channels
  // has only two 16-channel k tiles, so instead of leaving waves 2, 3 idle the four waves are (k tile = wave & 1) x (half of the
  // n tiles = wave >> 1); a block with <= 32 valid output channels simply skips its empty n tiles.  Uniform per workgroup:
  // this wave computes k tile `ktile` against the n tiles cbeg .. cbeg + ccnt - 1 (ccnt in 0..4).
  const int ntl = ((a.cdy - n0 < 64 ? a.cdy - n0 : 64) + 15) >> 4, ktl = ((cs - kloc < 64 ? cs - kloc : 64) + 15) >> 4;
  int ktile = wave, cbeg = 0, ccnt = 4;
  if constexpr (NARROW) {
    if (ktl > 2) { ktile = wave; cbeg = 0; ccnt = ntl; }
    else if (ntl > 2) { ktile = wave & 1; cbeg = 2 * (wave >> 1); ccnt = ntl - cbeg < 2 ? ntl - cbeg : 2; }
    else { ktile = wave & 1; cbeg = wave >> 1; ccnt = cbeg < ntl ? 1 : 0; }
    if (ktile >= ktl) ccnt = 0;
    if (ccnt < 0) ccnt = 0;
  }

  auto issue = [&](int img, int ty, int tx, unsigned stage_base) {
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const wi32x4 rx = wmake_rsrc_i(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
    const wi32x4 rd = wmake_rsrc_i(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
#pragma unroll
    for (int j = 0; j < MAXOWN; ++j) {
      const int pc = wave + 4 * j;  // wave-uniform piece index
      if (pc < XPIECES) {
        const int iy = pc / 3, xb = pc - 3 * iy;
        const int gy = iy0 + iy, gx = ix0 + 8 * xb + dr;
        const bool ok = ((unsigned)gy < (unsigned)a.Hx) & ((unsigned)gx < (unsigned)a.Wx) & (8 * xb + dr < 18) & ((xb & 1) ? xok1 : xok0);
        const unsigned off = (unsigned)((gy * a.Wx + ix0 + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0);
        lds_dma16(rx, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(stage_base + iy * XROW + xb * 1024));
      } else if (pc < PIECES) {
        const int q = pc - XPIECES;  // 8-pixel block of the dy tile: output row q>>1, pixels 8 (q&1) ..
        const int gy = oy0 + (q >> 1), gx = ox0 + 8 * (q & 1) + dr;
        const bool ok = (gy < a.Hy) & (gx < a.Wy) & ((q & 1) ? dok1 : dok0);
        const unsigned off = (unsigned)((gy * a.Wy + ox0 + 8 * (q & 1)) * a.cdy * 2) + ((q & 1) ? dlane1 : dlane0);
        lds_dma16(rd, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(stage_base + X_BYTES + q * 1024));
      }
    }
  };
  // Tiles whose 18 columns lie inside the image (all but the first / last tile of a row): the lane part of every piece's
  // offset is a constant (kept in registers, padding / channel-tail lanes already pointing out of range), the tile origin
  // goes into the descriptor base and a piece's rows are valid or not as a whole.  Measured with the stamps build: the
  // general issue() above costs ~1700 cycles per tile and wave (as long as the tile's MFMAs), this one a fraction.
  unsigned voffc[MAXOWN];
#pragma unroll
  for (int j = 0; j < MAXOWN; ++j) {
    const int pc = wave + 4 * j;
    if (pc < XPIECES) {
      const int iy = pc / 3, xb = pc - 3 * iy, ix = 8 * xb + dr;
      const bool ok = (ix < 18) & ((xb & 1) ? xok1 : xok0);
      voffc[j] = ok ? (unsigned)((iy * a.Wx + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0) : WSENT;
    } else {
      const int q = pc - XPIECES;
      const bool ok = (pc < PIECES) & ((q & 1) ? dok1 : dok0);
      voffc[j] = ok ? (unsigned)(((q >> 1) * a.Wy + 8 * (q & 1)) * a.cdy * 2) + ((q & 1) ? dlane1 : dlane0) : WSENT;
    }
  }
  auto issue_fast = [&](int img, int ty, int tx, unsigned stage_base) {
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    // descriptor bases at the tile origin (row iy0 may be -1: its pieces are dropped below, nothing is read through it)
    const long long xorg = ((long long)(img * a.Hx + iy0) * a.Wx + ix0) * cs;
    const long long dorg = ((long long)(img * a.Hy + oy0) * a.Wy + ox0) * a.cdy;
    const wi32x4 rx = wmake_rsrc_i(xsrc + xorg, (unsigned)(XH * a.Wx * cs * 2));
    const wi32x4 rd = wmake_rsrc_i(dy + dorg, (unsigned)(TH * a.Wy * a.cdy * 2));
    const unsigned m0base = stage_base + wave * 1024;  // piece pc of the tile image lives at byte 1024 pc
#pragma unroll
    for (int j = 0; j < MAXOWN; ++j) {
      const int pc = wave + 4 * j;
      const bool is_x = 4 * j + 3 < XPIECES || (4 * j < XPIECES && pc < XPIECES);
      const bool is_d = !is_x && (4 * j + 3 < PIECES || pc < PIECES);
      if (is_x) {
        const bool rowok = (unsigned)(iy0 + pc / 3) < (unsigned)a.Hx;
        lds_dma16(rx, rowok ? voffc[j] : WSENT, m0base + 4096 * j);
      } else if (is_d) {
        const bool rowok = oy0 + ((pc - XPIECES) >> 1) < a.Hy;
        lds_dma16(rd, rowok ? voffc[j] : WSENT, m0base + 4096 * j);
      }
    }
  };
  auto issue_any = [&](int img, int ty, int tx, unsigned stage_base) {
    if (tx > 0 && tx * 16 + 17 <= a.Wx && tx * 16 + 16 <= a.Wy) issue_fast(img, ty, tx, stage_base);
    else issue(img, ty, tx, stage_base);
  };
  // this wave's pieces per tile: waves with wave < PIECES % 4 own one more
  auto wait_own_in_flight = [&]() {  // all but the newest tile's own pieces have landed
    if (wave < (PIECES & 3)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN - 1) : "memory");
  };

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant fragment bases (absolute LDS bytes of the CURRENT stage; stepped by one stage per tile)
  const int g1 = grp >> 1, xb0 = 8 * (grp & 1) + qp, sub = 8 * (pp & 1);
  unsigned dbase[4][2], xbase[KS][2];  // dbase[c]: LOCAL n tile c of this wave = n tile (cbeg + c) & 3 of the block
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int cg = (cbeg + c) & 3;
    dbase[c][0] = lds0 + X_BYTES + swz_off(g1 * 16 + xb0, 2 * cg + (pp >> 1)) + sub;
    dbase[c][1] = lds0 + X_BYTES + swz_off(g1 * 16 + xb0 + 4, 2 * cg + (pp >> 1)) + sub;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xbase[kw][0] = lds0 + g1 * XROW + swz_off(xb0 + kw, 2 * ktile + (pp >> 1)) + sub;
    xbase[kw][1] = lds0 + g1 * XROW + swz_off(xb0 + kw + 4, 2 * ktile + (pp >> 1)) + sub;
  }

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;  // digits of the NEXT tile to issue
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }
  auto advance = [&]() {
    t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
    t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
    t_img += d_img;
  };
  int issue_tile = tile;       // index of the next tile to issue
  unsigned issue_stage = 0;    // ring slot it goes to
  // prologue: two tiles in flight
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    if (issue_tile < ntiles) { issue_any(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;
  }
  if (tile + a.ksplit < ntiles) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int stage = 0;
#ifdef CONV64_STAMPS
  unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, a_i = 0, a_m = 0, a_w = 0, a_b = 0, a_n = 0, c_t0, c_r0, c_t1, c_r1;
  WSTAMP(c_t0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_r0)::"memory");
#endif
  for (; tile < ntiles; tile += a.ksplit) {
    WSTAMP(w0);
    const bool more = issue_tile < ntiles;
    if (more) { issue_any(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;

    WSTAMP(w1);
    auto mma_tile = [&](auto cc_tag, auto c0_tag) __attribute__((always_inline)) {
      constexpr int CC = decltype(cc_tag)::value, C0 = decltype(c0_tag)::value;  // local n tiles C0 .. C0 + CC - 1 of this wave
      u32x4 af[2][CC], bf[3];
      auto load_a = [&](int kb, int c) -> u32x4 {
        const s16x4 lo = tr_read_at(dbase[C0 + c][0] + 4096 * kb);
        const s16x4 hi = tr_read_at(dbase[C0 + c][1] + 4096 * kb);
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      auto load_b = [&](int step) -> u32x4 {  // step = kb * 9 + tap
        const int kb = step / TAPS, t = step % TAPS, kh = t / KS, kw = t % KS;
        const s16x4 lo = tr_read_at(xbase[kw][0] + XROW * (2 * kb + kh));
        const s16x4 hi = tr_read_at(xbase[kw][1] + XROW * (2 * kb + kh));
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
#pragma unroll
      for (int c = 0; c < CC; ++c) af[0][c] = load_a(0, c);
      bf[0] = load_b(0);
      bf[1] = load_b(1);
#pragma unroll
      for (int step = 0; step < 2 * TAPS; ++step) {
        const int kb = step / TAPS, t = step % TAPS;
        if (step + 2 < 2 * TAPS) bf[(step + 2) % 3] = load_b(step + 2);
        if (kb == 0 && t >= 5 && t - 5 < CC) af[1][t - 5] = load_a(1, t - 5);  // second row block's dy fragments behind the first's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CC; ++c)
          acc[t][C0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kb][c]), __builtin_bit_cast(bf16x8, bf[step % 3]),
                                                                   acc[t][C0 + c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    if constexpr (!NARROW) {
      mma_tile(I4{}, I0{});
    } else {
      if (ccnt >= 2) mma_tile(I2{}, I0{}); else if (ccnt == 1) mma_tile(I1{}, I0{});
      if (ccnt >= 4) mma_tile(I2{}, I2{}); else if (ccnt == 3) mma_tile(I1{}, I2{});
    }
    // next stage's fragment bases
    const int delta = stage == NSTAGE - 1 ? -(NSTAGE - 1) * STAGE : STAGE;
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) { dbase[c][0] += delta; dbase[c][1] += delta; }
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) { xbase[kw][0] += delta; xbase[kw][1] += delta; }
    WSTAMP(w2);
    if (more) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WSTAMP(w3);
    __builtin_amdgcn_s_barrier();
    WSTAMP(w4);
#ifdef CONV64_STAMPS
    a_i += w1 - w0; a_m += w2 - w1; a_w += w3 - w2; a_b += w4 - w3; a_n += 1;
#endif
  }
#ifdef CONV64_STAMPS
  if (lane == 0 && bx == 0 && by < 512) {
    unsigned long long* d = wgrad_dbg + ((size_t)by * 4 + wave) * 8;
    WSTAMP(c_t1);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_r1)::"memory");
    d[0] = a_i; d[1] = a_m; d[2] = a_w; d[3] = a_b; d[4] = 0; d[5] = a_n; d[6] = c_t1 - c_t0; d[7] = c_r1 - c_r0;
  }
#endif
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (cbeg + c) * 16 + 4 * grp + r, k = k0 + ktile * 16 + i16;
        if (c < ccnt && kloc + ktile * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
  // (slab entries this block does not compute -- padded n / k tiles -- are never read: mia_wgrad_reduce sums n < nn, k < kk only)
}

// ---------------------------------------------------------------- bf16, LDS-DMA ring, 96-wide blocks (stride-1 3x3; cfg5's level 0)
// Channel counts that are multiples of 96 and not of 64 cost the 64-wide kernel 2 x 2 blocks per 96 x 96 of dW: 1.78x the MFMAs and the
// x / dy tiles fetched L2 -> LDS four times -- and that fill, not the MFMAs, is what the launch waits for (skipping the empty MFMA tiles
// measured +-0 in round 4; a register-staged 96-wide block measured slower in round 5).  This is the ring kernel on 96-wide images:
//   * a tile is 4 output rows x 16 pixels: x image [6 rows][24 pixels][96 ch] = 27 KB, dy image [64 px][96 ch] = 12 KB, as 32-channel
//     subtiles of 8 pixels x 64 B (THREE per 8-pixel group, same chunk swizzle as the 64-wide image); ring of three images = 117 KB,
//     one 768-thread workgroup per CU;
//   * a DMA piece is 1 KB = two consecutive subtiles: 27 + 12 = 39 pieces per tile dealt round-robin to the twelve waves (3 or 4 each);
//     lane L of a piece is chunk slot L & 3 of pixel (L >> 2) & 7 of subtile 2 p + (L >> 5), the source chunk un-swizzled, out-of-image /
//     out-of-channel lanes pointing past the descriptor (zero fill).  Per-lane offsets are tile-invariant (relative to the tile origin,
//     which rides in the descriptor); what changes per tile is which rows / columns exist;
//   * waves = 6 input-channel tiles x 2 halves of the six output-channel tiles: 27 accumulator tiles, 54 MFMAs per wave and tile;
//   * per tile: issue tile t + 2 -> MFMAs of tile t -> s_waitcnt vmcnt(own pieces of t + 2) -> s_barrier (the 64-wide kernel's protocol).
template <int DUMMY>
__global__ __launch_bounds__(768) void wgrad_bf16_dma96_kernel(const WgArgs a) {
  constexpr int KS = 3, TAPS = 9, TH = 4, CW = 96;
  constexpr int XH = TH + 2, XROW = 3 * 3 * 512;  // 24 pixels = 3 groups of 8, x 3 subtiles of 512 B
  constexpr int X_BYTES = XH * XROW, D_BYTES = TH * 2 * 3 * 512, STAGE = X_BYTES + D_BYTES, NSTAGE = 3;
  constexpr int XPIECES = X_BYTES / 1024, PIECES = XPIECES + D_BYTES / 1024;  // 27 + 12
  constexpr int NWAVE = 12, MAXOWN = (PIECES + NWAVE - 1) / NWAVE, NC = 3;
  static_assert(X_BYTES % 1024 == 0 && D_BYTES % 1024 == 0, "whole pieces");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];
  auto swz = [](int row, int ch) { return 512 * ((row >> 3) * 3 + (ch >> 2)) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3)); };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int ktile = wave % 6, nh = (wave / 6) * NC;
  const int kb1 = (a.c1 + CW - 1) / CW, nkb = kb1 + (a.c2 + CW - 1) / CW;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {  // XCD-aware order (1-D grid)
    const int ncol = nkb * ((a.cdy + CW - 1) / CW), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * CW;
  const int n0 = nblk * CW, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  // tile-invariant lane constants of this wave's pieces: byte offset from the tile origin (WSENT: padding pixel / channel tail) and the
  // (row, column) the lane's pixel has inside the tile, to be checked against the image per tile
  unsigned voffc[MAXOWN];
  int rcc[MAXOWN];  // row << 8 | column (one register per piece: the kernel sits at the 168 registers three waves per SIMD allow)
  const int hl = lane >> 5, r8 = (lane >> 2) & 7, slot4 = lane & 3;
#pragma unroll
  for (int j = 0; j < MAXOWN; ++j) {
    const int pc = wave + NWAVE * j;
    voffc[j] = WSENT; rcc[j] = 0;
    if (pc < XPIECES) {
      const int t = 2 * pc + hl;                       // subtile of the x image
      const int iy = t / 9, gx = (t % 9) / 3, sub = t % 3;
      const int px = 8 * gx + r8, c = 4 * sub + (slot4 ^ ((px >> 2) & 3));
      rcc[j] = iy << 8 | px;
      if (px < 18 && kloc + c * 8 < cs) voffc[j] = (unsigned)(((iy * a.Wx + px) * cs + kloc + c * 8) * 2);
    } else if (pc < PIECES) {
      const int t = 2 * (pc - XPIECES) + hl;           // subtile of the dy image
      const int g8 = t / 3, sub = t % 3;
      const int P = 8 * g8 + r8, c = 4 * sub + (slot4 ^ ((P >> 2) & 3));
      rcc[j] = (P >> 4) << 8 | (P & 15);
      if (n0 + c * 8 < a.cdy) voffc[j] = (unsigned)((((P >> 4) * a.Wy + (P & 15)) * a.cdy + n0 + c * 8) * 2);
    }
  }
  auto issue = [&](int img, int ty, int tx, unsigned stage_base) {
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    // descriptor bases at the tile origin (may lie one row / one pixel in front of the image: those lanes are masked, nothing is read through them)
    const long long xorg = ((long long)(img * a.Hx + iy0) * a.Wx + ix0) * cs;
    const long long dorg = ((long long)(img * a.Hy + oy0) * a.Wy + ox0) * a.cdy;
    // (ranges: the last tile row reaches 17 / 15 pixels past its first column, which is more than an image row when the image is
    // narrower than the tile -- the per-lane row / column masks, not the range, keep the loads inside the tensor)
    const wi32x4 rx = wmake_rsrc_i(xsrc + xorg, (unsigned)((XH * a.Wx + 24) * cs * 2));
    const wi32x4 rd = wmake_rsrc_i(dy + dorg, (unsigned)((TH * a.Wy + 16) * a.cdy * 2));
    const unsigned m0base = stage_base + wave * 1024;  // piece pc of the tile image lives at byte 1024 pc
#pragma unroll
    for (int j = 0; j < MAXOWN; ++j) {
      const int pc = wave + NWAVE * j;  // wave-uniform
      if (pc < XPIECES) {
        const bool ok = ((unsigned)(iy0 + (rcc[j] >> 8)) < (unsigned)a.Hx) & ((unsigned)(ix0 + (rcc[j] & 255)) < (unsigned)a.Wx);
        lds_dma16(rx, ok ? voffc[j] : WSENT, m0base + NWAVE * 1024 * j);
      } else if (pc < PIECES) {
        const bool ok = (oy0 + (rcc[j] >> 8) < a.Hy) & (ox0 + (rcc[j] & 255) < a.Wy);
        lds_dma16(rd, ok ? voffc[j] : WSENT, m0base + NWAVE * 1024 * j);
      }
    }
  };
  auto wait_own_in_flight = [&]() {  // all but the newest tile's own pieces have landed
    if (wave < (PIECES % NWAVE)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN - 1) : "memory");
  };

  f32x4 acc[TAPS][NC];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant fragment bases (absolute LDS bytes of the CURRENT stage; stepped by one stage per tile)
  const int g1 = grp >> 1, xb0 = 8 * (grp & 1) + qp, sub8 = 8 * (pp & 1);
  unsigned dbase[NC][2], xbase[KS][2];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    dbase[c][0] = lds0 + X_BYTES + swz(g1 * 16 + xb0, 2 * (nh + c) + (pp >> 1)) + sub8;
    dbase[c][1] = lds0 + X_BYTES + swz(g1 * 16 + xb0 + 4, 2 * (nh + c) + (pp >> 1)) + sub8;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xbase[kw][0] = lds0 + g1 * XROW + swz(xb0 + kw, 2 * ktile + (pp >> 1)) + sub8;
    xbase[kw][1] = lds0 + g1 * XROW + swz(xb0 + kw + 4, 2 * ktile + (pp >> 1)) + sub8;
  }

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;  // digits of the NEXT tile to issue
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }
  auto advance = [&]() {
    t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
    t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
    t_img += d_img;
  };
  int issue_tile = tile;
  unsigned issue_stage = 0;
#pragma unroll 1
  for (int s_ = 0; s_ < 2; ++s_) {  // prologue: two tiles in flight
    if (issue_tile < ntiles) { issue(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;
  }
  if (tile + a.ksplit < ntiles) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int stage = 0;
  for (; tile < ntiles; tile += a.ksplit) {
    const bool more = issue_tile < ntiles;
    if (more) { issue(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;

    {
      u32x4 af[NC], bf[3];  // (ONE set of dy fragments, reloaded between the two row blocks: a second set does not fit 168 registers)
      auto load_a = [&](int kb, int c) -> u32x4 {  // 32 pixels = 4 groups of 8 x 3 subtiles = 6144 B per row block
        const s16x4 lo = tr_read_at(dbase[c][0] + 6144 * kb);
        const s16x4 hi = tr_read_at(dbase[c][1] + 6144 * kb);
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      auto load_b = [&](int step) -> u32x4 {  // step = kb * 9 + tap
        const int kb = step / TAPS, t = step % TAPS, kh = t / KS, kw = t % KS;
        const s16x4 lo = tr_read_at(xbase[kw][0] + XROW * (2 * kb + kh));
        const s16x4 hi = tr_read_at(xbase[kw][1] + XROW * (2 * kb + kh));
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
#pragma unroll
      for (int c = 0; c < NC; ++c) af[c] = load_a(0, c);
      bf[0] = load_b(0);
      bf[1] = load_b(1);
#pragma unroll
      for (int step = 0; step < 2 * TAPS; ++step) {
        const int kb = step / TAPS, t = step % TAPS;
        if (step + 2 < 2 * TAPS) bf[(step + 2) % 3] = load_b(step + 2);
        if (step == TAPS) {
#pragma unroll
          for (int c = 0; c < NC; ++c) af[c] = load_a(1, c);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[c]), __builtin_bit_cast(bf16x8, bf[step % 3]),
                                                              acc[t][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int delta = stage == NSTAGE - 1 ? -(NSTAGE - 1) * STAGE : STAGE;
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) { dbase[c][0] += delta; dbase[c][1] += delta; }
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) { xbase[kw][0] += delta; xbase[kw][1] += delta; }
    if (more) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (nh + c) * 16 + 4 * grp + r, k = k0 + ktile * 16 + i16;
        if (kloc + ktile * 16 + i16 < cs && n < a.npad) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16, 512-thread big block (stride-1 3x3, cdy % 128 == 0)
// Round 3.  wgrad_bf16_dma_kernel moves 92 bytes L2 -> LDS per MFMA (an 18 KB x tile + an 8 KB dy tile per 288 MFMAs) and runs two
// 256-thread workgroups per CU.  Here ONE 512-thread workgroup per CU owns a 128 n x 64 k x 9 taps block: waves 0-3 take output
// channels n0 .. n0 + 63, waves 4-7 the next 64, both halves share the x tile (59 bytes per MFMA), wave (nh, kq) keeps the same
// 9 x 4 accumulator tiles as before.  Same swizzled LDS images, transposing reads and three-stage LDS-DMA ring; what changes with
// one workgroup per CU is that nothing hides a wave's non-matrix work any more, so (lesson of conv_bt.hip) the DMA pieces of tile
// t + 2 are issued BETWEEN the MFMA steps of tile t instead of in a block in front of them, and everything a piece needs is a lane
// constant or a scalar prepared once per tile.
__global__ __launch_bounds__(512, 2) void wgrad_bf16_bt_kernel(const WgArgs a) {
  constexpr int KS = 3, TAPS = 9, TH = 4;
  constexpr int XH = TH + 2, XROW = 3072;  // 24 pixels x 128 B per image row of the x tile
  constexpr int X_BYTES = XH * XROW, DH_BYTES = TH * 16 * 128, D_BYTES = 2 * DH_BYTES, STAGE = X_BYTES + D_BYTES, NSTAGE = 3;
  constexpr int XPIECES = XH * 3, DPIECES = TH * 2, PIECES = XPIECES + 2 * DPIECES;  // 18 + 8 + 8: piece pc lives at byte 1024 pc
  constexpr int MAXOWN = (PIECES + 7) / 8;                                            // 5 (waves 0, 1) or 4 pieces per wave and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave >> 2, kq = wave & 3;  // 64-channel half of dy and 16-input-channel tile of this wave
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {  // XCD-aware order (1-D grid): the column blocks of one split index share their tiles -> one XCD, one L2
    const int ncol = nkb * (a.npad / 128), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 128, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  // DMA lane constants (as in wgrad_bf16_dma_kernel): lane L of a piece is 16-byte chunk (L&3) of half (L>>5) of pixel row
  // r = (L>>2)&7 of the 8-row block; the source chunk is un-swizzled by the block's parity
  const int dr = (lane >> 2) & 7;
  const int ch8_0 = 4 * (lane >> 5) + ((lane & 3) ^ ((dr >> 2) & 3)), ch8_1 = 4 * (lane >> 5) + ((lane & 3) ^ ((2 + (dr >> 2)) & 3));
  const unsigned xlane0 = (unsigned)((dr * cs + kloc + ch8_0 * 8) * 2), xlane1 = (unsigned)((dr * cs + kloc + ch8_1 * 8) * 2);
  const unsigned dlane0 = (unsigned)((dr * a.cdy + n0 + ch8_0 * 8) * 2), dlane1 = (unsigned)((dr * a.cdy + n0 + ch8_1 * 8) * 2);
  const bool xok0 = kloc + ch8_0 * 8 < cs, xok1 = kloc + ch8_1 * 8 < cs;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  // per-piece lane offsets of a tile whose 18 columns lie inside the image (descriptor based at the tile origin); padding /
  // channel-tail lanes already point out of range.  dy piece q = (half h = q >> 3, 8-pixel block q & 7) reads channels n0 + 64 h ..
  unsigned voffc[MAXOWN];
#pragma unroll
  for (int j = 0; j < MAXOWN; ++j) {
    const int pc = wave + 8 * j;
    if (pc < XPIECES) {
      const int iy = pc / 3, xb = pc - 3 * iy, ix = 8 * xb + dr;
      const bool ok = (ix < 18) & ((xb & 1) ? xok1 : xok0);
      voffc[j] = ok ? (unsigned)((iy * a.Wx + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0) : WSENT;
    } else {
      const int q = pc - XPIECES, h = q >> 3, qq = q & 7;
      const bool ok = pc < PIECES;
      voffc[j] = ok ? (unsigned)(((qq >> 1) * a.Wy + 8 * (qq & 1)) * a.cdy * 2) + ((qq & 1) ? dlane1 : dlane0) + (unsigned)(h * 128) : WSENT;
    }
  }

  // ---- issue state of the tile being fetched (prepared once per tile, pieces issued between the MFMA steps)
  wi32x4 rx, rd;
  bool fastp = false;
  int i_oy0 = 0, i_ox0 = 0;
  unsigned i_stage = 0;
  auto prepare = [&](int img, int ty, int tx, unsigned stage_base) {
    i_oy0 = ty * TH; i_ox0 = tx * 16; i_stage = stage_base;
    fastp = tx > 0 && tx * 16 + 17 <= a.Wx && tx * 16 + 16 <= a.Wy;  // uniform
    if (fastp) {  // descriptor bases at the tile origin (row iy0 may be -1: its pieces are dropped, nothing is read through it)
      const long long xorg = ((long long)(img * a.Hx + i_oy0 - 1) * a.Wx + i_ox0 - 1) * cs;
      const long long dorg = ((long long)(img * a.Hy + i_oy0) * a.Wy + i_ox0) * a.cdy;
      rx = wmake_rsrc_i(xsrc + xorg, (unsigned)(XH * a.Wx * cs * 2));
      rd = wmake_rsrc_i(dy + dorg, (unsigned)(TH * a.Wy * a.cdy * 2));
    } else {
      rx = wmake_rsrc_i(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
      rd = wmake_rsrc_i(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
    }
  };
  auto issue_piece = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int pc = wave + 8 * j;  // wave-uniform piece index
    if (pc >= PIECES) return;
    const unsigned dst = i_stage + pc * 1024;
    const int iy0 = i_oy0 - 1, ix0 = i_ox0 - 1;
    if (fastp) {
      if (pc < XPIECES) {
        const bool rowok = (unsigned)(iy0 + pc / 3) < (unsigned)a.Hx;
        lds_dma16(rx, rowok ? voffc[j] : WSENT, __builtin_amdgcn_readfirstlane(dst));
      } else {
        const bool rowok = i_oy0 + (((pc - XPIECES) & 7) >> 1) < a.Hy;
        lds_dma16(rd, rowok ? voffc[j] : WSENT, __builtin_amdgcn_readfirstlane(dst));
      }
    } else if (pc < XPIECES) {
      const int iy = pc / 3, xb = pc - 3 * iy;
      const int gy = iy0 + iy, gx = ix0 + 8 * xb + dr;
      const bool ok = ((unsigned)gy < (unsigned)a.Hx) & ((unsigned)gx < (unsigned)a.Wx) & (8 * xb + dr < 18) & ((xb & 1) ? xok1 : xok0);
      const unsigned off = (unsigned)((gy * a.Wx + ix0 + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0);
      lds_dma16(rx, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    } else {
      const int q = pc - XPIECES, h = q >> 3, qq = q & 7;  // 8-pixel block of the dy tile: output row qq >> 1, pixels 8 (qq & 1) ..
      const int gy = i_oy0 + (qq >> 1), gx = i_ox0 + 8 * (qq & 1) + dr;
      const bool ok = (gy < a.Hy) & (gx < a.Wy);
      const unsigned off = (unsigned)((gy * a.Wy + i_ox0 + 8 * (qq & 1)) * a.cdy * 2) + ((qq & 1) ? dlane1 : dlane0) + (unsigned)(h * 128);
      lds_dma16(rd, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    }
  };
  auto issue_all = [&]() {
    issue_piece(std::integral_constant<int, 0>{}); issue_piece(std::integral_constant<int, 1>{}); issue_piece(std::integral_constant<int, 2>{});
    issue_piece(std::integral_constant<int, 3>{}); issue_piece(std::integral_constant<int, 4>{});
  };
  // this wave's pieces per tile: waves with wave < PIECES % 8 own one more
  auto wait_own_in_flight = [&]() {  // all but the newest tile's own pieces have landed
    if (wave < (PIECES & 7)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN - 1) : "memory");
  };

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant fragment bases (absolute LDS bytes of the CURRENT stage; stepped by one stage per tile)
  const int g1 = grp >> 1, xb0 = 8 * (grp & 1) + qp, sub = 8 * (pp & 1);
  unsigned dbase[2][2], xbase[KS][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {  // channel tiles c and c + 2 differ by +512 bytes
    dbase[c][0] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0, 2 * c + (pp >> 1)) + sub;
    dbase[c][1] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0 + 4, 2 * c + (pp >> 1)) + sub;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xbase[kw][0] = lds0 + g1 * XROW + swz_off(xb0 + kw, 2 * kq + (pp >> 1)) + sub;
    xbase[kw][1] = lds0 + g1 * XROW + swz_off(xb0 + kw + 4, 2 * kq + (pp >> 1)) + sub;
  }

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;  // digits of the NEXT tile to issue
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }
  auto advance = [&]() {
    t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
    t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
    t_img += d_img;
  };
  int issue_tile = tile;       // index of the next tile to issue
  unsigned issue_stage = 0;    // ring slot it goes to
  // prologue: two tiles in flight
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    if (issue_tile < ntiles) { prepare(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); issue_all(); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;
  }
  if (tile + a.ksplit < ntiles) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int stage = 0;
  for (; tile < ntiles; tile += a.ksplit) {
    const bool more = issue_tile < ntiles;  // uniform
    if (more) { prepare(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;

    u32x4 af[2][4], bf[3];
    auto load_a = [&](int kb, int c) -> u32x4 {
      const s16x4 lo = tr_read_at(dbase[c & 1][0] + 512 * (c >> 1) + 4096 * kb);
      const s16x4 hi = tr_read_at(dbase[c & 1][1] + 512 * (c >> 1) + 4096 * kb);
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto load_b = [&](int step) -> u32x4 {  // step = kb * 9 + tap
      const int kb = step / TAPS, t = step % TAPS, kh = t / KS, kw = t % KS;
      const s16x4 lo = tr_read_at(xbase[kw][0] + XROW * (2 * kb + kh));
      const s16x4 hi = tr_read_at(xbase[kw][1] + XROW * (2 * kb + kh));
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) af[0][c] = load_a(0, c);
    bf[0] = load_b(0);
    bf[1] = load_b(1);
#pragma unroll
    for (int step = 0; step < 2 * TAPS; ++step) {
      const int kb = step / TAPS, t = step % TAPS;
      if (step + 2 < 2 * TAPS) bf[(step + 2) % 3] = load_b(step + 2);
      if (kb == 0 && t >= 5 && t <= 8) af[1][t - 5] = load_a(1, t - 5);  // second row block's dy fragments behind the first's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kb][c]), __builtin_bit_cast(bf16x8, bf[step % 3]),
                                                            acc[t][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the pieces of tile t + 2 behind the MFMAs of steps 1, 4, 7, 10, 13
      if (more) {
        if (step == 1) issue_piece(std::integral_constant<int, 0>{});
        if (step == 4) issue_piece(std::integral_constant<int, 1>{});
        if (step == 7) issue_piece(std::integral_constant<int, 2>{});
        if (step == 10) issue_piece(std::integral_constant<int, 3>{});
        if (step == 13) issue_piece(std::integral_constant<int, 4>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // next stage's fragment bases
    const int delta = stage == NSTAGE - 1 ? -(NSTAGE - 1) * STAGE : STAGE;
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
#pragma unroll
    for (int c = 0; c < 2; ++c) { dbase[c][0] += delta; dbase[c][1] += delta; }
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) { xbase[kw][0] += delta; xbase[kw][1] += delta; }
    if (more) wait_own_in_flight(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * nh + c * 16 + 4 * grp + r, k = k0 + kq * 16 + i16;
        if (kloc + kq * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16, 512-thread big block, 3x3 STRIDE 2 (cdy % 128 == 0)
// The stride-2 weight gradient (first conv of every encoder level, unet.py:57) ran on wgrad_bf16_fast_kernel<MODE_W3S2>: 436
// registers = one wave per SIMD, register staging, two barriers per tile, 0.5 PFLOP/s.  Same block and wave roles as
// wgrad_bf16_bt_kernel (128 n x 64 k x 9 taps, waves = 2 n halves x 4 k tiles); differences:
//   * x tile of 4 output rows x 16 pixels = 9 input rows x 33 pixels, kept as [9 rows][40-pixel pitch][64 ch] (45 pieces of 8
//     pixels x 128 B, swizzled within a row like the stride-1 image): the pitch is a multiple of 8 pixels, so a tap row (kh) and a
//     row block (kb) are IMMEDIATE offsets of the transposing reads (5 KB, 20 KB): six lane-constant bases (kw x the two 4-pixel
//     halves of a fragment) serve all 36 x reads of a tile.
//     A transposing read takes its four K rows from four lane-supplied addresses, so the stride-2 pixel gather costs nothing;
//   * 61 KB per stage -> a ring of TWO stages (122 KB): tile t + 1 is issued during the first MFMA steps of tile t and waited
//     for (vmcnt(0)) at its end.
__global__ __launch_bounds__(512, 2) void wgrad_bf16_bt_s2_kernel(const WgArgs a) {
  constexpr int KS = 3, TAPS = 9, TH = 4;
  constexpr int XH = 2 * TH + 1, XBLK = 5, XROW = XBLK * 1024, XW = 33;  // 9 rows x 5 blocks of 8 pixels (33 used) x 128 B
  constexpr int X_BYTES = XH * XROW, DH_BYTES = TH * 16 * 128, D_BYTES = 2 * DH_BYTES, STAGE = X_BYTES + D_BYTES, NSTAGE = 2;
  constexpr int XPIECES = XH * XBLK, DPIECES = TH * 2, PIECES = XPIECES + 2 * DPIECES;  // 45 + 8 + 8: piece pc lives at byte 1024 pc
  constexpr int MAXOWN = (PIECES + 7) / 8;                                              // 8 (waves 0-4) or 7 pieces per wave and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave >> 2, kq = wave & 3;
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {
    const int ncol = nkb * (a.npad / 128), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 128, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  // DMA lane constants: lane L of a piece = 16-byte chunk (L&3) of half (L>>5) of pixel row (L>>2)&7 of the 8-pixel block; the
  // source chunk is un-swizzled by the block's parity within its image row
  const int dr = (lane >> 2) & 7;
  const int ch8_0 = 4 * (lane >> 5) + ((lane & 3) ^ ((dr >> 2) & 3)), ch8_1 = 4 * (lane >> 5) + ((lane & 3) ^ ((2 + (dr >> 2)) & 3));
  const unsigned xlane0 = (unsigned)((dr * cs + kloc + ch8_0 * 8) * 2), xlane1 = (unsigned)((dr * cs + kloc + ch8_1 * 8) * 2);
  const unsigned dlane0 = (unsigned)((dr * a.cdy + n0 + ch8_0 * 8) * 2), dlane1 = (unsigned)((dr * a.cdy + n0 + ch8_1 * 8) * 2);
  const bool xok0 = kloc + ch8_0 * 8 < cs, xok1 = kloc + ch8_1 * 8 < cs;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  // ---- issue state of the tile being fetched
  wi32x4 rx, rd;
  int i_oy0 = 0, i_ox0 = 0;
  unsigned i_stage = 0;
  auto prepare = [&](int img, int ty, int tx, unsigned stage_base) {
    i_oy0 = ty * TH; i_ox0 = tx * 16; i_stage = stage_base;
    rx = wmake_rsrc_i(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
    rd = wmake_rsrc_i(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
  };
  auto issue_piece = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int pc = wave + 8 * j;  // wave-uniform piece index
    if (pc >= PIECES) return;
    const unsigned dst = i_stage + pc * 1024;
    if (pc < XPIECES) {
      const int iy = pc / XBLK, xb = pc - XBLK * iy;
      const int iy0 = 2 * i_oy0 - 1, ix0 = 2 * i_ox0 - 1;
      const int gy = iy0 + iy, gx = ix0 + 8 * xb + dr;
      const bool ok = ((unsigned)gy < (unsigned)a.Hx) & ((unsigned)gx < (unsigned)a.Wx) & (8 * xb + dr < XW) & ((xb & 1) ? xok1 : xok0);
      const unsigned off = (unsigned)((gy * a.Wx + ix0 + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0);
      lds_dma16(rx, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    } else {
      const int q = pc - XPIECES, h = q >> 3, qq = q & 7;  // 8-pixel block of the dy tile: output row qq >> 1, pixels 8 (qq & 1) ..
      const int gy = i_oy0 + (qq >> 1), gx = i_ox0 + 8 * (qq & 1) + dr;
      const bool ok = (gy < a.Hy) & (gx < a.Wy);
      const unsigned off = (unsigned)((gy * a.Wy + i_ox0 + 8 * (qq & 1)) * a.cdy * 2) + ((qq & 1) ? dlane1 : dlane0) + (unsigned)(h * 128);
      lds_dma16(rd, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    }
  };
#define WG_PIECE(J) issue_piece(std::integral_constant<int, J>{})

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant fragment bases (absolute LDS bytes of stage 0; the stage offset is added per tile)
  const int g1 = grp >> 1, xb0 = 8 * (grp & 1) + qp, sub = 8 * (pp & 1);
  unsigned dbase[2][2], xbase[KS][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {  // channel tiles c and c + 2 differ by +512 bytes
    dbase[c][0] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0, 2 * c + (pp >> 1)) + sub;
    dbase[c][1] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0 + 4, 2 * c + (pp >> 1)) + sub;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {  // input row 4 kb + 2 g1 + kh, input column 2 (output pixel) + kw; +4 output pixels = +8 columns
    xbase[kw][0] = lds0 + 2 * g1 * XROW + swz_off(2 * xb0 + kw, 2 * kq + (pp >> 1)) + sub;
    xbase[kw][1] = lds0 + 2 * g1 * XROW + swz_off(2 * xb0 + kw + 8, 2 * kq + (pp >> 1)) + sub;  // (+8 flips the swizzle's bit 1)
  }

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;  // digits of the NEXT tile to issue
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }
  auto advance = [&]() {
    t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
    t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
    t_img += d_img;
  };
  int issue_tile = tile;
  unsigned stage = 0;  // stage the CURRENT tile is read from
  if (issue_tile < ntiles) {
    prepare(t_img, t_ty, t_tx, lds0);
    WG_PIECE(0); WG_PIECE(1); WG_PIECE(2); WG_PIECE(3); WG_PIECE(4); WG_PIECE(5); WG_PIECE(6); WG_PIECE(7);
    advance();
  }
  issue_tile += a.ksplit;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (; tile < ntiles; tile += a.ksplit) {
    const bool more = issue_tile < ntiles;  // uniform
    if (more) { prepare(t_img, t_ty, t_tx, lds0 + (stage ^ 1) * STAGE); advance(); }
    issue_tile += a.ksplit;
    const unsigned so = stage * STAGE;

    u32x4 af[2][4], bf[3];
    auto load_a = [&](int kb, int c) -> u32x4 {
      const s16x4 lo = tr_read_at(dbase[c & 1][0] + so + 512 * (c >> 1) + 4096 * kb);
      const s16x4 hi = tr_read_at(dbase[c & 1][1] + so + 512 * (c >> 1) + 4096 * kb);
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto load_b = [&](int step) -> u32x4 {  // step = kb * 9 + tap
      const int kb = step / TAPS, t = step % TAPS, kh = t / KS, kw = t % KS;
      const s16x4 lo = tr_read_at(xbase[kw][0] + so + XROW * (4 * kb + kh));
      const s16x4 hi = tr_read_at(xbase[kw][1] + so + XROW * (4 * kb + kh));
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) af[0][c] = load_a(0, c);
    bf[0] = load_b(0);
    bf[1] = load_b(1);
#pragma unroll
    for (int step = 0; step < 2 * TAPS; ++step) {
      const int kb = step / TAPS, t = step % TAPS;
      if (step + 2 < 2 * TAPS) bf[(step + 2) % 3] = load_b(step + 2);
      if (kb == 0 && t >= 5 && t <= 8) af[1][t - 5] = load_a(1, t - 5);  // second row block's dy fragments behind the first's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kb][c]), __builtin_bit_cast(bf16x8, bf[step % 3]),
                                                            acc[t][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // tile t + 1: two pieces behind each of the first four steps, so that they have the rest of the tile to land
      if (more) {
        if (step == 0) { WG_PIECE(0); WG_PIECE(1); }
        if (step == 1) { WG_PIECE(2); WG_PIECE(3); }
        if (step == 2) { WG_PIECE(4); WG_PIECE(5); }
        if (step == 3) { WG_PIECE(6); WG_PIECE(7); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage ^= 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef WG_PIECE
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * nh + c * 16 + 4 * grp + r, k = k0 + kq * 16 + i16;
        if (kloc + kq * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- bf16, 512-thread big block, ConvTranspose 2x2 / stride 2 (cdy % 128 == 0)
// The weight gradient of nn.ConvTranspose2d(c_below, c, 2, 2) (unet.py:142): "x" = the fine output gradient, "dy" = the coarse input,
// four taps without overlap.  Same block, wave roles, LDS images and transposing reads as wgrad_bf16_bt_s2_kernel with KS = 2 and no
// padding row / column: x tile = 8 fine rows x 32 pixels (32 pieces), 48 KB per stage -> a ring of THREE stages (tile t + 2 issued one
// piece per MFMA step of tile t, counted wait for tile t + 1): 32 MFMAs per wave and tile are too few to hide a DMA round trip
// inside one tile, which the two-stage ring of the 3x3 kernel relies on.
__global__ __launch_bounds__(512, 2) void wgrad_bf16_bt_t2_kernel(const WgArgs a) {
  constexpr int KS = 2, TAPS = 4, TH = 4;
  constexpr int XH = 2 * TH, XBLK = 4, XROW = XBLK * 1024, XW = 32;  // 8 fine rows x 4 blocks of 8 pixels x 128 B
  constexpr int X_BYTES = XH * XROW, DH_BYTES = TH * 16 * 128, D_BYTES = 2 * DH_BYTES, STAGE = X_BYTES + D_BYTES, NSTAGE = 3;
  constexpr int XPIECES = XH * XBLK, DPIECES = TH * 2, PIECES = XPIECES + 2 * DPIECES;  // 32 + 8 + 8: piece pc lives at byte 1024 pc
  constexpr int MAXOWN = PIECES / 8;                                                    // 6 pieces per wave and tile, every wave
  static_assert(PIECES % 8 == 0, "uniform piece count: one counted wait for all waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave >> 2, kq = wave & 3;
  const int grp = lane >> 4, i16 = lane & 15, qp = i16 >> 2, pp = i16 & 3;
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.opt & 16) {
    const int ncol = nkb * (a.npad / 128), slot = bx >> 3;
    by = (slot / ncol) * 8 + (bx & 7);
    bx = slot % ncol;
    if (by >= a.ksplit) return;
  }
  const int kblk = bx % nkb, nblk = bx / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 128, k0 = (second ? a.c1 : 0) + kloc;
  const bf16_t* xsrc = static_cast<const bf16_t*>(second ? a.x2 : a.x1);
  const bf16_t* dy = static_cast<const bf16_t*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  // DMA lane constants: lane L of a piece = 16-byte chunk (L&3) of half (L>>5) of pixel row (L>>2)&7 of the 8-pixel block; the
  // source chunk is un-swizzled by the block's parity within its image row
  const int dr = (lane >> 2) & 7;
  const int ch8_0 = 4 * (lane >> 5) + ((lane & 3) ^ ((dr >> 2) & 3)), ch8_1 = 4 * (lane >> 5) + ((lane & 3) ^ ((2 + (dr >> 2)) & 3));
  const unsigned xlane0 = (unsigned)((dr * cs + kloc + ch8_0 * 8) * 2), xlane1 = (unsigned)((dr * cs + kloc + ch8_1 * 8) * 2);
  const unsigned dlane0 = (unsigned)((dr * a.cdy + n0 + ch8_0 * 8) * 2), dlane1 = (unsigned)((dr * a.cdy + n0 + ch8_1 * 8) * 2);
  const bool xok0 = kloc + ch8_0 * 8 < cs, xok1 = kloc + ch8_1 * 8 < cs;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  // ---- issue state of the tile being fetched
  wi32x4 rx, rd;
  int i_oy0 = 0, i_ox0 = 0;
  unsigned i_stage = 0;
  auto prepare = [&](int img, int ty, int tx, unsigned stage_base) {
    i_oy0 = ty * TH; i_ox0 = tx * 16; i_stage = stage_base;
    rx = wmake_rsrc_i(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 2));
    rd = wmake_rsrc_i(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 2));
  };
  auto issue_piece = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int pc = wave + 8 * j;  // wave-uniform piece index
    if (pc >= PIECES) return;
    const unsigned dst = i_stage + pc * 1024;
    if (pc < XPIECES) {
      const int iy = pc / XBLK, xb = pc - XBLK * iy;
      const int iy0 = 2 * i_oy0, ix0 = 2 * i_ox0;
      const int gy = iy0 + iy, gx = ix0 + 8 * xb + dr;
      const bool ok = ((unsigned)gy < (unsigned)a.Hx) & ((unsigned)gx < (unsigned)a.Wx) & (8 * xb + dr < XW) & ((xb & 1) ? xok1 : xok0);
      const unsigned off = (unsigned)((gy * a.Wx + ix0 + 8 * xb) * cs * 2) + ((xb & 1) ? xlane1 : xlane0);
      lds_dma16(rx, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    } else {
      const int q = pc - XPIECES, h = q >> 3, qq = q & 7;  // 8-pixel block of the dy tile: output row qq >> 1, pixels 8 (qq & 1) ..
      const int gy = i_oy0 + (qq >> 1), gx = i_ox0 + 8 * (qq & 1) + dr;
      const bool ok = (gy < a.Hy) & (gx < a.Wy);
      const unsigned off = (unsigned)((gy * a.Wy + i_ox0 + 8 * (qq & 1)) * a.cdy * 2) + ((qq & 1) ? dlane1 : dlane0) + (unsigned)(h * 128);
      lds_dma16(rd, ok ? off : WSENT, __builtin_amdgcn_readfirstlane(dst));
    }
  };
#define WG_PIECE(J) issue_piece(std::integral_constant<int, J>{})

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant fragment bases (absolute LDS bytes of stage 0; the stage offset is added per tile)
  const int g1 = grp >> 1, xb0 = 8 * (grp & 1) + qp, sub = 8 * (pp & 1);
  unsigned dbase[2][2], xbase[KS][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {  // channel tiles c and c + 2 differ by +512 bytes
    dbase[c][0] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0, 2 * c + (pp >> 1)) + sub;
    dbase[c][1] = lds0 + X_BYTES + nh * DH_BYTES + swz_off(g1 * 16 + xb0 + 4, 2 * c + (pp >> 1)) + sub;
  }
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {  // input row 4 kb + 2 g1 + kh, input column 2 (output pixel) + kw; +4 output pixels = +8 columns
    xbase[kw][0] = lds0 + 2 * g1 * XROW + swz_off(2 * xb0 + kw, 2 * kq + (pp >> 1)) + sub;
    xbase[kw][1] = lds0 + 2 * g1 * XROW + swz_off(2 * xb0 + kw + 8, 2 * kq + (pp >> 1)) + sub;  // (+8 flips the swizzle's bit 1)
  }

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  int tile = by;
  int t_tx, t_ty, t_img;  // digits of the NEXT tile to issue
  { int tt = tile; t_tx = tt % a.tiles_x; tt /= a.tiles_x; t_ty = tt % a.tiles_y; t_img = tt / a.tiles_y; }
  int d_tx, d_ty, d_img;
  { int tt = a.ksplit; d_tx = tt % a.tiles_x; tt /= a.tiles_x; d_ty = tt % a.tiles_y; d_img = tt / a.tiles_y; }
  auto advance = [&]() {
    t_tx += d_tx; if (t_tx >= a.tiles_x) { t_tx -= a.tiles_x; t_ty += 1; }
    t_ty += d_ty; if (t_ty >= a.tiles_y) { t_ty -= a.tiles_y; t_img += 1; }
    t_img += d_img;
  };
  int issue_tile = tile;
  unsigned stage = 0, issue_stage = 0;  // ring slot the CURRENT tile is read from / the next tile goes to
  int issued = 0;
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {  // prologue: two tiles in flight
    if (issue_tile < ntiles) {
      prepare(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE);
      WG_PIECE(0); WG_PIECE(1); WG_PIECE(2); WG_PIECE(3); WG_PIECE(4); WG_PIECE(5);
      advance();
      ++issued;
    }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;
  }
  if (issued == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (; tile < ntiles; tile += a.ksplit) {
    const bool more = issue_tile < ntiles;  // uniform
    if (more) { prepare(t_img, t_ty, t_tx, lds0 + issue_stage * STAGE); advance(); }
    issue_tile += a.ksplit;
    issue_stage = issue_stage == NSTAGE - 1 ? 0 : issue_stage + 1;
    const unsigned so = stage * STAGE;

    u32x4 af[2][4], bf[3];
    auto load_a = [&](int kb, int c) -> u32x4 {
      const s16x4 lo = tr_read_at(dbase[c & 1][0] + so + 512 * (c >> 1) + 4096 * kb);
      const s16x4 hi = tr_read_at(dbase[c & 1][1] + so + 512 * (c >> 1) + 4096 * kb);
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto load_b = [&](int step) -> u32x4 {  // step = kb * 4 + tap
      const int kb = step / TAPS, t = step % TAPS, kh = t / KS, kw = t % KS;
      const s16x4 lo = tr_read_at(xbase[kw][0] + so + XROW * (4 * kb + kh));
      const s16x4 hi = tr_read_at(xbase[kw][1] + so + XROW * (4 * kb + kh));
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) af[0][c] = load_a(0, c);
    bf[0] = load_b(0);
    bf[1] = load_b(1);
#pragma unroll
    for (int step = 0; step < 2 * TAPS; ++step) {
      const int kb = step / TAPS, t = step % TAPS;
      if (step + 2 < 2 * TAPS) bf[(step + 2) % 3] = load_b(step + 2);
      if (kb == 0) af[1][t] = load_a(1, t);  // second row block's dy fragments behind the first's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kb][c]), __builtin_bit_cast(bf16x8, bf[step % 3]),
                                                            acc[t][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // tile t + 2: one piece behind each of the first six steps
      if (more) {
        if (step == 0) WG_PIECE(0);
        if (step == 1) WG_PIECE(1);
        if (step == 2) WG_PIECE(2);
        if (step == 3) WG_PIECE(3);
        if (step == 4) WG_PIECE(4);
        if (step == 5) WG_PIECE(5);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    // own pieces of the tile after next may stay in flight; the next tile's have landed
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXOWN) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef WG_PIECE
  float* slab = a.slabs + (size_t)by * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * nh + c * 16 + 4 * grp + r, k = k0 + kq * 16 + i16;
        if (kloc + kq * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- fp32
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgArgs a) {
  using G = WGeo<MODE>;
  constexpr int KS = G::KS, S = G::S, PAD = G::PAD, TAPS = G::TAPS;
  constexpr int TH = (S == 1) ? 4 : 2;
  constexpr int XH = (TH - 1) * S + KS, XW = 15 * S + KS;
  constexpr int PS = 80;  // LDS pixel stride in dwords (64 channels + 16 pad)
  __shared__ __attribute__((aligned(16))) float smem[(XH * XW + TH * 16) * PS];
  float* xs = smem;
  float* ds = smem + XH * XW * PS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  const int nkb = a.kpad / 64;
  const int kblk = blockIdx.x % nkb, nblk = blockIdx.x / nkb;
  const int n0 = nblk * 64, k0 = kblk * 64;
  const int kin = a.c1 + a.c2;
  const float* x1 = static_cast<const float*>(a.x1);
  const float* x2 = static_cast<const float*>(a.x2);
  const float* dy = static_cast<const float*>(a.dy);

  f32x4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_active = (k0 + wave * 16) < kin;

  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  for (int tile = blockIdx.y; tile < ntiles; tile += a.ksplit) {
    int tt = tile;
    const int tx = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty = tt % a.tiles_y; tt /= a.tiles_y;
    const int img = tt;
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    __syncthreads();
    for (int u = tid; u < XH * XW * 16; u += 256) {
      const int ch = u & 15, pix = u >> 4;
      const int iy = pix / XW, ix = pix - iy * XW;
      const int gy = iy0 + iy, gx = ix0 + ix, c = k0 + ch * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < a.Hx && gx >= 0 && gx < a.Wx && c < kin) {
        const size_t p = ((size_t)img * a.Hx + gy) * a.Wx + gx;
        if (a.vec_x) {
          const float* src = (c < a.c1) ? x1 + p * a.c1 + c : x2 + p * a.c2 + (c - a.c1);
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ce = c + e;
            v[e] = ce < a.c1 ? x1[p * a.c1 + ce] : (ce < kin ? x2[p * a.c2 + (ce - a.c1)] : 0.f);
          }
        }
      }
      *reinterpret_cast<f32x4*>(xs + pix * PS + ch * 4) = v;
    }
    for (int u = tid; u < TH * 16 * 16; u += 256) {
      const int ch = u & 15, pix = u >> 4;
      const int y = pix >> 4, xx = pix & 15;
      const int gy = oy0 + y, gx = ox0 + xx, c = n0 + ch * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gy < a.Hy && gx < a.Wy && c < a.cdy) {
        const size_t p = ((size_t)img * a.Hy + gy) * a.Wy + gx;
        if (a.vec_dy) {
          v = *reinterpret_cast<const f32x4*>(dy + p * a.cdy + c);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (c + e < a.cdy) ? dy[p * a.cdy + c + e] : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(ds + pix * PS + ch * 4) = v;
    }
    __syncthreads();
    if (wave_active) {
      for (int y = 0; y < TH; ++y) {
#pragma unroll
        for (int xq = 0; xq < 4; ++xq) {
          const int xx = xq * 4 + q;  // this lane's pixel (k index) within the row
          float af[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) af[c] = ds[(y * 16 + xx) * PS + c * 16 + i16];
#pragma unroll
          for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
              const float b = xs[((y * S + kh) * XW + xx * S + kw) * PS + wave * 16 + i16];
#pragma unroll
              for (int c = 0; c < 4; ++c)
                acc[kh * KS + kw][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c], b, acc[kh * KS + kw][c], 0, 0, 0);
            }
        }
      }
    }
  }
  float* slab = a.slabs + (size_t)blockIdx.y * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + c * 16 + 4 * q + r, k = k0 + wave * 16 + i16;
        slab[((size_t)t * a.npad + n) * a.kpad + k] = acc[t][c][r];
      }
}

// ---------------------------------------------------------------- fp32 fast path (same contract as the bf16 one)
// NARROW (32-channel layers: cdy <= 32 and every source <= 32 channels, e.g. al_train's first level): the 64 x 64 block would
// multiply 75 % zeros (1.22 ms vs 0.41 ms for the forward conv of the same layer).  The four waves become 2 input-channel tiles x 2
// halves of the tile's pixel rows with 2 output-channel tiles each; the two pixel halves are summed through LDS at the end.
// SPLIT (option f32_split): the products run on the f16 matrix cores from two-part split operands (SplitF16, common.h), each operand
// tensor scaled by the power of two its maximum dictates (a block's input channels lie in ONE source, so x1 and x2 keep their own scale).
// LDS holds one (h | l << 16) word per element in the exact kernel's layout; the reduction dimension of an MFMA is 16 pixels
// (one tile row) x 2 parts: lane group q takes pixels q, q + 4, q + 8, q + 12 (the exact kernel's conflict-free bank pattern),
// the dy fragment is expanded to its (H, H) and (L, L) forms once per row and meets every tap's (h, l) x fragment in two MFMAs; the
// accumulators are scaled back when the slab is written.
// Full blocks (round 5): 512 threads -- eight waves = 4 input-channel tiles x 2 halves of the block's output channels, 72 accumulator
// registers per wave instead of 144.  The 256-thread form needed 364 registers (hipcc parks accumulators in AGPRs), i.e. ONE wave per
// SIMD and one workgroup per CU: staging and MFMA phases never overlapped (PMC: 45 % matrix-pipe busy); forced to 256 registers it spilled.
template <int MODE, bool NARROW = false, bool SPLIT = false>
__global__ __launch_bounds__(NARROW ? 256 : 512, 2) void wgrad_f32_fast_kernel(const WgArgs a) {
  using G = WGeo<MODE>;
  constexpr int KS = G::KS, S = G::S, PAD = G::PAD, TAPS = G::TAPS;
  // N8 (narrow stride-1 3x3, round 5: al_train's 32-channel first level): 8-row tiles -- twice the MFMAs per barrier pair, a 10-row halo tile for
  // 8 rows instead of 6 for 4 -- on a 48-dword pixel stride (32 channels + 16 pad: the same conflict-free bank pattern as 80) and staging
  // lanes dealt 32 pixels x 8 units, so no lane idles on the 32 channels that do not exist.  61 KB of LDS: still two workgroups per CU.
  constexpr bool N8 = NARROW && MODE == MODE_W3S1;
  constexpr int TH = N8 ? 8 : ((S == 1) ? 4 : 2);
  constexpr int XH = (TH - 1) * S + KS, XW = 15 * S + KS;
  constexpr int PS = N8 ? 48 : 80;  // LDS pixel stride in dwords (64 channels + 16 pad): conflict-free b32 fragment reads
  constexpr int NTHR = NARROW ? 256 : 512;
  constexpr int UL = N8 ? 8 : 16, PPI = NTHR / UL;  // four-channel units staged per pixel; pixels per staging iteration
  constexpr int X_IT = (XH * XW + PPI - 1) / PPI, D_IT = TH * 16 / PPI;
  __shared__ __attribute__((aligned(16))) float smem[(X_IT * PPI + TH * 16) * PS];
  float* xs = smem;
  float* ds = smem + X_IT * PPI * PS;

  constexpr int NC = 2;  // output-channel tiles per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = NARROW ? (wave & 1) : (wave & 3), ph = NARROW ? (wave >> 1) : 0;  // input-channel tile; half of the tile's pixel rows
  const int nh = NARROW ? 0 : 2 * (wave >> 2);  // first output-channel tile of this wave (full blocks: waves 4 .. 7 take tiles 2, 3)
  const int q = lane >> 4, i16 = lane & 15;
  const int ch4 = tid % UL, p16 = tid / UL;
  // 64-channel input blocks are cut per SOURCE (ceil(c1/64) + ceil(c2/64) of them), so a block never straddles the
  // two tensors of a concatenated input whatever c1 is; a source's last block may be partial (lanes beyond cs read
  // zeros and do not store)
  const int kb1 = (a.c1 + 63) / 64, nkb = kb1 + (a.c2 + 63) / 64;
  const int kblk = blockIdx.x % nkb, nblk = blockIdx.x / nkb;
  const bool second = kblk >= kb1;
  const int cs = second ? a.c2 : a.c1, kloc = (second ? kblk - kb1 : kblk) * 64;
  const int n0 = nblk * 64, k0 = (second ? a.c1 : 0) + kloc;
  const float* xsrc = static_cast<const float*>(second ? a.x2 : a.x1);
  const float* dy = static_cast<const float*>(a.dy);
  const size_t xpix = (size_t)a.Hx * a.Wx, ypix = (size_t)a.Hy * a.Wy;

  int x_iy[X_IT], x_ix[X_IT];
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int pix = p16 + PPI * i;
    x_iy[i] = pix < XH * XW ? pix / XW : -100000;
    x_ix[i] = pix - (pix / XW) * XW;
  }
  f32x4 acc[TAPS][NC];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 px[X_IT], pd[D_IT];
  float sc_x = 1.f, sc_d = 1.f;
  int e_out = 0;
  if constexpr (SPLIT) {
    const int ex = SplitF16::exp_of(*(second ? a.amax_x2 : a.amax_x1) & 0x7FFFFFFFu), ed = SplitF16::exp_of(*a.amax_dy & 0x7FFFFFFFu);
    sc_x = SplitF16::pow2(ex); sc_d = SplitF16::pow2(ed); e_out = -(ex + ed);
  }
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  auto fetch = [&](int tile) {
    int tt = tile;
    const int tx = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty = tt % a.tiles_y; tt /= a.tiles_y;
    const int img = tt;
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const wrsrc_t rx = wmake_rsrc(xsrc + (size_t)img * xpix * cs, (unsigned)(xpix * cs * 4));
    const wrsrc_t rd = wmake_rsrc(dy + (size_t)img * ypix * a.cdy, (unsigned)(ypix * a.cdy * 4));
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int gy = iy0 + x_iy[i], gx = ix0 + x_ix[i];
      const bool ok = gy >= 0 && gy < a.Hx && gx >= 0 && gx < a.Wx;
      const unsigned voff = (ok && kloc + ch4 * 4 < cs) ? (unsigned)(((gy * a.Wx + gx) * cs + kloc + ch4 * 4) * 4) : WSENT;
      px[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int pix = p16 + PPI * i;
      const int gy = oy0 + (pix >> 4), gx = ox0 + (pix & 15);
      const bool ok = gy < a.Hy && gx < a.Wy;
      const unsigned voff = (ok && n0 + ch4 * 4 < a.cdy) ? (unsigned)(((gy * a.Wy + gx) * a.cdy + n0 + ch4 * 4) * 4) : WSENT;
      pd[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)voff, 0, 0);
    }
  };

  int tile = blockIdx.y;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += a.ksplit) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < X_IT; ++i) *reinterpret_cast<u32x4*>(xs + (p16 + PPI * i) * PS + ch4 * 4) = SPLIT ? SplitF16::unit(px[i], sc_x) : px[i];
#pragma unroll
    for (int i = 0; i < D_IT; ++i) *reinterpret_cast<u32x4*>(ds + (p16 + PPI * i) * PS + ch4 * 4) = SPLIT ? SplitF16::unit(pd[i], sc_d) : pd[i];
    __syncthreads();
    if (tile + a.ksplit < ntiles) fetch(tile + a.ksplit);
    constexpr int YR = NARROW ? TH / 2 : TH;
    if constexpr (SPLIT) {
      const unsigned* xw = reinterpret_cast<const unsigned*>(xs);
      const unsigned* dw = reinterpret_cast<const unsigned*>(ds);
#pragma unroll
      for (int yy = 0; yy < YR; ++yy) {
        const int y = ph * YR + yy;
        u32x4 ah[NC], al[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          u32x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = dw[(y * 16 + 4 * j + q) * PS + (nh + c) * 16 + i16];
          ah[c] = w; al[c] = SplitF16::swap_hl(w);  // (H, L) and (L, H) against the (h, l) x fragment: all four products
        }
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            u32x4 b;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = xw[((y * S + kh) * XW + (4 * j + q) * S + kw) * PS + kq * 16 + i16];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[kh * KS + kw][c] = SplitF16::mma_a(ah[c], al[c], b, acc[kh * KS + kw][c]);
          }
      }
      continue;
    }
#pragma unroll
    for (int yy = 0; yy < YR; ++yy) {
      const int y = ph * YR + yy;
#pragma unroll
      for (int xq = 0; xq < 4; ++xq) {
        const int xx = xq * 4 + q;
        float af[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) af[c] = ds[(y * 16 + xx) * PS + (nh + c) * 16 + i16];
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const float b = xs[((y * S + kh) * XW + xx * S + kw) * PS + kq * 16 + i16];
#pragma unroll
            for (int c = 0; c < NC; ++c)
              acc[kh * KS + kw][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c], b, acc[kh * KS + kw][c], 0, 0, 0);
          }
      }
    }
  }
  if constexpr (NARROW) {  // the second pixel half hands its partial sums over through LDS
    static_assert(TAPS * NC * 4 * 128 * 4 <= (int)sizeof(smem), "exchange buffer fits the staging LDS");
    __syncthreads();
    float* ex = smem + (kq * 64 + lane) * (TAPS * NC * 4);
    if (ph == 1) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(ex + (t * NC + c) * 4) = acc[t][c];
    }
    __syncthreads();
    if (ph == 1) return;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(ex + (t * NC + c) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][c][r] += o[r];
      }
  }
  float* slab = a.slabs + (size_t)blockIdx.y * TAPS * a.npad * a.kpad;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (nh + c) * 16 + 4 * q + r, k = k0 + kq * 16 + i16;
        if (kloc + kq * 16 + i16 < cs) slab[((size_t)t * a.npad + n) * a.kpad + k] = SPLIT ? SplitF16::unscale(acc[t][c][r], e_out) : acc[t][c][r];
      }
}

// ---------------------------------------------------------------- slab reduce -> native parameter layout
// layout 0: conv   grad[n][k][kh][kw]  (OIHW, n = Cout, k = Cin)
// layout 1: convT  grad[n][k][kh][kw] where the parameter is [Cin_T][Cout_T][2][2] and the GEMM ran with
//           n := Cin_T (coarse-side channels, "dy" operand) and k := Cout_T (fine-side channels, "x" operand)
// Small gradients (few (n, k) pairs, many split-K slabs): 32 elements x 8 slab lanes per block -- every thread sums an
// eighth of the slabs for one element (elements ordered k-fastest so the slab reads coalesce), LDS combines the lanes in
// a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ slabs, int ksplit, int taps, int npad,
                                                                 int kpad, float* __restrict__ grad, int nn, int kk, int accumulate) {
  __shared__ float sh[8][33];
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int total = nn * kk * taps;
  const int i = blockIdx.x * 32 + e;  // (t, n, k), k fastest
  const size_t sstride = (size_t)taps * npad * kpad;
  float s = 0.f;
  int t = 0, n = 0, k = 0;
  if (i < total) {
    k = i % kk; n = (i / kk) % nn; t = i / (kk * nn);
    const float* src = slabs + ((size_t)t * npad + n) * kpad + k;
    // eight slab loads in flight per thread, added in the same order as a plain loop (latency-bound: 24 us at 512 slabs before)
    int z = zl;
    for (; z + 56 < ksplit; z += 64) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(z + 8 * j) * sstride];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; z < ksplit; z += 8) s += src[(size_t)z * sstride];
  }
  sh[zl][e] = s;
  __syncthreads();
  if (zl == 0 && i < total) {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += sh[j][e];
    float* dst = grad + ((size_t)n * kk + k) * taps + t;
    *dst = accumulate ? *dst + r : r;
  }
}

// The same with four consecutive k per thread (16-byte slab loads; kk % 4 == 0): the 64-channel level's 75 MB of slabs took 52 us with
// 4-byte loads.  Per element the slabs are added in the same order as in wgrad_reduce_small_kernel.
__global__ __launch_bounds__(256) void wgrad_reduce_small4_kernel(const float* __restrict__ slabs, int ksplit, int taps, int npad,
                                                                  int kpad, float* __restrict__ grad, int nn, int kk, int accumulate) {
  __shared__ f32x4 sh[8][33];
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int kq = kk >> 2, total = nn * kq * taps;
  const int i = blockIdx.x * 32 + e;  // (t, n, k / 4), k fastest
  const size_t sstride = (size_t)taps * npad * kpad;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  int t = 0, n = 0, k4 = 0;
  if (i < total) {
    k4 = i % kq; n = (i / kq) % nn; t = i / (kq * nn);
    const float* src = slabs + ((size_t)t * npad + n) * kpad + 4 * k4;
    int z = zl;
    for (; z + 24 < ksplit; z += 32) {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 8 * j) * sstride);
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[j];
    }
    for (; z < ksplit; z += 8) s += *reinterpret_cast<const f32x4*>(src + (size_t)z * sstride);
  }
  sh[zl][e] = s;
  __syncthreads();
  if (zl == 0 && i < total) {
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) r += sh[j][e];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float* dst = grad + ((size_t)n * kk + 4 * k4 + c) * taps + t;
      *dst = accumulate ? *dst + r[c] : r[c];
    }
  }
}

// One thread per (n, k): consecutive lanes walk consecutive k, so every slab read is a coalesced row segment; the taps
// of one (n, k) are adjacent in the parameter layout [n][k][taps], so a wave's stores tile a contiguous span.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int ksplit, int taps, int npad,
                                                           int kpad, float* __restrict__ grad, int nn, int kk, int accumulate) {
  const int total = nn * kk;
  const size_t plane = (size_t)npad * kpad, sstride = (size_t)taps * plane;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / kk, k = i - n * kk;
    const float* src = slabs + (size_t)n * kpad + k;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    for (int z = 0; z < ksplit; ++z) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (t < taps) s[t] += src[z * sstride + t * plane];
    }
    float* dst = grad + (size_t)i * taps;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if (t < taps) dst[t] = accumulate ? dst[t] + s[t] : s[t];
  }
}

// tile heights (rows of 16 output pixels per split-K step)
static int wgrad_tile_h(const MiaOptions& o, int mode, int dtype, int hy, bool fast) {
  const int s = mode == MODE_W3S1 ? 1 : 2;
  if (dtype != MIA_BF16) return s == 1 ? 4 : 2;
  (void)hy;  // 16-row tiles measured slower (more VGPRs, partial unroll): 575 vs 621 TFLOP/s at 64ch 512x512
  if (s == 1 && fast && o.wgrad_dma) return 4;  // wgrad_bf16_dma_kernel
  return s == 1 ? 8 : 4;
}

static bool wgrad_two_wg(const MiaOptions& o, int mode, int dtype) {
  (void)o;
  return mode == MODE_W3S1 && dtype == MIA_BF16;
}

/* split-K workgroups to aim for: one per CU, or two where the kernel is built for two workgroups per CU */
extern "C" int mia_wgrad_target_blocks(int mode, int dtype) {
  const MiaOptions o = mia_options();
  int per_cu = 1;
  if (wgrad_two_wg(o, mode, dtype)) per_cu = 2;
  else if (mode == MODE_W3S2 && dtype == MIA_BF16 && o.wgrad_bt && o.wgrad_dma) per_cu = 2;  // 128 n x 64 k blocks: half as many column blocks
  else if (mode == MODE_W2S2 && dtype == MIA_BF16) per_cu = 2;  // 4 taps: 172 registers, 40 KB LDS -> two workgroups fit a CU
  // option reserve_cus: the split count follows the CUs left to the persistent kernels (a different split count is a
  // different -- still fixed -- fp32 summation order of the slabs: deterministic per setting, not bit-identical across settings)
  const int cus = o.reserve_cus > 0 ? ((256 - o.reserve_cus) & ~7) : 256;
  return per_cu * (cus < 8 ? 8 : cus);
}

#ifdef CONV64_STAMPS
/* Diagnostic build only: workgroups of each persistent weight-gradient kernel the runtime keeps resident on one CU. */
extern "C" int mia_wgrad_debug_occupancy(int which) {
  int n = -1;
  hipError_t e = hipErrorInvalidValue;
  if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_bf16_dma_kernel<false>, 256, 0);
  else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_bf16_bt_kernel, 512, 0);
  else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_bf16_bt_s2_kernel, 512, 0);
  else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_bf16_2wg_kernel<8>, 256, 0);
  return e == hipSuccess ? n : -1;
}
#endif

// 96-wide blocks (wgrad_bf16_dma96_kernel): 3x3 stride 1, bf16, every channel count a multiple of 96 and at least one of them not a
// multiple of 64 (cfg5's level 0: 96 -> 96 and (96 | 96) -> 96).
#ifndef MIA_WGRAD_W96
#define MIA_WGRAD_W96 1  /* 0: probe builds that A/B against the 64-wide blocks */
#endif
static bool wgrad_w96(const MiaOptions& o, int mode, int dtype, int c1, int c2, int cdy) {
  if (!MIA_WGRAD_W96 || dtype != MIA_BF16 || mode != MODE_W3S1 || !o.wgrad_dma) return false;
  if (c1 % 96 != 0 || c2 % 96 != 0 || cdy % 96 != 0) return false;
  return c1 % 64 != 0 || cdy % 64 != 0 || (c2 != 0 && c2 % 64 != 0);
}

/* Column blocks of one split-K slice, the workgroup count to aim for and the tile height of the kernel mia_conv_wgrad will pick for THIS
   shape (the caller sizes ksplit from them: ops.conv_wgrad). */
extern "C" int mia_wgrad_plan(int mode, int dtype, int c1, int c2, int cdy, int npad, int hy, int* column_blocks, int* target_blocks,
                              int* tile_h) {
  const MiaOptions o = mia_options();
  const int cus = o.reserve_cus > 0 ? ((256 - o.reserve_cus) & ~7) : 256;
  if (wgrad_w96(o, mode, dtype, c1, c2, cdy)) {
    if (column_blocks) *column_blocks = ceil_div(cdy, 96) * (ceil_div(c1, 96) + ceil_div(c2, 96));
    if (target_blocks) *target_blocks = cus < 8 ? 8 : cus;  // one 768-thread workgroup per CU
    if (tile_h) *tile_h = 4;
    return MIA_OK;
  }
  if (column_blocks) *column_blocks = (npad / 64) * (ceil_div(c1, 64) + ceil_div(c2, 64));
  if (target_blocks) *target_blocks = mia_wgrad_target_blocks(mode, dtype);
  if (tile_h) *tile_h = wgrad_tile_h(o, mode, dtype, hy, true);
  return MIA_OK;
}

extern "C" int mia_wgrad_geometry(int mode, int dtype, int hy, int wy, int* tiles_y, int* tiles_x) {
  const int th = wgrad_tile_h(mia_options(), mode, dtype, hy, true);  // the finest tiling any kernel of this mode uses (bounds ksplit)
  if (tiles_y) *tiles_y = ceil_div(hy, th);
  if (tiles_x) *tiles_x = ceil_div(wy, 16);
  return MIA_OK;
}

static int conv_wgrad_run(int mode, int dtype, const void* x1, int c1, const void* x2, int c2, const void* dy,
                          int cdy, float* slabs, int ksplit, int npad, int kpad, int n, int hx, int wx, int hy,
                          int wy, void* stream, const float* nl_scale, const float* nl_shift, float nl_slope,
                          const void* amax_x1 = nullptr, const void* amax_x2 = nullptr, const void* amax_dy = nullptr) {
  MIA_CHECK_ARG(mode >= 0 && mode <= MODE_W2S2, "mia_conv_wgrad: bad mode %d", mode);
  MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_conv_wgrad: bad dtype");
  MIA_CHECK_ARG(x1 && dy && slabs && c1 > 0 && c2 >= 0 && cdy > 0, "mia_conv_wgrad: null/empty operand");
  MIA_CHECK_ARG((c2 == 0) == (x2 == nullptr), "mia_conv_wgrad: split operand mismatch");
  MIA_CHECK_ARG(npad % 64 == 0 && kpad % 64 == 0 && npad >= cdy && kpad >= c1 + c2, "mia_conv_wgrad: bad padding");
  MIA_CHECK_ARG(ksplit >= 1 && ksplit <= 65535, "mia_conv_wgrad: bad ksplit %d", ksplit);
  bool ok = mode == MODE_W3S1 ? (hx == hy && wx == wy)
          : mode == MODE_W3S2 ? (hy == (hx + 1) / 2 && wy == (wx + 1) / 2) : (hx == 2 * hy && wx == 2 * wy);
  MIA_CHECK_ARG(ok, "mia_conv_wgrad: mode %d shape mismatch x %dx%d dy %dx%d", mode, hx, wx, hy, wy);
  const MiaOptions o = mia_options();  // one snapshot per call
  WgArgs a;
  a.x1 = x1; a.x2 = x2; a.c1 = c1; a.c2 = c2; a.dy = dy; a.cdy = cdy; a.slabs = slabs;
  a.N = n; a.Hx = hx; a.Wx = wx; a.Hy = hy; a.Wy = wy; a.npad = npad; a.kpad = kpad; a.ksplit = ksplit;
  a.nl_scale = nl_scale; a.nl_shift = nl_shift; a.nl_slope = nl_slope;
  a.amax_x1 = static_cast<const unsigned*>(amax_x1); a.amax_x2 = static_cast<const unsigned*>(amax_x2);
  a.amax_dy = static_cast<const unsigned*>(amax_dy);
  // fp32: split f16 products when the caller knows every operand's maximum (otherwise, or with the option off, exact fp32 MFMAs)
  const bool split = dtype == MIA_F32 && o.f32_split && amax_x1 != nullptr && amax_dy != nullptr && (c2 == 0 || amax_x2 != nullptr);
  const int epu = dtype == MIA_BF16 ? 8 : 4;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec_x = (c1 % epu == 0) && (c2 % epu == 0) && al16(x1) && (x2 == nullptr || al16(x2));
  a.vec_dy = (cdy % epu == 0) && al16(dy);
  a.opt = 1;
  dim3 grid((npad / 64) * (kpad / 64), ksplit);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lim = (size_t)1 << 31;
  // fast paths: channel tails are zero-filled through out-of-range offsets and input-channel blocks are cut per source
  const bool chan_ok = a.vec_x && a.vec_dy;
  dim3 fgrid((npad / 64) * (ceil_div(c1, 64) + ceil_div(c2, 64)), ksplit);
  const bool fast = dtype == MIA_BF16 && chan_ok && (size_t)hx * wx * (c1 > c2 ? c1 : c2) * 2 < lim &&
                    (size_t)hy * wy * cdy * 2 < lim;
  if (fast && o.wgrad_xcd && ksplit % 8 == 0) {  // bf16 fast kernels: 1-D grid in XCD-aware order (a split count below 8 would leave XCDs idle)
    a.opt |= 16;
    fgrid = dim3(fgrid.x * (unsigned)(ceil_div(ksplit, 8) * 8), 1);
  }
  if (nl_scale != nullptr) {  // normalise-on-load: the register-staged two-workgroup kernel is the one that transforms
    if (!(fast && mode == MODE_W3S1 && c2 == 0)) {
      mia_set_error("mia_conv_wgrad_nl: shape outside the normalise-on-load kernel's contract (ask mia_wgrad_nl_supported first)");
      return MIA_EUNSUPPORTED;
    }
    a.tiles_y = ceil_div(hy, 8);
    a.tiles_x = ceil_div(wy, 16);
    hipLaunchKernelGGL((wgrad_bf16_2wg_kernel<8, true>), fgrid, dim3(256), 0, st, a);
    MIA_LAUNCH_CHECK();
    return MIA_OK;
  }
  const int th = wgrad_tile_h(o, mode, dtype, hy, fast);
  a.tiles_y = ceil_div(hy, th);
  a.tiles_x = ceil_div(wy, 16);
  if (fast && wgrad_w96(o, mode, dtype, c1, c2, cdy)) {  // 96-wide ring blocks: one block per pixel tile where 64-wide ones need 2 x 2
    const unsigned ncol = (unsigned)(ceil_div(cdy, 96) * (ceil_div(c1, 96) + ceil_div(c2, 96)));
    const dim3 wgrid = (a.opt & 16) ? dim3(ncol * (unsigned)(ceil_div(ksplit, 8) * 8), 1) : dim3(ncol, ksplit);
    a.tiles_y = ceil_div(hy, 4);
    hipLaunchKernelGGL(wgrad_bf16_dma96_kernel<0>, wgrid, dim3(768), 0, st, a);
  } else if (fast && th == 4 && mode == MODE_W3S1 && o.wgrad_bt && cdy % 128 == 0 && npad % 128 == 0) {
    // 512-thread workgroups on 128 n x 64 k blocks: half as many column blocks
    dim3 bgrid(fgrid.x / 2, fgrid.y);
    hipLaunchKernelGGL(wgrad_bf16_bt_kernel, bgrid, dim3(512), 0, st, a);
  } else if (fast && th == 4 && mode == MODE_W3S2 && o.wgrad_bt && cdy % 128 == 0 && npad % 128 == 0) {
    dim3 bgrid(fgrid.x / 2, fgrid.y);
    hipLaunchKernelGGL(wgrad_bf16_bt_s2_kernel, bgrid, dim3(512), 0, st, a);
  } else if (fast && th == 4 && mode == MODE_W2S2 && o.wgrad_bt >= 1 && o.wgrad_t2 && cdy % 128 == 0 && npad % 128 == 0) {
    dim3 bgrid(fgrid.x / 2, fgrid.y);
    hipLaunchKernelGGL(wgrad_bf16_bt_t2_kernel, bgrid, dim3(512), 0, st, a);
  } else if (fast && th == 4 && mode == MODE_W3S1) {
    // (wgrad_bf16_dma_kernel<true>, the form that skips a narrow block's empty 16-channel tiles, round 4: same sums, no gain --
    // profiles/r04_ab_wgrad_narrow.txt -- and no longer instantiated)
    hipLaunchKernelGGL(wgrad_bf16_dma_kernel<false>, fgrid, dim3(256), 0, st, a);
  } else if (fast && wgrad_two_wg(o, mode, dtype)) {  // stride-2 / transposed shapes stay on the one-workgroup-per-CU kernel
    hipLaunchKernelGGL(wgrad_bf16_2wg_kernel<8>, fgrid, dim3(256), 0, st, a);
  } else if (fast) {
    if (mode == MODE_W3S1 && th == 16) hipLaunchKernelGGL((wgrad_bf16_fast_kernel<MODE_W3S1, 16>), fgrid, dim3(256), 0, st, a);
    else if (mode == MODE_W3S1) hipLaunchKernelGGL((wgrad_bf16_fast_kernel<MODE_W3S1, 8>), fgrid, dim3(256), 0, st, a);
    else if (mode == MODE_W3S2) hipLaunchKernelGGL((wgrad_bf16_fast_kernel<MODE_W3S2, 4, true>), fgrid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad_bf16_fast_kernel<MODE_W2S2, 4>), fgrid, dim3(256), 0, st, a);  // (four taps: 172 registers, two waves per SIMD as it is; the 512-thread form measured 10 % slower)
  } else if (dtype == MIA_F32 && chan_ok && (size_t)hx * wx * (c1 > c2 ? c1 : c2) * 4 < lim &&
             (size_t)hy * wy * cdy * 4 < lim) {
    const bool narrow = cdy <= 32 && c1 <= 32 && c2 <= 32;  // 32-channel layers: half-width blocks, all four waves busy
    if (narrow && mode == MODE_W3S1) a.tiles_y = ceil_div(hy, 8);  // (8-row tiles: wgrad_f32_fast_kernel N8)
    if (split) {  // fp32 tensors, two-part split f16 products
      if (narrow) {
        if (mode == MODE_W3S1) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S1, true, true>), fgrid, dim3(256), 0, st, a);
        else if (mode == MODE_W3S2) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S2, true, true>), fgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W2S2, true, true>), fgrid, dim3(256), 0, st, a);
      } else if (mode == MODE_W3S1) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S1, false, true>), fgrid, dim3(512), 0, st, a);
      else if (mode == MODE_W3S2) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S2, false, true>), fgrid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W2S2, false, true>), fgrid, dim3(512), 0, st, a);
    } else if (narrow) {
      if (mode == MODE_W3S1) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S1, true>), fgrid, dim3(256), 0, st, a);
      else if (mode == MODE_W3S2) hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W3S2, true>), fgrid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((wgrad_f32_fast_kernel<MODE_W2S2, true>), fgrid, dim3(256), 0, st, a);
    } else if (mode == MODE_W3S1) hipLaunchKernelGGL(wgrad_f32_fast_kernel<MODE_W3S1>, fgrid, dim3(512), 0, st, a);
    else if (mode == MODE_W3S2) hipLaunchKernelGGL(wgrad_f32_fast_kernel<MODE_W3S2>, fgrid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(wgrad_f32_fast_kernel<MODE_W2S2>, fgrid, dim3(512), 0, st, a);
  } else if (dtype == MIA_BF16) {
    if (mode == MODE_W3S1) hipLaunchKernelGGL(wgrad_bf16_kernel<MODE_W3S1>, grid, dim3(256), 0, st, a);
    else if (mode == MODE_W3S2) hipLaunchKernelGGL(wgrad_bf16_kernel<MODE_W3S2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_bf16_kernel<MODE_W2S2>, grid, dim3(256), 0, st, a);
  } else {
    if (mode == MODE_W3S1) hipLaunchKernelGGL(wgrad_f32_kernel<MODE_W3S1>, grid, dim3(256), 0, st, a);
    else if (mode == MODE_W3S2) hipLaunchKernelGGL(wgrad_f32_kernel<MODE_W3S2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_f32_kernel<MODE_W2S2>, grid, dim3(256), 0, st, a);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_conv_wgrad(int mode, int dtype, const void* x1, int c1, const void* x2, int c2, const void* dy,
                              int cdy, float* slabs, int ksplit, int npad, int kpad, int n, int hx, int wx, int hy,
                              int wy, const void* amax_x1, const void* amax_x2, const void* amax_dy, void* stream) {
  return conv_wgrad_run(mode, dtype, x1, c1, x2, c2, dy, cdy, slabs, ksplit, npad, kpad, n, hx, wx, hy, wy, stream, nullptr, nullptr, 0.f,
                        amax_x1, amax_x2, amax_dy);
}

// Weight gradient with normalise-on-load of x (the backward half of the fused PlainBlock): see include/mia_hip.h.
extern "C" int mia_wgrad_nl_supported(int mode, int dtype, int c1, int cdy) {
  if (mode != MODE_W3S1) return 0;
  return (dtype == MIA_BF16 && c1 % 8 == 0 && cdy % 8 == 0) ? 1 : 0;
}

extern "C" int mia_conv_wgrad_nl(int mode, int dtype, const void* y_in, int c1, const float* in_scale, const float* in_shift,
                                 float slope, const void* dy, int cdy, float* slabs, int ksplit, int npad, int kpad, int n,
                                 int hx, int wx, int hy, int wy, void* stream) {
  MIA_CHECK_ARG(in_scale != nullptr && in_shift != nullptr, "mia_conv_wgrad_nl: null coefficient arrays");
  MIA_CHECK_ARG(slope >= 0.f && slope <= 1.f, "mia_conv_wgrad_nl: slope %g outside [0, 1]", (double)slope);
  MIA_CHECK_ARG(mia_wgrad_nl_supported(mode, dtype, c1, cdy), "mia_conv_wgrad_nl: unsupported shape (mode %d dtype %d %d x %d)", mode,
                dtype, c1, cdy);
  return conv_wgrad_run(mode, dtype, y_in, c1, nullptr, 0, dy, cdy, slabs, ksplit, npad, kpad, n, hx, wx, hy, wy, stream, in_scale,
                        in_shift, slope);
}

extern "C" int mia_wgrad_reduce(const float* slabs, int ksplit, int taps, int npad, int kpad, float* grad, int nn,
                                int kk, int accumulate, void* stream) {
  MIA_CHECK_ARG(slabs && grad && ksplit >= 1 && taps >= 1 && nn >= 1 && kk >= 1 && nn <= npad && kk <= kpad,
                "mia_wgrad_reduce: bad arguments");
  MIA_CHECK_ARG(taps <= 9 && (int64_t)nn * kk < ((int64_t)1 << 31), "mia_wgrad_reduce: taps > 9 or gradient too large");
  const int64_t total = (int64_t)nn * kk;
  // many slabs: split them over 8 lanes per element (the per-(n,k) kernel below walks all `ksplit` slabs serially, which is
  // latency bound -- 57 us for 75 MB at 512 slabs); few slabs: one thread per (n,k), all taps
  if ((total < 32768 || ksplit >= 8) && total * taps < ((int64_t)1 << 31)) {
    if (kk % 4 == 0 && kpad % 4 == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0) {
      const int blocks = (int)((total / 4 * taps + 31) / 32);
      hipLaunchKernelGGL(wgrad_reduce_small4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), slabs, ksplit,
                         taps, npad, kpad, grad, nn, kk, accumulate);
    } else {
      const int blocks = (int)((total * taps + 31) / 32);
      hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), slabs, ksplit,
                         taps, npad, kpad, grad, nn, kk, accumulate);
    }
  } else {
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), slabs, ksplit,
                       taps, npad, kpad, grad, nn, kk, accumulate);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
