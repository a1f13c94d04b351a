// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels, bf16 -- the "canonical block" launch of the
// benchmark (the full-resolution C0 -> C0 PlainBlock conv, reference src/models/unet/blocks.py:83-90)
// and its input gradient (same kernel, taps flipped).  Persistent, weights held in registers.
//
// Why a second kernel for this one shape: at 64 channels the generic tile kernel (conv_mma_fast.hip) re-stages the whole
// 72 KB weight tensor for every 256-pixel tile -- more bytes through the load / ds_write path than the tile's own input
// (41 KB) -- and spends a third of each tile in an LDS-transposed epilogue.  Here
//   * a workgroup walks MANY tiles; the weights are read from memory ONCE per workgroup: wave w owns output channels
//     16w .. 16w+15 and keeps their 2 x 9 MFMA A-operand fragments (all taps, both 32-channel halves) in 72 VGPRs;
//   * the MFMA operands are swapped (A = weights: M = output channel, B = pixels: N = pixel), so a lane's four accumulator
//     registers are four CONSECUTIVE output channels of one pixel: the epilogue stores 8 bytes per lane straight from the
//     accumulators (each pixel's 32-byte segment per wave), no LDS transpose, and the bias rides in as the first MFMA's C;
//   * only the input tile is staged per tile (12 buffer loads + 12 ds_write_b128 per thread instead of 30 + 30), prefetched
//     into registers behind the previous tile's MFMAs;
//   * per (channel half, horizontal tap) a wave reads each of the 18 input rows once and slides the three vertical taps over
//     it: 108 ds_read_b128 per 288 MFMAs;
//   * measured and rejected here: s_setprio 1 around the matrix phase (0.543 -> 0.552 ms on the canonical launch);
//   * blockIdx -> tile mapping gives each XCD a contiguous run of tiles per step (neighbouring tiles share halo rows in
//     that XCD's L2).
// LDS image of a tile: the layout of conv_mma_fast.hip ([channel-group plane][pixel], plane pitch == 2 (mod 16) units, pixel
// <-> MFMA column permuted by pi16) so fragment reads and staging writes are bank-conflict free.
//
// Contract (checked by conv64_eligible, otherwise conv_mma_fast runs): bf16, one 64-channel source, one or two 64-channel
// destinations (two = the input gradient of a decoder block whose input was [skip | up]: one launch per destination over
// the same input), packed weights [9][64 or 128][64], 16-byte aligned pointers, per-image tensors < 2 GiB.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define SENT 0xFFFFFFF0u /* always beyond num_records */

namespace {

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

constexpr int C = 64;             // channels in == out
constexpr int TH = 16, TW = 16;   // output tile
constexpr int IH = TH + 2, IW = TW + 2, PITCH = IW;
constexpr int A_IT = (IH * IW + 63) / 64;                  // 6 staging iterations of 64 pixels x 4 channel groups (x 2 halves)
constexpr int NPA = ((A_IT * 64 + 13) / 16) * 16 + 2;      // plane pitch in 16-byte units, == 2 (mod 16)
constexpr int LDS_BYTES = 8 * NPA * 16;

struct Tile { int img, ty, tx; };

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair (low = a): ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// s1 += v; s2 += v * v as two single-issue VALU instructions (hipcc otherwise SLP-packs the pair into mov + mul + v_pk_add_f32,
// three instructions of which the packed one costs double beside MFMAs)
__device__ __forceinline__ void stat_acc(float& s1, float& s2, float v) {
  asm volatile("v_add_f32 %0, %2, %0\n\tv_fmac_f32 %1, %2, %2" : "+v"(s1), "+v"(s2) : "v"(v));
}

// sum over the 16 lanes of a DPP row (every lane gets the total): 4 v_add_f32 with DPP operands, no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
  int iv;
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false));   // lane ^ 1
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false));   // lane ^ 2
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false));  // row_half_mirror
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

}  // namespace

// Diagnostic build only (-DCONV64_STAMPS, tools/conv64_stamps.py): per-wave cycle sums of the phases of a tile.  The shipped
// library executes no stamp.
#ifdef CONV64_STAMPS
__device__ unsigned long long conv64_dbg[512 * 4 * 8];
#define STAMP(var)                                                                          \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
extern "C" int mia_conv64_debug_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(conv64_dbg), sizeof(conv64_dbg));
}
#else
#define STAMP(var) do { } while (0)
#endif

// NL = normalise-on-load (the "fused block": reference blocks.py:98-102 of the PRODUCING block folded into this conv): `a.in1`
// holds the producer's raw conv output y and the staged value is bf16(lrelu(scale[n][c] * y + shift[n][c])) -- the same fp32
// fma / select / round-to-nearest-even as norm_act_fwd_stream_kernel, so the conv sees bit for bit the activation the apply
// pass would have written, and that pass (read y + write z) never runs.  The transform sits in commit() (between the two tile
// barriers, beside the co-resident workgroup's MFMAs): per 16-byte unit 2 x 4 unpack + 4 v_pk_fma + 4 v_pk_mul + 8 v_max + 4
// v_cvt_pk; the 16 x 2 coefficients of a thread's channels come from a 512-byte LDS table refreshed per tile by threads
// 0..127 (one extra VGPR across the matrix phase).  Zero padding is padding of z, not of y: halo units outside the image
// are forced back to zero (border tiles only; 6 validity bits per thread).
// CR = column-reduce epilogue (ConvArgs::cr_*): the launch computes an input gradient dz and, while the tile's accumulators are
// still in registers, the per-tile sums of g = dz * lrelu'(scale*y + shift) and g * xhat for the norm backward of the block
// that produced this activation -- from the bf16-rounded dz it is about to store and the y tile it loads here (8 bytes per lane
// and row).  Replaces that block's stand-alone reduction pass (read dz + read y: colreduce_vec_kernel) by one read of y.
template <bool NL, bool CR>
__global__ __launch_bounds__(256, 2) void conv64_persist_kernel(const ConvArgs a, int total_tiles, int tiles_per_img, int run, int n_base) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + (NL ? 2 * C * 4 : 0)];
  u32x4* ldsA = reinterpret_cast<u32x4*>(smem);
  float* cf = reinterpret_cast<float*>(smem + LDS_BYTES);  // NL: [0, 64) scale, [64, 128) shift of the image being committed

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int pr = pi16(c16);          // pixel (within a 16-pixel run) of this lane's MFMA column
  const int g = tid & 3, p4 = tid >> 2;

  // ---- weights: A operand fragments, lane (row = c16 -> output channel 16*wave + c16, k group q), resident for the whole launch
  // n_base: first of this launch's 64 output channels inside a wider packed weight tensor / statistics row (a two-destination
  // input gradient runs as two launches, one per destination)
  const rsrc_t rsw = make_rsrc(a.wp, (unsigned)(9 * a.npad * C * 2));
  u32x4 wf[2][3][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int ta = 0; ta < 3; ++ta)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const int t = ta * 3 + tb;
        const int tw = a.flip ? 8 - t : t;
        wf[c][ta][tb] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((n_base + 16 * wave + c16) * C + 32 * c + 8 * q) * 2, tw * a.npad * C * 2, 0);
      }
  // bias of this lane's four output channels = the first MFMA's C operand
  f32x4 bv;
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[n_base + 16 * wave + 4 * q + r] : 0.f;

  // tile walk: per step every XCD (blocks b, b+8, ... share one) takes a contiguous run of `run` tiles
  const int nblk = gridDim.x, b = blockIdx.x;
  const int first = (b & 7) * run + (b >> 3);
  const int stride = 8 * run;  // == nblk when nblk % 8 == 0 (the host launches multiples of 8)
  auto decode = [&](int t) -> Tile {
    Tile r;
    r.img = t / tiles_per_img;
    const int rem = t - r.img * tiles_per_img;
    r.ty = rem / a.tiles_x;
    r.tx = rem - r.ty * a.tiles_x;
    return r;
  };
  // the walk advances by a constant number of tiles: its (image, row, column) digits once, then carries -- no divisions per tile
  const Tile dstep = decode(stride);
  auto advance = [&](Tile r) -> Tile {
    r.tx += dstep.tx;
    if (r.tx >= a.tiles_x) { r.tx -= a.tiles_x; r.ty += 1; }
    r.ty += dstep.ty;
    if (r.ty >= a.tiles_y) { r.ty -= a.tiles_y; r.img += 1; }
    r.img += dstep.img;
    return r;
  };

  const size_t ipix = (size_t)a.Hin * a.Win;
  const unsigned img_bytes = (unsigned)(ipix * C * 2);
  const bf16_t* in = static_cast<const bf16_t*>(a.in1);
  bf16_t* out = static_cast<bf16_t*>(n_base ? a.out2 : a.out1);

  u32x4 pf[2 * A_IT];
  float cpf = 0.f;            // NL: this thread's entry of the fetched tile's coefficient table (threads 0..127)
  auto fetch = [&](const Tile& t) {
    const rsrc_t rs = make_rsrc(in + (size_t)t.img * ipix * C, img_bytes);
    const int iy0 = t.ty * TH - 1, ix0 = t.tx * TW - 1;
    // staging geometry is recomputed per tile from an opaque copy of the thread's pixel slot: a hoisted per-thread table
    // would be spilled around the tile loop, and its reload's vmcnt(0) would serialise the prefetch behind the stores
    int p4v = p4;
    asm volatile("" : "+v"(p4v));
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= a.Hin && ix0 + IW <= a.Win;  // uniform: no border tests needed
    const int tile_off = (iy0 * a.Win + ix0) * (C * 2) + g * 16;
    if constexpr (NL) {
      if (wave < 2) cpf = (wave == 0 ? a.nl_scale : a.nl_shift)[(size_t)t.img * C + lane];  // wave-uniform pointer: wave 0 = scales, wave 1 = shifts
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int pix = p4v + 64 * i, iy = pix / IW, ix = pix - iy * IW;
      unsigned voff;
      if (interior) {
        voff = (unsigned)(tile_off + (iy * a.Win + ix) * (C * 2));
        if (i == A_IT - 1) voff = (pix < IH * IW) ? voff : SENT;
      } else {
        const int gy = iy0 + iy, gx = ix0 + ix;
        const int okm = -(int)(((unsigned)gy < (unsigned)a.Hin) & ((unsigned)gx < (unsigned)a.Win) & (pix < IH * IW));
        const unsigned off = (unsigned)((gy * a.Win + gx) * (C * 2) + g * 16);
        voff = (off & (unsigned)okm) | (SENT & ~(unsigned)okm);
      }
      pf[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
      pf[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 64, 0);
    }
  };
  // NL: one 16-byte unit (8 bf16 channels: dword d = channels 2d | 2d+1) -> bf16(lrelu(sc * y + sh)), zero if !keep.
  // Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: one issue slot per channel PAIR); per dword 2 unpack + pk_fma + pk_mul +
  // 2 v_max + v_cvt_pk (+ v_and in border tiles).
  auto xform = [&](const u32x4& raw, const f32x4& sa, const f32x4& sb, const f32x4& ha, const f32x4& hb, auto mask_tag, unsigned keep) -> u32x4 {
    constexpr bool MASK = decltype(mask_tag)::value;
    u32x4 o;
    const f32x2_t sl2 = {a.nl_slope, a.nl_slope};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned w = raw[d];
      const f32x2_t sc = d == 0 ? f32x2_t{sa[0], sa[1]} : d == 1 ? f32x2_t{sa[2], sa[3]} : d == 2 ? f32x2_t{sb[0], sb[1]} : f32x2_t{sb[2], sb[3]};
      const f32x2_t sh = d == 0 ? f32x2_t{ha[0], ha[1]} : d == 1 ? f32x2_t{ha[2], ha[3]} : d == 2 ? f32x2_t{hb[0], hb[1]} : f32x2_t{hb[2], hb[3]};
      const f32x2_t x = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
      const f32x2_t v = __builtin_elementwise_fma(sc, x, sh);
      const f32x2_t m = v * sl2;
      const unsigned r = pack_bf16x2(__builtin_fmaxf(v[0], m[0]), __builtin_fmaxf(v[1], m[1]));  // max(v, slope v) == (v > 0 ? v : slope v), 0 <= slope <= 1
      o[d] = MASK ? (r & keep) : r;
    }
    return o;
  };
  auto commit = [&](const Tile& t) {
    if constexpr (NL) {
      const f32x4* cf4 = reinterpret_cast<const f32x4*>(cf);  // channels 8g .. 8g+7 (half 0) and 32 + 8g .. (half 1)
      const int iy0 = t.ty * TH - 1, ix0 = t.tx * TW - 1;
      const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= a.Hin && ix0 + IW <= a.Win;  // uniform
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // one channel half at a time: 16 coefficient registers live, not 32
        const f32x4 sa = cf4[8 * h + 2 * g], sb = cf4[8 * h + 2 * g + 1], ha = cf4[16 + 8 * h + 2 * g], hb = cf4[17 + 8 * h + 2 * g];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) pf[2 * i + h] = xform(pf[2 * i + h], sa, sb, ha, hb, std::false_type{}, 0u);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!interior) {  // border tile: halo units outside the image go back to zero (validity recomputed as in fetch)
        int p4v = p4;
        asm volatile("" : "+v"(p4v));
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int pix = p4v + 64 * i, iy = pix / IW, ix = pix - iy * IW;
          const unsigned keep = 0u - (unsigned)(((unsigned)(iy0 + iy) < (unsigned)a.Hin) & ((unsigned)(ix0 + ix) < (unsigned)a.Win));
#pragma unroll
          for (int d = 0; d < 4; ++d) { pf[2 * i][d] &= keep; pf[2 * i + 1][d] &= keep; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ldsA[g * NPA + p4 + 64 * i] = pf[2 * i];
      ldsA[(4 + g) * NPA + p4 + 64 * i] = pf[2 * i + 1];
    }
  };

  int t = first;
  if (t >= total_tiles) return;  // uniform per workgroup
  Tile cur = decode(t);
  fetch(cur);
  if constexpr (NL) {
    if (tid < 2 * C) cf[tid] = cpf;
    __syncthreads();
  }
  commit(cur);
  __syncthreads();

#ifdef CONV64_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, acc_f = 0, acc_m = 0, acc_e = 0, acc_b = 0, acc_c = 0, ntl = 0;
#endif
  while (true) {
    STAMP(ts0);
    const int tn = t + stride;
    const bool more = tn < total_tiles;  // uniform
    Tile nxt = cur;
    if (more) { nxt = advance(cur); fetch(nxt); }

    STAMP(ts1);
    // ---- 288 MFMAs: 2 channel halves x 3 horizontal taps x 18 input rows, each row feeding up to three output rows
    f32x4 acc[TH];
    constexpr int GROUPS = 6, RPG = 3;  // rows per prefetch group
    u32x4 fr[2][RPG];
    auto load_group = [&](int gi, u32x4* f) {  // gi in [0, 36): phase = gi / 6 -> (c, tb), rows 3*(gi%6) .. +2
      const int ph = gi / GROUPS, c = ph / 3, tb = ph % 3, r0 = (gi % GROUPS) * RPG;
#pragma unroll
      for (int j = 0; j < RPG; ++j) f[j] = ldsA[(4 * c + q) * NPA + (r0 + j) * PITCH + tb + pr];
    };
    load_group(0, fr[0]);
#pragma unroll
    for (int gi = 0; gi < 6 * GROUPS; ++gi) {
      const int ph = gi / GROUPS, c = ph / 3, tb = ph % 3, r0 = (gi % GROUPS) * RPG;
      if (gi + 1 < 6 * GROUPS) load_group(gi + 1, fr[(gi + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the next group's reads ahead of this group's MFMAs
#pragma unroll
      for (int j = 0; j < RPG; ++j) {
        const int r = r0 + j;
#pragma unroll
        for (int ta = 0; ta < 3; ++ta) {
          const int m = r - ta;
          if (m >= 0 && m < TH) {
            const bool first_touch = (ph == 0 && ta == 0);  // (c, tb) = (0, 0), ta = 0 is the first product into acc[m]
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][ta][tb]),
                                                             __builtin_bit_cast(bf16x8, fr[gi & 1][j]), first_touch ? bv : acc[m], 0, 0, 0);
          }
        }
      }
    }

    STAMP(ts2);
    // ---- epilogue: statistics + 8-byte stores straight from the accumulators
    const int oy0 = cur.ty * TH, ox0 = cur.tx * TW;
    const bool full = (oy0 + TH <= a.Hout) && (ox0 + TW <= a.Wout);  // uniform
    const bool colok = ox0 + pr < a.Wout;
    const rsrc_t rso = make_rsrc(out + (size_t)cur.img * ipix * C, img_bytes);
    // Two output rows per store: v_permlane16_swap trades the 8-byte halves between the lane pairs (q, q^1) that hold the
    // same pixel column, so an even-q lane ends up with 8 consecutive channels (16 bytes) of row m and its odd-q partner
    // with the same 8 channels of row m+1 -- 8 dwordx4 stores per tile instead of 16 dwordx2 (the tail is store-ISSUE bound).
    const int qodd = q & 1;
    const unsigned obase = (unsigned)((((oy0 + qodd) * a.Wout) + ox0 + pr) * (C * 2) + (16 * wave + 8 * (q >> 1)) * 2);
    const int row_bytes = a.Wout * (C * 2);
    const bool want_stats = a.stats != nullptr;  // uniform
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // CR: this lane's four channels of the producing block's coefficient rows, and the y tile (row m: 8 bytes per lane,
    // fetched in two batches of eight rows so that at most 28 registers hold them)
    f32x4 csc, csf, cxa, cxb;
    u32x2 yv[TH];
    const unsigned ybase = (unsigned)(((oy0 * a.Wout) + ox0 + pr) * (C * 2) + (16 * wave + 4 * q) * 2);
    rsrc_t rsy;
    constexpr int YB = 2;  // rows per y batch
    auto load_y = [&](int m0, bool full_t) {
#pragma unroll
      for (int mm = m0; mm < m0 + YB; ++mm) {
        const bool ok = full_t || (colok && oy0 + mm < a.Hout);
        yv[mm] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsy, (int)(ok ? ybase + (unsigned)(mm * row_bytes) : SENT), 0, 0));
      }
    };
    if constexpr (CR) {
      const unsigned cbytes = (unsigned)(a.N * C * 4);
      const int co = (cur.img * C + 16 * wave + 4 * q) * 4;  // scalar descriptors + one 32-bit lane offset: no 64-bit lane pointers
      csc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.cr_scale, cbytes), co, 0, 0));
      csf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.cr_shift, cbytes), co, 0, 0));
      cxa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.cr_xa, cbytes), co, 0, 0));
      cxb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.cr_xb, cbytes), co, 0, 0));
      rsy = make_rsrc(static_cast<const bf16_t*>(a.cr_y) + (size_t)cur.img * ipix * C, img_bytes);
      load_y(0, full);
      load_y(YB, full);
    }
    // CR: one row of this lane (4 channels: packed bf16 pairs p01, p23 = the dz about to be stored; y pairs from yv)
    auto cr_row = [&](unsigned p01, unsigned p23, const u32x2& yr, float wgt) {
      const unsigned dzw[2] = {p01, p23}, yw[2] = {yr[0], yr[1]};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 2 * h + e;
          const float dzv = __builtin_bit_cast(float, e ? (dzw[h] & 0xFFFF0000u) : (dzw[h] << 16));
          const float y = __builtin_bit_cast(float, e ? (yw[h] & 0xFFFF0000u) : (yw[h] << 16));
          float g = dzv * wgt;
          if (!(__builtin_fmaf(csc[r], y, csf[r]) > 0.f)) g *= a.cr_slope;
          s1[r] += g;
          s2[r] = __builtin_fmaf(g, __builtin_fmaf(cxa[r], y, cxb[r]), s2[r]);
        }
      }
    };
    auto store_rows = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      // the inline-asm statistics below read MFMA results; hipcc pads hazards only for instructions it emits itself, so the
      // last accumulators written (rows 15, 14, 13) get their 12+ wait states here
      if constexpr (FULL) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int m = 0; m < TH; m += 2) {
        if constexpr (CR) {  // y rows two batches (eight rows) ahead of their use
          if (m % YB == 0 && m + 2 * YB < TH) load_y(m + 2 * YB, FULL);
        }
        if (want_stats && !CR) {
#pragma unroll
          for (int mm = m; mm < m + 2; ++mm) {
            const float w = (FULL || (colok && oy0 + mm < a.Hout)) ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = acc[mm][r];
              if constexpr (FULL) stat_acc(s1[r], s2[r], v);
              else { const float vm = v * w; s1[r] += vm; s2[r] += vm * v; }
            }
          }
        }
        // packed bf16 pairs: x = row m, y = row m+1 (this lane's 4 channels each)
        const unsigned x0 = pack_bf16x2(acc[m][0], acc[m][1]), x1 = pack_bf16x2(acc[m][2], acc[m][3]);
        const unsigned y0 = pack_bf16x2(acc[m + 1][0], acc[m + 1][1]), y1 = pack_bf16x2(acc[m + 1][2], acc[m + 1][3]);
        if constexpr (CR) {
          cr_row(x0, x1, yv[m], (FULL || (colok && oy0 + m < a.Hout)) ? 1.f : 0.f);
          cr_row(y0, y1, yv[m + 1], (FULL || (colok && oy0 + m + 1 < a.Hout)) ? 1.f : 0.f);
        }
        // vdst rows 1,3 (odd q) <-> src rows 0,2 (even q): even q gets its partner's x in y, odd q its partner's y in x
        const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        const u32x4 d = {r0[0], r1[0], r0[1], r1[1]};
        if constexpr (FULL) {
          __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)obase, m * row_bytes, 0);  // row offset rides in the scalar offset
        } else {
          const bool ok = colok && (oy0 + m + qodd < a.Hout);
          const unsigned voff = ok ? obase + (unsigned)(m * row_bytes) : SENT;
          __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)voff, 0, 0);
        }
        if constexpr (CR) {
          // Store-data hazard (found on hardware, tools/probe/diag_cr3.py): under the register pressure of this epilogue hipcc
          // reuses a store's data registers three instructions after the store (a y-row offset went into dword 0), and the
          // 16-byte store reads its data in four lane phases AFTER issue -- lanes {12-15, 28-31, 44-47, 60-63} of ~5 % of the
          // tiles then stored the new value.  hipcc's own rule (one wait state after a >= 12-byte store) does not cover it
          // while the vector-memory path is busy; 16 wait states do (8 stores per tile: ~130 cycles of an ~8000-cycle tile).
          // (-DCR_PAD=0 / 2 / 4 / 8 builds of this line are the measurement: tools/r5_store_hazard.sh -> profiles/r05_store_hazard.txt)
          __builtin_amdgcn_sched_barrier(0);
#ifndef CR_PAD
#define CR_PAD 16
#endif
#if CR_PAD >= 16
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#elif CR_PAD >= 8
          asm volatile("s_nop 7" ::: "memory");
#elif CR_PAD >= 4
          asm volatile("s_nop 3" ::: "memory");
#elif CR_PAD >= 2
          asm volatile("s_nop 1" ::: "memory");
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (full) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    if (want_stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
      if (c16 == 0) {
        const size_t tile = (size_t)cur.img * tiles_per_img + (size_t)cur.ty * a.tiles_x + cur.tx;
        typedef __attribute__((address_space(1))) f32x4 gf32x4;  // global (not flat) store: see conv_mma_fast.hip
        gf32x4* dst = (gf32x4*)(a.stats + (tile * (size_t)(a.o1 + a.o2) + n_base + 16 * wave + 4 * q) * 2);
        dst[0] = f32x4{s1[0], s2[0], s1[1], s2[1]};
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dst[1] = f32x4{s1[2], s2[2], s1[3], s2[3]};
        // store-data hazard (tools/check_store_hazard.py): hipcc re-used these data registers for the next tile's sums at its own
        // minimum distance, the distance that failed in the column-reduce epilogue above; two more wait states were always enough
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(ts3);
#ifdef CONV64_STAMPS
    acc_f += ts1 - ts0; acc_m += ts2 - ts1; acc_e += ts3 - ts2; ntl += 1;
#endif
    if (!more) break;
    if constexpr (NL) {  // nobody reads the table outside commit(): safe to refresh it in front of the barrier
      if (tid < 2 * C) cf[tid] = cpf;
    }
    __syncthreads();  // every wave has read the current image out of LDS
    STAMP(ts4);
    commit(nxt);
    __syncthreads();
    STAMP(ts5);
#ifdef CONV64_STAMPS
    acc_b += ts4 - ts3; acc_c += ts5 - ts4;
#endif
    t = tn;
    cur = nxt;
  }
#ifdef CONV64_STAMPS
  if (lane == 0) {
    unsigned long long* d = conv64_dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    d[0] = acc_f; d[1] = acc_m; d[2] = acc_e; d[3] = acc_b; d[4] = acc_c; d[5] = ntl;
  }
#endif
}

bool conv64_eligible(int mode, int dtype, const ConvArgs& a) {
  if (mode != MODE_G3S1 || dtype != MIA_BF16) return false;
  if (a.nl_scale != nullptr && (a.o2 != 0 || a.nl_shift == nullptr || !(a.nl_slope >= 0.f && a.nl_slope <= 1.f))) return false;
  if (a.cr_y != nullptr && (a.o2 != 0 || a.nl_scale != nullptr || a.stats == nullptr || !a.cr_scale || !a.cr_shift || !a.cr_xa || !a.cr_xb)) return false;
  if (a.c1 != C || a.c2 != 0 || a.o1 != C || (a.o2 != 0 && a.o2 != C) || a.npad != a.o1 + a.o2 || a.kpad != C) return false;
  if (!a.vec_in || !a.vec_out) return false;
  if ((size_t)a.Hin * a.Win * C * 2 >= ((size_t)1 << 31)) return false;
  if (a.Hout <= 8) return false;  // the statistics layout of small maps uses 8-row tiles (mia_conv_mma_tiles)
  return true;
}

int conv64_launch(const ConvArgs& a, int blocks_override, int reserve, hipStream_t st) {
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int total = a.N * tiles_per_img;
  // two workgroups per CU (256 CUs); a multiple of 8 so the per-XCD runs tile the step exactly
  const int cap = 2 * persistent_cus(256, reserve);  // two workgroups per CU
  int nblk = total < cap ? ((total + 7) / 8) * 8 : cap;
  if (blocks_override >= 8 && blocks_override % 8 == 0 && blocks_override <= nblk) nblk = blocks_override;  // option conv64_blocks (diagnostics)
  const int run = nblk / 8;
  if (a.nl_scale != nullptr) {
    hipLaunchKernelGGL((conv64_persist_kernel<true, false>), dim3(nblk), dim3(256), 0, st, a, total, tiles_per_img, run, 0);
    return MIA_OK;
  }
#ifdef MIA_EXPERIMENTS  // the column-reduce epilogue (round 4: correct, +-0 in the step; the store-data hazard was found in it -- probe builds keep it)
  if (a.cr_y != nullptr) {
    hipLaunchKernelGGL((conv64_persist_kernel<false, true>), dim3(nblk), dim3(256), 0, st, a, total, tiles_per_img, run, 0);
    return MIA_OK;
  }
#else
  if (a.cr_y != nullptr) { mia_set_error("column-reduce epilogue: experiment build only (-DMIA_EXPERIMENTS)"); return MIA_EUNSUPPORTED; }
#endif
  hipLaunchKernelGGL((conv64_persist_kernel<false, false>), dim3(nblk), dim3(256), 0, st, a, total, tiles_per_img, run, 0);
  if (a.o2 == C)  // second destination (e.g. the up-sampled half of a decoder block's input gradient): same input, next 64 filters
    hipLaunchKernelGGL((conv64_persist_kernel<false, false>), dim3(nblk), dim3(256), 0, st, a, total, tiles_per_img, run, C);
  return MIA_OK;
}
