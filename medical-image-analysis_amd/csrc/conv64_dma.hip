// 3x3 / stride 1 / pad 1 convolution with 64 input channels and 64 (or 64 | 64) output channels, bf16 -- second generation of
// the "canonical block" kernel (reference src/models/unet/blocks.py:83-90 at channels_list[0], and its
// input gradient).  Same idea as conv64.hip (persistent, weights in registers, operands swapped so the tile is stored straight
// from the accumulators), rebuilt on what round 3 learned from conv_bt.hip:
//   * ONE 512-thread workgroup per CU; the input tile (18 x 18 halo pixels x 128 B) arrives by LDS-DMA into a ring of THREE
//     tile images (`buffer_load_dwordx4 ... offen lds`, issued between the MFMA groups of the current tile and counted by hand):
//     no staging registers, no ds_write pass, ONE barrier per tile instead of two, two tiles of loads in flight per CU;
//   * waves = 2 pixel halves (8 output rows each) x 4 channel groups of 16 (NOUT = 64), or 8 channel groups over all 16 rows
//     (NOUT = 128: the two-destination input gradient of the decoder's first conv in ONE pass over the input instead of two
//     launches); a wave keeps its 2 x 9 weight fragments in 72 registers for the whole launch;
//   * per (channel half, horizontal tap) a wave reads each of its MR + 2 input rows once and slides the three vertical taps over
//     it: 60 ds_read_b128 per 144 MFMAs (MR = 8); no barrier inside a tile (the whole K = 64 x 9 is resident);
//   * the statistics of a tile are parked in LDS and combined behind the NEXT tile's barrier (no extra barrier).
// LDS image of a tile: linear halo pixels p = row * 18 + col, 128 B each; the 16-byte slot of k-chunk ch (0..7) of a pixel is
// ch ^ ((col >> 1) & 7), MFMA column n <-> pixel pi16(n): ds_read_b128 is bank-conflict free for every tap (simulated per 16-lane
// read group).  The DMA destination is lane-linear (8 pixels x 128 B per piece), so the swizzle sits on each lane's SOURCE
// address.
//
// Contract (conv64_dma_eligible, otherwise conv64 / conv_mma_fast run): bf16, MODE_G3S1, c1 = 64, c2 = 0, o1 = 64, o2 in {0, 64},
// packed weights [9][npad = o1 + o2][64], Hout > 8, 16-byte aligned pointers, per-image tensors < 2 GiB.
#include "conv_common.h"
#include <type_traits>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define CD_SENT 0xFFFFFFF0u

__device__ __forceinline__ i32x4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long addr = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)addr);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(addr >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// One LDS-DMA piece (64 lanes x 16 B -> lds_dst + 16 L); see conv_bt.hip::dma16.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row
  int iv;
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false));
  return v;
}

constexpr int C = 64, TH = 16, TW = 16, IH = TH + 2, IW = TW + 2;
constexpr int NPIX = IH * IW;            // 324 halo pixels
constexpr int PIECES = (NPIX + 7) / 8;   // 41 pieces of 8 pixels x 128 B
constexpr int STAGE = PIECES * 1024;     // 41984
constexpr int NSTAGE = 3;
constexpr int PPW = (PIECES + 7) / 8;    // 6: pieces of wave 0 per tile (the other waves: 5)
constexpr int ROWB = IW * 128;           // 2304 bytes per halo row
constexpr int RED = NSTAGE * STAGE;      // statistics exchange: [2 tile parities][8 waves][16 channels][2]
constexpr int LDS_BYTES = RED + 2 * 8 * 16 * 2 * 4;

struct Tile { int img, ty, tx; };

}  // namespace

#ifdef CONV64_STAMPS
__device__ unsigned long long conv64_dma_dbg[256 * 8 * 8];
#define DSTAMP(var)                                                                        \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define DACC(dst, t1, t0) dst += (t1) - (t0)
extern "C" int mia_conv64_dma_debug_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(conv64_dma_dbg), sizeof(conv64_dma_dbg));
}
#else
#define DSTAMP(var) do { } while (0)
#define DACC(dst, t1, t0) do { } while (0)
#endif

template <int NOUT>
__global__ __launch_bounds__(512, 2) void conv64_dma_kernel(const ConvArgs a, int total_tiles, int tiles_per_img, int run) {
  constexpr int MR = NOUT == 64 ? 8 : 16;   // output rows per wave
  constexpr int NST = MR / 2;               // 16-byte store instructions per wave and tile
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int pr = pi16(c16);
  const int cg = NOUT == 64 ? (wave & 3) : wave;   // 16-channel group of this wave
  const int ph = NOUT == 64 ? (wave >> 2) : 0;     // pixel half (rows 8 ph ..)

  // ---- weights: A operand fragments, lane (row c16 -> output channel 16 cg + c16, k group q), resident for the whole launch
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
        wf[c][ta][tb] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((16 * cg + c16) * C + 32 * c + 8 * q) * 2, tw * a.npad * C * 2, 0);
      }
  f32x4 bv;
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[16 * cg + 4 * q + r] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no compiler-visible load stays outstanding inside the counted DMA pipeline

  // ---- tile walk: per step every XCD (blocks b, b + 8, ... share one) takes a contiguous run of `run` tiles
  const int b = blockIdx.x;
  const int first = (b & 7) * run + (b >> 3);
  const int stride = 8 * run;
  auto decode = [&](int t) __attribute__((always_inline)) -> Tile {
    Tile r;
    r.img = t / tiles_per_img;
    const int rem = t - r.img * tiles_per_img;
    r.ty = rem / a.tiles_x;
    r.tx = rem - r.ty * a.tiles_x;
    return r;
  };
  const size_t ipix = (size_t)a.Hin * a.Win;
  const unsigned img_bytes = (unsigned)(ipix * C * 2);
  const bf16_t* in = static_cast<const bf16_t*>(a.in1);

  // ---- DMA lane constants.  Piece k = wave + 8 j covers halo pixels 8 k .. 8 k + 7; lane L = (pixel L >> 3, slot L & 7)
  unsigned toff[PPW];   // offset from the tile's first halo pixel (interior tiles), CD_SENT for padding pixels
  unsigned rc[PPW];     // row | col << 8 of this lane's pixel (border tiles)
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int p = 8 * (wave + 8 * j) + (lane >> 3);
    const int row = p / IW, col = p - row * IW;
    const int chunk = (lane & 7) ^ ((col >> 1) & 7);
    rc[j] = (unsigned)(row | (col << 8) | (chunk << 16) | ((p < NPIX ? 1 : 0) << 24));
    toff[j] = p < NPIX ? (unsigned)(((row * a.Win + col) * C + chunk * 8) * 2) : CD_SENT;
  }
  auto issue_piece = [&](const Tile& t, unsigned stage_base, auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int k = wave + 8 * j;
    if (k >= PIECES) return;  // uniform
    const int iy0 = t.ty * TH - 1, ix0 = t.tx * TW - 1;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= a.Hin && ix0 + IW <= a.Win;  // uniform
    const unsigned dst = __builtin_amdgcn_readfirstlane(stage_base + k * 1024);
    if (interior) {
      const i32x4 rs = rsrc_words(in + ((size_t)t.img * ipix + (size_t)iy0 * a.Win + ix0) * C, (unsigned)(IH * a.Win * C * 2));
      dma16(rs, toff[j], dst);
    } else {
      const i32x4 rs = rsrc_words(in + (size_t)t.img * ipix * C, img_bytes);
      const int row = rc[j] & 0xFF, col = (rc[j] >> 8) & 0xFF, chunk = (rc[j] >> 16) & 0xFF;
      const int gy = iy0 + row, gx = ix0 + col;
      const bool ok = ((rc[j] >> 24) != 0) & ((unsigned)gy < (unsigned)a.Hin) & ((unsigned)gx < (unsigned)a.Win);
      dma16(rs, ok ? (unsigned)(((gy * a.Win + gx) * C + chunk * 8) * 2) : CD_SENT, dst);
    }
  };
#define CD_PIECE(T, S, J) issue_piece(T, S, std::integral_constant<int, J>{})
  auto issue_all = [&](const Tile& t, unsigned sb) __attribute__((always_inline)) {
    CD_PIECE(t, sb, 0); CD_PIECE(t, sb, 1); CD_PIECE(t, sb, 2); CD_PIECE(t, sb, 3); CD_PIECE(t, sb, 4); CD_PIECE(t, sb, 5);
  };
  const bool big = wave < (PIECES & 7);  // waves owning PPW pieces per tile (wave 0); the others PPW - 1
  // all but the newest `ntiles_in_flight` tiles' own pieces (+ extra younger operations) have landed
#define CD_WAIT(N) do { if (big) wait_vm<(N)>(); else wait_vm<((N) - 1)>(); } while (0)

  // ---- fragment read bases: halo pixel (8 ph + r, pr + tb), k-chunk 4 c + q
  unsigned fb[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
      const int col = pr + tb;
      fb[c][tb] = (unsigned)(((8 * ph) * IW + col) * 128 + (((4 * c + q) ^ ((col >> 1) & 7)) * 16));
    }
  bf16_t* out = static_cast<bf16_t*>((NOUT == 128 && wave >= 4) ? a.out2 : a.out1);
  const int ch_out = (NOUT == 128 && wave >= 4) ? 16 * (cg - 4) : 16 * cg;  // first channel of this wave inside its destination
  const bool want_stats = a.stats != nullptr;  // uniform
  float* red = reinterpret_cast<float*>(smem + RED);

  int t = first;
  if (t >= total_tiles) return;  // uniform per workgroup
  Tile cur = decode(t);
  // prologue: two tiles in flight
  issue_all(cur, lds0);
  Tile nx1 = cur;
  const bool has1 = t + stride < total_tiles;
  if (has1) { nx1 = decode(t + stride); issue_all(nx1, lds0 + STAGE); }
  if (has1) CD_WAIT(PPW); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // ---- epilogue pieces of ONE tile (its accumulators `ac`): statistics rows, parking, stores.  NOUT = 64 runs them on the
  // PREVIOUS tile's accumulators between the MFMA groups of the current tile (PIPE): with one workgroup per CU nothing else would
  // hide them -- eight waves finishing a 4.6k-cycle tile together spent ~1.7k cycles in statistics + stores with the matrix pipe
  // idle (0.69 ms against 0.59 for the two-workgroup kernel).  NOUT = 128 has no registers for a second accumulator set.
  constexpr bool PIPE = false && NOUT == 64;  // measured: +2 % only (the compute phase itself was the problem), costs 32 registers
  struct Geo { int oy0, ox0; bool full, colok; };
  auto geo_of = [&](const Tile& tl) __attribute__((always_inline)) -> Geo {
    Geo g;
    g.oy0 = tl.ty * TH + 8 * ph; g.ox0 = tl.tx * TW;
    g.full = (tl.ty * TH + TH <= a.Hout) && (g.ox0 + TW <= a.Wout);  // uniform
    g.colok = g.ox0 + pr < a.Wout;
    return g;
  };
  auto stats_rows = [&](const f32x4* ac, const Geo& g, int m0, int m1, float* s1, float* s2) __attribute__((always_inline)) {
#pragma unroll
    for (int m = m0; m < m1; ++m) {
      const float w = (g.full || (g.colok && g.oy0 + m < a.Hout)) ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = ac[m][r], vm = g.full ? v : v * w; s1[r] += vm; s2[r] += vm * v; }
    }
  };
  auto stats_park = [&](float* s1, float* s2, int parity) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
    if (c16 == 0) {
      float* dst = red + parity * (8 * 16 * 2) + (wave * 16 + 4 * q) * 2;
      *reinterpret_cast<f32x4*>(dst) = f32x4{s1[0], s2[0], s1[1], s2[1]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{s1[2], s2[2], s1[3], s2[3]};
    }
  };
  auto store_rows = [&](const f32x4* ac, const Tile& tl, const Geo& g, int m0, int m1) __attribute__((always_inline)) {
    const rsrc_t rso = make_rsrc(out + (size_t)tl.img * ipix * C, img_bytes);
    const int qodd = q & 1;
    const unsigned obase = (unsigned)((((g.oy0 + qodd) * a.Wout) + g.ox0 + pr) * (C * 2) + (ch_out + 8 * (q >> 1)) * 2);
    const int row_bytes = a.Wout * (C * 2);
#pragma unroll
    for (int m = m0; m < m1; m += 2) {
      const unsigned x0 = pack_bf16x2(ac[m][0], ac[m][1]), x1 = pack_bf16x2(ac[m][2], ac[m][3]);
      const unsigned y0 = pack_bf16x2(ac[m + 1][0], ac[m + 1][1]), y1 = pack_bf16x2(ac[m + 1][2], ac[m + 1][3]);
      // even q gets its partner's row-m half (8 consecutive channels of row m), odd q the same 8 channels of row m + 1
      const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      const u32x4 d = {r0[0], r1[0], r0[1], r1[1]};
      const bool ok = g.full || (g.colok && (g.oy0 + m + qodd < a.Hout));
      const unsigned voff = ok ? obase + (unsigned)(m * row_bytes) : CD_SENT;
      __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)voff, 0, 0);
    }
  };
  // statistics of a tile whose sums every wave has parked (behind a barrier): sum the pixel halves, write the partial
  auto combine = [&](const Tile& tl, int parity) __attribute__((always_inline)) {
    if (tid < NOUT) {
      const float* rp = red + parity * (8 * 16 * 2);
      const int ch = tid, cgc = ch >> 4, cl = ch & 15;
      float s1, s2;
      if (NOUT == 64) {  // two pixel halves per channel group
        const f32x2_t v0 = *reinterpret_cast<const f32x2_t*>(rp + (cgc * 16 + cl) * 2);
        const f32x2_t v1 = *reinterpret_cast<const f32x2_t*>(rp + ((4 + cgc) * 16 + cl) * 2);
        s1 = v0[0] + v1[0]; s2 = v0[1] + v1[1];
      } else {
        const f32x2_t v0 = *reinterpret_cast<const f32x2_t*>(rp + (cgc * 16 + cl) * 2);
        s1 = v0[0]; s2 = v0[1];
      }
      const size_t tile = (size_t)tl.img * tiles_per_img + (size_t)tl.ty * a.tiles_x + tl.tx;
      typedef __attribute__((address_space(1))) f32x2_t gf32x2;  // global (not flat) store: see conv_mma_fast.hip
      *(gf32x2*)(a.stats + (tile * (size_t)(a.o1 + a.o2) + ch) * 2) = f32x2_t{s1, s2};
    }
  };

  int stage = 0, it = 0;
  f32x4 pacc[PIPE ? MR : 1];   // PIPE: the previous tile's accumulators, drained between this tile's MFMA groups
  Tile ptile = cur;            // PIPE: that tile;  !PIPE: unused
  bool have_prev = false;      // PIPE: pacc holds a tile whose epilogue has not run
  Tile ctile = cur;            // tile whose statistics are parked and not yet combined
  int cparity = 0;
  bool pending = false;
#ifdef CONV64_STAMPS
  unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, d_head = 0, d_cmp = 0, d_tail = 0, d_wait = 0, d_bar = 0, d_n = 0;
#endif
  while (true) {
    DSTAMP(u0);
    const int t2 = t + 2 * stride;
    const bool more2 = t2 < total_tiles;            // a tile to issue now (two ahead)
    const bool more1 = t + stride < total_tiles;    // a tile already in flight
    Tile nx2 = cur;
    if (more2) nx2 = decode(t2);
    const unsigned sb2 = lds0 + ((stage + 2) % NSTAGE) * STAGE;

    // parked statistics: every wave parked them before the barrier that ended the previous iteration
    if (pending) { combine(ctile, cparity); pending = false; }
    const Geo pg = geo_of(ptile);
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- 6 x (MR + 2) fragment reads, 18 MR MFMAs; the pieces of the tile two ahead are issued between the groups.  A row feeds
    // only 3 MFMAs (48 cycles), so the reads run as ONE stream over the 6 groups, FD rows ahead of their MFMAs (ring of FD + 1)
    f32x4 acc[MR];
    const unsigned so = stage * STAGE;
    constexpr int NR = MR + 2, NRD = 6 * NR, FD = (NOUT == 64 ? 10 : 5);
    unsigned base[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) base[g] = fb[g / 3][g % 3] + so;
    asm volatile("" : "+v"(base[0]), "+v"(base[1]), "+v"(base[2]), "+v"(base[3]), "+v"(base[4]), "+v"(base[5]));  // opaque: row steps fold into offsets
    DSTAMP(u1); DACC(d_head, u1, u0);
    u32x4 fr[FD + 1];
    auto rd = [&](int idx) __attribute__((always_inline)) {
      fr[idx % (FD + 1)] = *reinterpret_cast<const u32x4*>(smem + base[idx / NR] + (idx % NR) * ROWB);
    };
#pragma unroll
    for (int i = 0; i < FD; ++i) rd(i);
#pragma unroll
    for (int ph6 = 0; ph6 < 6; ++ph6) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int idx = ph6 * NR + r, c = ph6 / 3, tb = ph6 % 3;
        if (idx + FD < NRD) rd(idx + FD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < 3; ++ta) {
          const int m = r - ta;
          if (m >= 0 && m < MR) {
            const bool first_touch = (ph6 == 0 && ta == 0);  // the first product into acc[m]: C operand = bias
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][ta][tb]), __builtin_bit_cast(bf16x8, fr[idx % (FD + 1)]),
                                                             first_touch ? bv : acc[m], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more2) {  // one piece of the tile two ahead behind each of the six groups
        if (ph6 == 0) CD_PIECE(nx2, sb2, 0);
        if (ph6 == 1) CD_PIECE(nx2, sb2, 1);
        if (ph6 == 2) CD_PIECE(nx2, sb2, 2);
        if (ph6 == 3) CD_PIECE(nx2, sb2, 3);
        if (ph6 == 4) CD_PIECE(nx2, sb2, 4);
        if (ph6 == 5) CD_PIECE(nx2, sb2, 5);
      }
      if (PIPE && have_prev) {  // the previous tile's epilogue, a slice behind each group
        if (ph6 == 0 && want_stats) stats_rows(pacc, pg, 0, MR / 2, ps1, ps2);
        if (ph6 == 1 && want_stats) { stats_rows(pacc, pg, MR / 2, MR, ps1, ps2); stats_park(ps1, ps2, (it + 1) & 1); }
        if (ph6 == 2) store_rows(pacc, ptile, pg, 0, MR / 2);
        if (ph6 == 3) store_rows(pacc, ptile, pg, MR / 2, MR);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    DSTAMP(u2); DACC(d_cmp, u2, u1);
    int nst = 0;  // stores issued in this iteration (younger than or interleaved with this iteration's pieces)
    if constexpr (PIPE) {
      if (have_prev) {
        nst = NST;
        if (want_stats) { pending = true; ctile = ptile; cparity = (it + 1) & 1; }  // parked during this tile: combine after its barrier
      }
#pragma unroll
      for (int m = 0; m < MR; ++m) pacc[m] = acc[m];
      ptile = cur; have_prev = true;
    } else {
      const Geo g = geo_of(cur);
      if (want_stats) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        stats_rows(acc, g, 0, MR, s1, s2);
        stats_park(s1, s2, it & 1);
        pending = true; ctile = cur; cparity = it & 1;
      }
      store_rows(acc, cur, g, 0, MR);
      nst = NST;
    }
    DSTAMP(u3); DACC(d_tail, u3, u2);
    // the tile in flight has landed (everything older than this iteration's own issue); this iteration's stores may stay in flight
    if (more2) { if (nst) CD_WAIT(PPW + NST); else CD_WAIT(PPW); }
    else if (more1) { if (nst) wait_vm<NST>(); else wait_vm<0>(); }
    else wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DSTAMP(u4); DACC(d_wait, u4, u3);
    __builtin_amdgcn_s_barrier();
    DSTAMP(u0);
#ifdef CONV64_STAMPS
    d_bar += u0 - u4; d_n += 1;
    if (!more1 && lane == 0 && blockIdx.x < 256) {
      unsigned long long* d = conv64_dma_dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      d[0] = d_head; d[1] = d_cmp; d[2] = d_tail; d[3] = d_wait; d[4] = d_bar; d[5] = d_n;
    }
#endif
    if (!more1) break;
    cur = nx1; nx1 = nx2;
    t += stride;
    stage = (stage + 1) % NSTAGE;
    ++it;
  }
  // ---- drain
  if (pending) { combine(ctile, cparity); pending = false; }  // parked before the closing barrier
  if constexpr (PIPE) {
    if (have_prev) {  // the last tile's epilogue, not overlapped
      const Geo g = geo_of(ptile);
      if (want_stats) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        stats_rows(pacc, g, 0, MR, s1, s2);
        stats_park(s1, s2, it & 1);
      }
      store_rows(pacc, ptile, g, 0, MR);
      if (want_stats) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        combine(ptile, it & 1);
      }
    }
  }
#undef CD_PIECE
#undef CD_WAIT
}

bool conv64_dma_eligible(int mode, int dtype, const ConvArgs& a) {
  if (mode != MODE_G3S1 || dtype != MIA_BF16) return false;
  if (a.c1 != C || a.c2 != 0 || a.o1 != C || (a.o2 != 0 && a.o2 != C) || a.npad != a.o1 + a.o2 || a.kpad != C) return false;
  if (!a.vec_in || !a.vec_out) return false;
  if ((size_t)a.Hin * a.Win * C * 2 >= ((size_t)1 << 31)) return false;
  if (a.Hout <= 8) return false;  // the statistics layout of small maps uses 8-row tiles (mia_conv_mma_tiles)
  return true;
}

static int cd_num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

int conv64_dma_launch(const ConvArgs& a, int reserve, hipStream_t st) {
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int total = a.N * tiles_per_img;
  const int ncu = persistent_cus(cd_num_cus(), reserve);
  const int nblk = total < ncu ? ((total + 7) / 8) * 8 : ncu;  // one workgroup per CU; a multiple of 8: per-XCD runs tile the step
  const int run = nblk / 8;
  if (a.o2 == C) hipLaunchKernelGGL(conv64_dma_kernel<128>, dim3(nblk), dim3(512), 0, st, a, total, tiles_per_img, run);
  else hipLaunchKernelGGL(conv64_dma_kernel<64>, dim3(nblk), dim3(512), 0, st, a, total, tiles_per_img, run);
  return MIA_OK;
}
