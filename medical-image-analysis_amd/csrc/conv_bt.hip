// 3x3 / stride 1 / pad 1 bf16 convolution (and its input gradient, taps flipped) for the wide levels of the network: the
// "big tile" kernel.  Reference layer: src/models/unet/blocks.py:83-90 (Conv2d 3x3 of a PlainBlock) at channels_list[l >= 1],
// and the decoder's first conv of every level (two sources = torch.cat([skip, up]) eliminated, src/models/unet/unet.py:213).
//
// Why a third conv kernel: the 256-thread tile kernel (conv_mma_fast.hip) moves 100 bytes L2 -> LDS per MFMA (a 36.9 KB weight
// chunk + a 20.7 KB input chunk for every 576 MFMAs), through registers and a ds_write pass, with two barriers and a
// vmcnt(0) drain per chunk.  Here
//   * ONE 512-thread workgroup per CU owns a 16 x 32 pixel tile x NCH output channels (128, 96 or 64): a weight chunk serves
//     512 pixels and an input chunk NCH channels -- 49 bytes per MFMA at NCH = 128;
//   * every byte arrives by LDS-DMA (`buffer_load_dwordx4 ... offen lds`, issued and counted in inline asm): no staging
//     registers, no ds_write pass.  K loop = 32-channel chunks x the three horizontal taps tb; a step (chunk, tb) needs the
//     chunk's input image (18 x 34 pixels x 64 B, two buffers) and the weights of the three taps (ta, tb) (3 x NCH x 64 B, a
//     ring of three slots, slot index == tb).  Step s issues the pieces of step s + 2, runs its MFMAs, waits with a COUNTED
//     vmcnt for everything older than its own issue, and passes ONE barrier;
//   * MFMA operands swapped (A = weights, M = output channel; B = pixels, N = pixel): a lane's four accumulator registers
//     are four consecutive output channels of one pixel, so the tile is stored straight from the accumulators
//     (v_permlane16_swap pairs two rows into 16-byte stores) and the statistics are DPP row sums -- no LDS transpose;
//   * per tb a wave reads each of its MT + 2 input rows once and slides the three vertical taps over it in registers
//     (22 ds_read_b128 per 96 MFMAs at NCH = 128).
// LDS images (both bank-conflict free for ds_read_b128, checked by simulation of the 16-lane read groups):
//   input : [pixel p = row * 34 + col][4 slots of 16 B]; slot = k-group ^ ((col >> 2) & 3); MFMA column n <-> pixel pi16(n)
//   weight: [tap ta][channel][4 slots]; slot = k-group ^ h[(channel >> 2) & 3], h = {0, 2, 3, 1}
// The DMA destination is lane-linear (1 KB per wave instruction = 16 pixels / channels x 64 B), so both swizzles are applied
// on the SOURCE address of each lane.  The weights are read from the ordinary packed tensor [tap][npad][kpad].
//
// Contract (conv_bt_eligible, otherwise conv_mma_fast runs): bf16, MODE_G3S1, Hout > 8, c1 % 32 == 0, c2 in {0, c1},
// (o1 + o2) % NCH == 0 and o1 % NCH == 0, 16-byte aligned pointers, per-image tensors and the packed weights < 2 GiB.
#include "conv_common.h"
#include <type_traits>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define BT_SENT 0xFFFFFFF0u /* always beyond num_records: loads return zero, stores are dropped */

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
// One LDS-DMA piece: 64 lanes x 16 bytes, lane L lands at lds_dst + 16 L.  M0 (the LDS base) is written and read inside this
// one statement (hipcc uses M0 for nothing else in this kernel); s_nop 4 covers the VALU-written-SGPR -> VMEM hazard of the
// descriptor / offset operands, which hipcc does not pad inside an asm statement.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

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

constexpr int TW = 32, TH = 16, IW = TW + 2, IH = TH + 2;
constexpr int NPIX = IH * IW;                  // 612 halo pixels
constexpr int IPW = 5;                         // image pieces per wave and chunk (8 x 5 = 40 >= 39; piece 39 is padding)
constexpr int IMG_BYTES = 8 * IPW * 1024;      // 40960
constexpr int ROW_BYTES = IW * 64;             // 2176

template <int WC, int NCT, int MT>
struct BtGeo {
  static constexpr int NCH = WC * NCT * 16;
  static constexpr int SLOT = 3 * NCH * 64;            // weights of the three taps (ta, tb) of one chunk
  static constexpr int WPIECES = 3 * NCH / 16;
  static constexpr int WPW = (WPIECES + 7) / 8;        // pieces per wave and step (padded with a dummy piece)
  static constexpr bool WDUMMY = (WPIECES % 8) != 0;
  static constexpr int RING = 2 * IMG_BYTES;           // byte offset of the weight ring
  static constexpr int DUMP = RING + 3 * SLOT;         // 1 KB target of dummy pieces
  static constexpr int LDS = DUMP + (WDUMMY ? 1024 : 0);
  static_assert((8 / WC) * MT * 16 == TW * TH, "tile = 512 pixels");
};

}  // namespace

template <int WC, int NCT, int MT>
__global__ __launch_bounds__(512, 2) void conv_bt_kernel(const ConvArgs a, int ptiles, int ptx, int tx32) {
  using G = BtGeo<WC, NCT, MT>;
  constexpr int NCH = G::NCH, SLOT = G::SLOT, WPW = G::WPW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n16 = lane & 15;
  const int pr = pi16(n16);
  const int wc = wave % WC, wp = wave / WC;
  const int wpx = wp & 1, wpy = wp >> 1;

  // ---- block -> (channel block, pixel tile): every XCD (blocks b, b + 8, ...) walks the pixel tiles == xcd (mod 8) with the
  // channel block as the slow index, so the workgroups resident at one time stream the SAME weight chunks (L2 / MALL hits)
  const int b = blockIdx.x, slot_id = b >> 3;
  const int cb = slot_id / ptx;
  const int tile = (slot_id - cb * ptx) * 8 + (b & 7);
  if (tile >= ptiles) return;  // uniform
  const int per_img = a.tiles_y * tx32;
  const int img = tile / per_img;
  const int trem = tile - img * per_img;
  const int ty = trem / tx32, tx = trem - ty * tx32;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = cb * NCH;

  const int ctot = a.c1 + a.c2, nchunks = ctot >> 5;
  const size_t ipix = (size_t)a.Hin * a.Win;
  const unsigned img_bytes = (unsigned)(ipix * a.c1 * 2);
  const bf16_t* in1 = static_cast<const bf16_t*>(a.in1);
  const bf16_t* in2 = static_cast<const bf16_t*>(a.c2 ? a.in2 : a.in1);
  const i32x4 rs1 = rsrc_words(in1 + (size_t)img * ipix * a.c1, img_bytes);
  const i32x4 rs2 = rsrc_words(in2 + (size_t)img * ipix * a.c1, img_bytes);  // c2 == c1 (contract)
  const i32x4 rsw = rsrc_words(a.wp, (unsigned)((size_t)9 * a.npad * a.kpad * 2));

  // ---- DMA lane constants.  Image piece k = wave + 8 j covers halo pixels 16 k .. 16 k + 15; lane L = (pixel L >> 2, slot L & 3)
  unsigned ioff[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int p = 16 * (wave + 8 * j) + (lane >> 2);
    const int row = p / IW, col = p - row * IW;
    const int gy = oy0 - 1 + row, gx = ox0 - 1 + col;
    const bool ok = (p < NPIX) & ((unsigned)gy < (unsigned)a.Hin) & ((unsigned)gx < (unsigned)a.Win);
    const int chunk = (lane & 3) ^ ((col >> 2) & 3);
    ioff[j] = ok ? (unsigned)(((gy * a.Win + gx) * a.c1 + chunk * 8) * 2) : BT_SENT;
  }
  // weight piece i = wave + 8 jj = (tap row ta, 16-channel block j): lane L = (channel L >> 2, slot L & 3)
  const int hsw = (0x1E >> (2 * ((lane >> 4) & 3))) & 3;  // h = {0, 2, 3, 1}
  const unsigned wlane = (unsigned)(((lane >> 2) * a.kpad + ((lane & 3) ^ hsw) * 8) * 2);
  int w_ta[WPW], w_row[WPW];   // uniform: tap row and first weight row (n0 + 16 j) of this wave's pieces
  bool w_ok[WPW];
#pragma unroll
  for (int jj = 0; jj < WPW; ++jj) {
    const int i = wave + 8 * jj;
    w_ok[jj] = i < G::WPIECES;
    w_ta[jj] = i / (NCH / 16);
    w_row[jj] = n0 + 16 * (i - w_ta[jj] * (NCH / 16));
  }
  auto issue_w = [&](int chunk, int tb) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < WPW; ++jj) {
      if (!G::WDUMMY || w_ok[jj]) {
        const int t = w_ta[jj] * 3 + tb;
        const int tw = a.flip ? 8 - t : t;
        const unsigned soff = (unsigned)(((tw * a.npad + w_row[jj]) * a.kpad + chunk * 32) * 2);
        dma16(rsw, wlane, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(lds0 + G::RING + tb * SLOT + (wave + 8 * jj) * 1024));
      } else {
        dma16(rsw, BT_SENT, 0, __builtin_amdgcn_readfirstlane(lds0 + G::DUMP));
      }
    }
  };
  auto issue_img = [&](int chunk, int j) __attribute__((always_inline)) {
    const int c0 = chunk * 32;
    const bool second = c0 >= a.c1;  // uniform: chunks never straddle the two sources
    const unsigned soff = (unsigned)((second ? c0 - a.c1 : c0) * 2);
    const unsigned dst = lds0 + (chunk & 1) * IMG_BYTES + (wave + 8 * j) * 1024;
    dma16(second ? rs2 : rs1, ioff[j], __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(dst));
  };

  // ---- fragment read addresses (bytes from the start of LDS)
  // B (pixels): column n of the MFMA <-> pixel pr of the wave's 16-pixel strip; halo column = 16 wpx + pr + tb, halo row = MT wpy + r
  unsigned bbase[3];
#pragma unroll
  for (int tb = 0; tb < 3; ++tb) {
    const int col = 16 * wpx + pr + tb;
    bbase[tb] = (unsigned)(((MT * wpy) * IW + col) * 64 + ((q ^ ((col >> 2) & 3)) * 16));
  }
  // A (weights): row m of the MFMA = output channel 16 (NCT wc + ct) + n16
  const int hrd = (0x1E >> (2 * ((n16 >> 2) & 3))) & 3;
  const unsigned abase = (unsigned)(G::RING + (wc * NCT * 16 + n16) * 64 + ((q ^ hrd) * 16));

  // ---- accumulators start at the bias (lane: channels 16 ct + 4 q .. + 3 of its wave's channel range)
  f32x4 acc[MT][NCT];
  {
    f32x4 bv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[ct][r] = a.bias ? a.bias[n0 + (wc * NCT + ct) * 16 + 4 * q + r] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[m][ct] = bv[ct];
  }
  // the bias loads are the only compiler-visible vector-memory loads before the epilogue: retire them here so that no
  // compiler-inserted wait lands inside the counted DMA pipeline
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- prologue: image of chunk 0 and the weights of steps 0 and 1
#pragma unroll
  for (int j = 0; j < IPW; ++j) issue_img(0, j);
  issue_w(0, 0);
  issue_w(0, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
  __builtin_amdgcn_s_barrier();

  auto compute = [&](auto tbc, unsigned imgoff) __attribute__((always_inline)) {
    constexpr int tb = decltype(tbc)::value;
    // opaque copies: hipcc would otherwise hoist one address register per (ta, ct) fragment out of the chunk loop (36 VGPRs)
    // instead of folding the constants into the ds_read offset fields
    unsigned ab = abase, bb = bbase[tb] + imgoff;
    asm volatile("" : "+v"(ab), "+v"(bb));
    const unsigned char* ap = smem + ab + tb * SLOT;
    const unsigned char* bp = smem + bb;
    u32x4 wf[3][NCT], fr[3];
    auto load_w = [&](int ta) __attribute__((always_inline)) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) wf[ta][ct] = *reinterpret_cast<const u32x4*>(ap + (ta * NCH + ct * 16) * 64);
    };
    auto load_row = [&](int r) __attribute__((always_inline)) { fr[r % 3] = *reinterpret_cast<const u32x4*>(bp + r * ROW_BYTES); };
    load_w(0); load_row(0);
    __builtin_amdgcn_sched_barrier(0);
    load_w(1); load_row(1);
#pragma unroll
    for (int r = 0; r < MT + 2; ++r) {
      if (r + 2 < MT + 2) load_row(r + 2);
      if (r == 0) load_w(2);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this row's MFMAs
#pragma unroll
      for (int ta = 0; ta < 3; ++ta) {
        const int m = r - ta;
        if (m >= 0 && m < MT) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[m][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ta][ct]), __builtin_bit_cast(bf16x8, fr[r % 3]),
                                                                 acc[m][ct], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using TB0 = std::integral_constant<int, 0>;
  using TB1 = std::integral_constant<int, 1>;
  using TB2 = std::integral_constant<int, 2>;

  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;  // uniform
    const unsigned imgoff = (c & 1) * IMG_BYTES;
    // step (c, 0): issue W(c, 2) and the first three image pieces of chunk c + 1
    issue_w(c, 2);
    if (more) { issue_img(c + 1, 0); issue_img(c + 1, 1); issue_img(c + 1, 2); }
    compute(TB0{}, imgoff);
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 3) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
    __builtin_amdgcn_s_barrier();
    // step (c, 1): issue W(c + 1, 0) and the last two image pieces
    if (more) { issue_w(c + 1, 0); issue_img(c + 1, 3); issue_img(c + 1, 4); }
    compute(TB1{}, imgoff);
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // step (c, 2): issue W(c + 1, 1)
    if (more) issue_w(c + 1, 1);
    compute(TB2{}, imgoff);
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: statistics partials + 16-byte stores straight from the accumulators
  const int nout = a.o1 + a.o2;
  const bool to2 = n0 >= a.o1;  // uniform: a channel block lies in one destination (contract)
  const int cn = to2 ? a.o2 : a.o1;
  const int nloc = (to2 ? n0 - a.o1 : n0) + wc * NCT * 16;
  bf16_t* obase_p = static_cast<bf16_t*>(to2 ? a.out2 : a.out1);
  const size_t opix = (size_t)a.Hout * a.Wout;
  const rsrc_t rso = make_rsrc(obase_p + (size_t)img * opix * cn, (unsigned)(opix * cn * 2));
  const int wy0 = oy0 + MT * wpy, wx = ox0 + 16 * wpx + pr;  // first output row of the wave, this lane's output column
  const bool colok = wx < a.Wout;
  const bool full = (oy0 + TH <= a.Hout) && (ox0 + TW <= a.Wout);  // uniform

  if (a.stats != nullptr) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][16 NCT channels][2]; the K loop's last barrier has passed
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float wgt = (full || (colok && wy0 + m < a.Hout)) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[m][ct][r], vm = v * wgt;
          s1[r] += vm; s2[r] += vm * v;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
      if (n16 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[(wave * NCT * 16 + ct * 16 + 4 * q + r) * 2 + 0] = s1[r];
          red[(wave * NCT * 16 + ct * 16 + 4 * q + r) * 2 + 1] = s2[r];
        }
      }
    }
    __syncthreads();
    // one 16 x 16 statistics tile per x strip: sum the waves stacked in y (same strip, same channel group)
    constexpr int NWY = (8 / WC) / 2;
    constexpr int ENT = 2 * NCH;  // (strip, channel) entries of the workgroup
    for (int e = tid; e < ENT; e += 512) {
      const int strip = e / NCH, ch = e - strip * NCH;
      const int wcc = ch / (NCT * 16), chl = ch - wcc * (NCT * 16);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int y = 0; y < NWY; ++y) {
        const int w = (2 * y + strip) * WC + wcc;
        t1 += red[(w * NCT * 16 + chl) * 2 + 0];
        t2 += red[(w * NCT * 16 + chl) * 2 + 1];
      }
      const int stx = 2 * tx + strip;
      if (stx < a.tiles_x) {
        const size_t st = ((size_t)img * a.tiles_y + ty) * a.tiles_x + stx;
        float* dst = a.stats + (st * nout + n0 + ch) * 2;
        dst[0] = t1; dst[1] = t2;
      }
    }
  }

  const int qodd = q & 1;
  const int row_bytes = a.Wout * cn * 2;
  const unsigned obase = (unsigned)((((wy0 + qodd) * a.Wout) + wx) * cn * 2 + (nloc + 8 * (q >> 1)) * 2);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
    for (int m = 0; m < MT; m += 2) {
      const unsigned x0 = pack_bf16x2(acc[m][ct][0], acc[m][ct][1]), x1 = pack_bf16x2(acc[m][ct][2], acc[m][ct][3]);
      const unsigned y0 = pack_bf16x2(acc[m + 1][ct][0], acc[m + 1][ct][1]), y1 = pack_bf16x2(acc[m + 1][ct][2], acc[m + 1][ct][3]);
      // even q gets its partner's row-m half (8 consecutive channels of row m), odd q the same 8 channels of row m + 1
      const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      const u32x4 d = {r0[0], r1[0], r0[1], r1[1]};
      const bool ok = full || (colok && (wy0 + m + qodd < a.Hout));
      const unsigned voff = ok ? obase + (unsigned)(m * row_bytes + ct * 32) : BT_SENT;
      __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)voff, 0, 0);
    }
  }
}

static int bt_nch(const ConvArgs& a) {
  const int nout = a.o1 + a.o2;
  if (nout % 128 == 0 && a.o1 % 128 == 0) return 128;
  if (nout % 96 == 0 && a.o1 % 96 == 0) return 96;
  if (nout % 64 == 0 && a.o1 % 64 == 0) return 64;
  return 0;
}

bool conv_bt_eligible(int mode, int dtype, const ConvArgs& a) {
  if (mode != MODE_G3S1 || dtype != MIA_BF16) return false;
  if (!a.vec_in || !a.vec_out) return false;
  if (a.c1 % 32 != 0 || (a.c2 != 0 && a.c2 != a.c1) || a.c1 + a.c2 < 64) return false;
  if (bt_nch(a) == 0) return false;
  if (a.Hout <= 8) return false;  // the statistics layout of small maps uses 8-row tiles (mia_conv_mma_tiles)
  const size_t lim = (size_t)1 << 31;
  const size_t pix = (size_t)a.Hin * a.Win;
  if (pix * a.c1 * 2 >= lim || pix * (a.o1 > a.o2 ? a.o1 : a.o2) * 2 >= lim) return false;
  if ((size_t)9 * a.npad * a.kpad * 2 >= lim) return false;
  return true;
}

template <int WC, int NCT, int MT>
static void bt_launch(const ConvArgs& a, int nch, hipStream_t st) {
  const int tx32 = (a.tiles_x + 1) / 2;
  const int ptiles = a.N * a.tiles_y * tx32;
  const int ptx = (ptiles + 7) / 8;
  const int nb = (a.o1 + a.o2) / nch;
  hipLaunchKernelGGL((conv_bt_kernel<WC, NCT, MT>), dim3(8 * ptx * nb), dim3(512), 0, st, a, ptiles, ptx, tx32);
}

int conv_bt_launch(const ConvArgs& a, hipStream_t st) {
  const int nch = bt_nch(a);
  if (nch == 128) bt_launch<2, 4, 8>(a, nch, st);
  else if (nch == 96) bt_launch<2, 3, 8>(a, nch, st);
  else if (nch == 64) bt_launch<1, 4, 4>(a, nch, st);
  else return MIA_EARG;
  return MIA_OK;
}
