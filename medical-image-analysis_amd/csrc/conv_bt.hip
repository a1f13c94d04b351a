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
#include <stdlib.h>
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
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory", "m0");
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
  static constexpr int WPW = (WPIECES + 7) / 8;        // weight pieces per step of waves 0 .. WPIECES % 8 - 1 (the others: one less when WDUMMY)
  static constexpr bool WDUMMY = (WPIECES % 8) != 0;  // the pieces do not divide evenly among the 8 waves
  static constexpr int MAXP = WPW + 3;                  // most pieces a wave issues in one step
  static constexpr int RING = 2 * IMG_BYTES;           // byte offset of the weight ring
  static constexpr int BIAS = RING + 3 * SLOT;         // bias of the current channel block (1 KB piece)
  static constexpr int RED = BIAS + 1024;                  // [8 waves][16 NCT channels][2] statistics exchange
  static constexpr int LDS = RED + 8 * NCT * 16 * 8;
  static_assert((8 / WC) * MT * 16 == TW * TH, "tile = 512 pixels");
  static_assert(WPW + 3 <= MT + 1, "one piece per MFMA row");
};

}  // namespace

// Diagnostic build only (-DCONV64_STAMPS, tools/conv64_stamps.py bt): per-wave cycle sums of the phases.  The shipped library
// executes no stamp.
#ifdef CONV64_STAMPS
__device__ unsigned long long conv_bt_dbg[256 * 8 * 8];
#define BSTAMP(var)                                                                       \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define BACC(dst, t1, t0) dst += (t1) - (t0)
extern "C" int mia_conv_bt_debug_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(conv_bt_dbg), sizeof(conv_bt_dbg));
}
#else
#define BSTAMP(var) do { } while (0)
#define BACC(dst, t1, t0) do { } while (0)
#endif

// wait until at most N vector-memory operations of this wave are outstanding
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct BtWork { int img, ty, tx, n0; };

template <int WC, int NCT, int MT>
__global__ __launch_bounds__(512, 2) void conv_bt_kernel(const ConvArgs a, int ptiles, int tx32, int nwork, int nb_tile_major) {
  using G = BtGeo<WC, NCT, MT>;
  constexpr int NCH = G::NCH, SLOT = G::SLOT, WPW = G::WPW;
  constexpr int NSTORE = (MT / 2) * NCT;  // tile stores per wave (every wave issues all of them, masked lanes out of range)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Every lane-dependent constant below (fragment addresses, DMA lane offsets, store offsets) is RECOMPUTED where it is used
  // from an opaque copy of the thread id: hipcc otherwise keeps each of them (and their sub-expressions) in a register for
  // the whole kernel -- at 128 accumulator + 60 fragment registers per lane that spills, and a spill breaks the hand-counted
  // vmcnt (the build checks for zero scratch).
  auto lane_id = [&]() __attribute__((always_inline)) -> int {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t & 63;
  };
  const int wc = wave % WC, wp = wave / WC;
  const int wpx = wp & 1, wpy = wp >> 1;

  // ---- work list: a workgroup walks items b, b + grid, b + 2 grid, ...; all workgroups run in step, so at any time they stream
  // the same K chunk.  Item -> (channel block, pixel tile):
  //   nb_tile_major = NB > 0 (tile count a multiple of 8): w = ((tile / 8) * NB + cb) * 8 + tile % 8 -- the NB channel blocks of
  //     a tile run at the same time on ONE XCD (blocks b, b + 8, ... share an XCD and its L2), so the tile's input is fetched
  //     into that L2 once instead of NB times a whole pass apart (PMC at 256 ch: 1.11 GB -> see profiles/), and every XCD
  //     streams all NB weight blocks (NB x 74 KB per chunk);
  //   nb_tile_major = 0: channel block slow, w = cb * tiles + tile.
  const int per_img = a.tiles_y * tx32;
  auto decode = [&](int w) __attribute__((always_inline)) -> BtWork {
    BtWork t;
    int cb, tile;
    if (nb_tile_major > 0) {
      const int slot = w >> 3;
      const int grp = slot / nb_tile_major;
      cb = slot - grp * nb_tile_major;
      tile = grp * 8 + (w & 7);
    } else {
      cb = w / ptiles;
      tile = w - cb * ptiles;
    }
    t.img = tile / per_img;
    const int trem = tile - t.img * per_img;
    t.ty = trem / tx32;
    t.tx = trem - t.ty * tx32;
    t.n0 = cb * NCH;
    return t;
  };

  const int ctot = a.c1 + a.c2, nchunks = ctot >> 5;
  const size_t ipix = (size_t)a.Hin * a.Win;
  const unsigned img_bytes = (unsigned)(ipix * a.c1 * 2);
  const bf16_t* in1 = static_cast<const bf16_t*>(a.in1);
  const bf16_t* in2 = static_cast<const bf16_t*>(a.c2 ? a.in2 : a.in1);
  const i32x4 rsw = rsrc_words(a.wp, (unsigned)((size_t)9 * a.npad * a.kpad * 2));
  const bool has_bias = a.bias != nullptr;  // uniform
  const i32x4 rsb = rsrc_words(has_bias ? a.bias : (const float*)a.wp, (unsigned)((a.o1 + a.o2) * 4));

  // ---- issue state: the tile whose input image is being fetched (the NEXT tile during the last chunk of the current one)
  i32x4 irs1, irs2;
  unsigned ioff[IPW];  // image piece k = wave + 8 j covers halo pixels 16 k .. 16 k + 15; lane L = (pixel L >> 2, slot L & 3)
  auto set_img_state = [&](const BtWork& t) __attribute__((always_inline)) {
    irs1 = rsrc_words(in1 + (size_t)t.img * ipix * a.c1, img_bytes);
    irs2 = rsrc_words(in2 + (size_t)t.img * ipix * a.c1, img_bytes);  // c2 == c1 (contract)
    const int oy0 = t.ty * TH, ox0 = t.tx * TW;
    // opaque copy of the lane id: hipcc would otherwise hoist the tile-independent parts of the five offsets (row, column,
    // swizzled chunk of every piece) out of the tile loop and keep them in ~15 registers
    const int lv = lane_id();
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int p = 16 * (wave + 8 * j) + (lv >> 2);
      const int row = p / IW, col = p - row * IW;
      const int gy = oy0 - 1 + row, gx = ox0 - 1 + col;
      const bool ok = (p < NPIX) & ((unsigned)gy < (unsigned)a.Hin) & ((unsigned)gx < (unsigned)a.Win);
      const int chunk = (lv & 3) ^ ((col >> 2) & 3);
      ioff[j] = ok ? (unsigned)(((gy * a.Win + gx) * a.c1 + chunk * 8) * 2) : BT_SENT;
    }
  };
  // weight piece i = wave + 8 jj = (tap row ta, 16-channel block j): lane L = (channel L >> 2, slot L & 3).  wsoff[tb][jj] =
  // byte offset of the piece's first weight row in the packed tensor [tap][npad][kpad] for the channel block being fetched
  auto make_wlane = [&]() __attribute__((always_inline)) -> unsigned {
    const int lane = lane_id();
    const int hsw = (0x1E >> (2 * ((lane >> 4) & 3))) & 3;  // h = {0, 2, 3, 1}
    return (unsigned)(((lane >> 2) * a.kpad + ((lane & 3) ^ hsw) * 8) * 2);
  };
  unsigned wsoff[3][WPW];
  bool w_ok[WPW];
  auto set_w_state = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < WPW; ++jj) {
      const int i = wave + 8 * jj;
      w_ok[jj] = i < G::WPIECES;
      const int ta = i / (NCH / 16), j16 = 16 * (i - ta * (NCH / 16));
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const int t = ta * 3 + tb;
        const int tw = a.flip ? 8 - t : t;
        wsoff[tb][jj] = __builtin_amdgcn_readfirstlane((unsigned)(((tw * a.npad + n0 + j16) * a.kpad) * 2));
      }
    }
  };
  auto issue_w1 = [&](unsigned wlane, int chunk, int tb, int jj) __attribute__((always_inline)) {
    if (!G::WDUMMY || w_ok[jj])  // a wave without a piece in the last round issues nothing: its waits count one piece less (wait_w)
      dma16(rsw, wlane, wsoff[tb][jj] + (unsigned)(chunk * 64), __builtin_amdgcn_readfirstlane(lds0 + G::RING + tb * SLOT + (wave + 8 * jj) * 1024));
  };
  auto issue_w = [&](int chunk, int tb) __attribute__((always_inline)) {
    const unsigned wlane = make_wlane();
#pragma unroll
    for (int jj = 0; jj < WPW; ++jj) issue_w1(wlane, chunk, tb, jj);
  };
  auto issue_img = [&](int chunk, int buf, int j) __attribute__((always_inline)) {
    const int c0 = chunk * 32;
    const bool second = c0 >= a.c1;  // uniform: chunks never straddle the two sources
    const unsigned soff = (unsigned)((second ? c0 - a.c1 : c0) * 2);
    const unsigned dst = lds0 + buf * IMG_BYTES + (wave + 8 * j) * 1024;
    dma16(second ? irs2 : irs1, ioff[j], __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(dst));
  };
  // bias of a channel block: NCH floats, lanes 0 .. NCH / 4 - 1 (every wave writes the same bytes; a wave reads them after
  // its OWN piece has landed)
  auto issue_bias = [&](int n0) __attribute__((always_inline)) {
    const int lane = lane_id();
    const unsigned blane = lane < NCH / 4 ? (unsigned)(lane * 16) : BT_SENT;
    dma16(rsb, blane, __builtin_amdgcn_readfirstlane((unsigned)(n0 * 4)), __builtin_amdgcn_readfirstlane(lds0 + G::BIAS));
  };

  // ---- fragment read addresses (bytes from the start of LDS), computed per step in compute()
  // B (pixels): column n of the MFMA <-> pixel pi16(n) of the wave's 16-pixel strip; halo column = 16 wpx + pixel + tb, halo row
  // = MT wpy + r
  // A (weights): which output channel (within the wave's 16 NCT) row i = n16 of tile ct is.  Even NCT: channel = 32 (ct >> 1) +
  // 8 (i >> 2) + 4 (ct & 1) + (i & 3), so a lane's accumulators of the tile pair (2 k, 2 k + 1) are EIGHT consecutive channels
  // of its pixel = one 16-byte store, and the four lanes of a pixel write 64 contiguous bytes.  Odd NCT: channel = 16 ct + i
  // (four consecutive channels per tile; rows are paired by v_permlane16_swap in the epilogue).
  constexpr bool PAIRCT = (NCT % 2) == 0;
  auto ch_local = [&](int ct, int i) __attribute__((always_inline)) -> int {
    return PAIRCT ? 32 * (ct >> 1) + 8 * (i >> 2) + 4 * (ct & 1) + (i & 3) : 16 * ct + i;
  };
  // fragment-read lane constants: cq = halo column of this lane's pixel (16 wpx + pi16(n)) | k-group << 8; ab0 / ab1 = byte
  // address of this lane's weight row in the even / odd channel tiles of ring slot 0
  unsigned cq, ab0, ab1;
  {
    const int lane = lane_id(), q = lane >> 4, n16 = lane & 15;
    cq = (unsigned)((16 * wpx + pi16(n16)) | (q << 8));
    unsigned abv[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int chl = ch_local(par, n16);
      const int hrd = (0x1E >> (2 * ((chl >> 2) & 3))) & 3;
      abv[par] = (unsigned)(G::RING + (wc * NCT * 16 + chl) * 64 + ((q ^ hrd) * 16));
    }
    ab0 = abv[0]; ab1 = abv[1];
    asm volatile("" : "+v"(cq), "+v"(ab0), "+v"(ab1));  // opaque: not rematerialised from the thread id at every use
  }
  f32x4 acc[MT][NCT];
  // one step: fragment reads + MFMAs of tap column tb; piece(k) (k = 0 .. 5) issues the k-th LDS-DMA piece of the step, placed
  // behind the MFMAs of row k + 1 so that its scalar work and issue slot hide under the matrix pipe
  // fin = the last step of a tile: every piece is issued behind row 1 (so that all of them are OLDER than the tile stores) and
  // fin_row(m) runs as soon as output row m is complete (behind the MFMAs of input row m + 2)
  auto compute = [&](auto tbc, auto finc, unsigned imgoff, auto&& piece, auto&& fin_row) __attribute__((always_inline)) {
    constexpr int tb = decltype(tbc)::value;
    constexpr bool fin = decltype(finc)::value;
    // lane constants kept in three registers (cq, ab0, ab1): recomputing them here costs ~40 instructions (with divergent
    // branches for the pixel permutation) on the critical path right after every barrier
    const int c0 = (int)(cq & 0xFF), q = (int)(cq >> 8);
    const int col = c0 + tb;
    unsigned bb = (unsigned)(((MT * wpy) * IW + col) * 64 + ((q ^ ((col >> 2) & 3)) * 16)) + imgoff;
    unsigned ab[2] = {ab0 + (unsigned)(tb * SLOT), ab1 + (unsigned)(tb * SLOT)};
    // opaque: the constants below must fold into the ds_read offset fields, not into one hoisted register per fragment
    asm volatile("" : "+v"(ab[0]), "+v"(ab[1]), "+v"(bb));
    const unsigned char* ap0 = smem + ab[0];
    const unsigned char* ap1 = smem + ab[1];
    const unsigned char* bp = smem + bb;
    u32x4 wf[3][NCT], fr[3];
    auto load_w = [&](int ta) __attribute__((always_inline)) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int step = (PAIRCT ? 32 * (ct >> 1) : 16 * (ct & ~1)) * 64;  // bytes from tile (ct & 1) to tile ct
        wf[ta][ct] = *reinterpret_cast<const u32x4*>(((ct & 1) ? ap1 : ap0) + ta * NCH * 64 + step);
      }
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
      if (tb == 2) {  // only W pieces in this step
        if (fin) {
          if (r == 1) {
#pragma unroll
            for (int k = 0; k < WPW; ++k) piece(k);
          }
          if (r >= 2) fin_row(r - 2);
        } else if (r >= 1 && r - 1 < WPW) {
          piece(r - 1);
        }
      } else if (r >= 1 && r - 1 < G::MAXP) {
        piece(r - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using TB0 = std::integral_constant<int, 0>;
  using TB1 = std::integral_constant<int, 1>;
  using TB2 = std::integral_constant<int, 2>;

  // ---- output of one tile.  store_row(m): 16-byte stores of output row m straight from the accumulators (called from the final
  // step as rows complete, so the ~70 cycles the texture path spends on every store instruction overlap the remaining MFMAs
  // and the statistics below).  The accumulators are left intact.
  const int nout = a.o1 + a.o2;
  struct OutState { rsrc_t rso; unsigned obase; int row_bytes, wy0; bool colok, full; };
  auto out_state = [&](const BtWork& t) __attribute__((always_inline)) -> OutState {
    OutState o;
    const int oy0 = t.ty * TH, ox0 = t.tx * TW;
    const bool to2 = t.n0 >= a.o1;  // uniform: a channel block lies in one destination (contract)
    const int cn = to2 ? a.o2 : a.o1;
    const int nloc = (to2 ? t.n0 - a.o1 : t.n0) + wc * NCT * 16;
    bf16_t* obase_p = static_cast<bf16_t*>(to2 ? a.out2 : a.out1);
    const size_t opix = (size_t)a.Hout * a.Wout;
    o.rso = make_rsrc(obase_p + (size_t)t.img * opix * cn, (unsigned)(opix * cn * 2));
    o.wy0 = oy0 + MT * wpy;
    const int lane = lane_id(), lq = lane >> 4;
    const int wx = ox0 + 16 * wpx + pi16(lane & 15);  // this lane's output column
    o.colok = wx < a.Wout;
    o.full = (oy0 + TH <= a.Hout) && (ox0 + TW <= a.Wout);  // uniform
    o.row_bytes = a.Wout * cn * 2;
    if (PAIRCT) o.obase = (unsigned)(((o.wy0 * a.Wout) + wx) * cn * 2 + (nloc + 8 * lq) * 2);
    else o.obase = (unsigned)((((o.wy0 + (lq & 1)) * a.Wout) + wx) * cn * 2 + (nloc + 8 * (lq >> 1)) * 2);
    return o;
  };
  auto store_row = [&](const OutState& o, int m) __attribute__((always_inline)) {
    if constexpr (PAIRCT) {
      // a lane's tile pair (2 k, 2 k + 1) = channels 32 k + 8 q .. + 7 of its pixel: one 16-byte store per row and pair
#pragma unroll
      for (int k = 0; k < NCT / 2; ++k) {
        const u32x4 d = {pack_bf16x2(acc[m][2 * k][0], acc[m][2 * k][1]), pack_bf16x2(acc[m][2 * k][2], acc[m][2 * k][3]),
                         pack_bf16x2(acc[m][2 * k + 1][0], acc[m][2 * k + 1][1]), pack_bf16x2(acc[m][2 * k + 1][2], acc[m][2 * k + 1][3])};
        const bool ok = o.full || (o.colok && (o.wy0 + m < a.Hout));
        const unsigned voff = ok ? o.obase + (unsigned)(m * o.row_bytes + k * 64) : BT_SENT;
        __builtin_amdgcn_raw_buffer_store_b128(d, o.rso, (int)voff, 0, 0);
      }
    } else {
      if (m & 1) {  // rows are stored in pairs (m - 1, m): v_permlane16_swap trades the 8-byte halves of the lane pairs (q, q ^ 1)
        const int qodd = (lane_id() >> 4) & 1;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const unsigned x0 = pack_bf16x2(acc[m - 1][ct][0], acc[m - 1][ct][1]), x1 = pack_bf16x2(acc[m - 1][ct][2], acc[m - 1][ct][3]);
          const unsigned y0 = pack_bf16x2(acc[m][ct][0], acc[m][ct][1]), y1 = pack_bf16x2(acc[m][ct][2], acc[m][ct][3]);
          // even q gets its partner's row-(m-1) half (8 consecutive channels), odd q the same 8 channels of row m
          const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          const u32x4 d = {r0[0], r1[0], r0[1], r1[1]};
          const bool ok = o.full || (o.colok && (o.wy0 + m - 1 + qodd < a.Hout));
          const unsigned voff = ok ? o.obase + (unsigned)((m - 1) * o.row_bytes + ct * 32) : BT_SENT;
          __builtin_amdgcn_raw_buffer_store_b128(d, o.rso, (int)voff, 0, 0);
        }
      }
    }
  };
  // statistics partials of the tile (after its last step): per-channel sum and sum of squares over the valid pixels
  // EARLY: variants with registers to spare (odd NCT, MT = 4) accumulate the per-lane sums row by row inside the final step
  // (stats_row, behind that step's MFMAs) -- the 96-channel launches of cfg5 have 9 steps per tile, where the ~2k cycles of
  // packed adds after the K loop weigh 6 %; at NCT = 4 / MT = 8 the 32 extra live registers would spill.
  constexpr bool EARLY = (NCT % 2 == 1) || MT == 4;
  struct LaneStats { float s1[NCT][4], s2[NCT][4]; };
  auto stats_row = [&](LaneStats& ls, const OutState& o, int m) __attribute__((always_inline)) {
    const float wgt = (o.full || (o.colok && o.wy0 + m < a.Hout)) ? 1.f : 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[m][ct][r], vm = o.full ? v : v * wgt;
        ls.s1[ct][r] += vm; ls.s2[ct][r] += vm * v;
      }
  };
  auto tile_stats = [&](const BtWork& t, const OutState& o, const LaneStats& ls) __attribute__((always_inline)) {
    float* red = reinterpret_cast<float*>(smem + G::RED);  // [wave][16 NCT channels][2]
    const int lane = lane_id(), q = lane >> 4, n16 = lane & 15;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      if (EARLY) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[r] = ls.s1[ct][r]; s2[r] = ls.s2[ct][r]; }
      } else if (o.full) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[m][ct][r]; s1[r] += v; s2[r] += v * v; }
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float wgt = (o.colok && o.wy0 + m < a.Hout) ? 1.f : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[m][ct][r], vm = v * wgt; s1[r] += vm; s2[r] += vm * v; }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
      if (n16 == 0) {
        const int chl = ch_local(ct, 4 * q);
        *reinterpret_cast<f32x4*>(red + (wave * NCT * 16 + chl) * 2) = f32x4{s1[0], s2[0], s1[1], s2[1]};
        *reinterpret_cast<f32x4*>(red + (wave * NCT * 16 + chl + 2) * 2) = f32x4{s1[2], s2[2], s1[3], s2[3]};
      }
    }
    // raw barrier (no fence): __syncthreads() would make hipcc drain the vector-memory queue, i.e. the prefetched DMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(red + (w * NCT * 16 + chl) * 2);
        t1 += v[0]; t2 += v[1];
      }
      const int stx = 2 * t.tx + strip;
      if (stx < a.tiles_x) {
        const size_t st = ((size_t)t.img * a.tiles_y + t.ty) * a.tiles_x + stx;
        typedef __attribute__((address_space(1))) f32x2_t gf32x2;  // global (not flat) store: see conv_mma_fast.hip
        *(gf32x2*)(a.stats + (st * nout + t.n0 + ch) * 2) = f32x2_t{t1, t2};
      }
    }
  };
  // accumulators of a new tile = the bias of the channel block (LDS copy fetched with the tile's first pieces)
  auto init_acc = [&]() __attribute__((always_inline)) {
    const int q = lane_id() >> 4;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_bias) bv = *reinterpret_cast<const f32x4*>(smem + G::BIAS + (wc * NCT * 16 + ch_local(ct, 4 * q)) * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][ct] = bv;
    }
  };

  // counted waits: N = operations that may stay outstanding, written for a wave that owns WPW weight pieces per step; a wave
  // that owns one less (WDUMMY: the piece count is not a multiple of 8) waits for one less.  NW = weight-piece rounds in N.
  const bool bigw = !G::WDUMMY || wave < (G::WPIECES % 8);  // uniform
#define WAIT_W(N) do { if (bigw) wait_vm<(N)>(); else wait_vm<((N) - 1)>(); } while (0)
  // ---- prologue of the workgroup: image of chunk 0, bias and the weights of steps 0 and 1 of its first item
  int w = blockIdx.x;
  BtWork cur = decode(w);
  set_img_state(cur);
  set_w_state(cur.n0);
  int gchunk = 0;     // chunks issued so far: image buffer = parity
#pragma unroll
  for (int j = 0; j < IPW; ++j) issue_img(0, 0, j);
  if (has_bias) issue_bias(cur.n0);
  issue_w(0, 0);
  issue_w(0, 1);
  WAIT_W(WPW);
  __builtin_amdgcn_s_barrier();

#ifdef CONV64_STAMPS
  unsigned long long z0 = 0, z1 = 0, z3 = 0, z4 = 0, d_cmp = 0, d_wait = 0, d_bar = 0, d_epi = 0, d_prep = 0, d_fin = 0, d_steps = 0, d_tiles = 0;
#endif
  auto no_row = [&](int) __attribute__((always_inline)) {};
  using F = std::false_type;
  using T = std::true_type;
  const bool want_stats = a.stats != nullptr;  // uniform
  bool first = true;
  while (true) {
    const int wnext = w + (int)gridDim.x;
    const bool has_next = wnext < nwork;  // uniform
    BtWork nxt = cur;
    init_acc();

    // one 32-channel chunk = three steps.  LAST = the tile's last chunk: the "following chunk" is chunk 0 of the next item (if
    // any), the bias of that item rides along, and the final step stores the tile row by row.
    auto chunk = [&](int c, auto lastc) __attribute__((always_inline)) {
      constexpr bool last = decltype(lastc)::value;
      const bool pre = !last || has_next;  // a following chunk exists
      const int cn = last ? 0 : c + 1;
      BSTAMP(z0);
      if (last && has_next) { nxt = decode(wnext); set_img_state(nxt); }
      BSTAMP(z1); BACC(d_prep, z1, z0);
      const unsigned imgoff = (gchunk & 1) * IMG_BYTES;
      const int nbuf = (gchunk + 1) & 1;
      const unsigned wlane = make_wlane();
      // step (c, 0): weights of (c, 2); first three image pieces of the following chunk
      compute(TB0{}, F{}, imgoff, [&](int k) __attribute__((always_inline)) {
        if (k < WPW) issue_w1(wlane, c, 2, k);
        else if (pre) issue_img(cn, nbuf, k - WPW);
      }, no_row);
      BSTAMP(z3); BACC(d_cmp, z3, z1);
      // the tile stores of the previous item (issued after everything this step reads) may stay in flight
      if (c == 0 && !first) { if (pre) WAIT_W(WPW + 3 + NSTORE); else WAIT_W(WPW + NSTORE); }
      else { if (pre) WAIT_W(WPW + 3); else WAIT_W(WPW); }
      BSTAMP(z4); BACC(d_wait, z4, z3);
      __builtin_amdgcn_s_barrier();
      BSTAMP(z1); BACC(d_bar, z1, z4);
      // step (c, 1): weights of (following chunk, 0), last two image pieces; in the last chunk the bias of the next item
      if (last && has_next && nxt.n0 != cur.n0) set_w_state(nxt.n0);
      const bool wbias = last && has_next && has_bias;
      compute(TB1{}, F{}, imgoff, [&](int k) __attribute__((always_inline)) {
        if (k < WPW) { if (pre) issue_w1(wlane, cn, 0, k); }
        else if (k < WPW + 2) { if (pre) issue_img(cn, nbuf, 3 + k - WPW); }
        else if (wbias) issue_bias(nxt.n0);
      }, no_row);
      BSTAMP(z3); BACC(d_cmp, z3, z1);
      if (!pre) wait_vm<0>(); else if (wbias) WAIT_W(WPW + 3); else WAIT_W(WPW + 2);
      BSTAMP(z4); BACC(d_wait, z4, z3);
      __builtin_amdgcn_s_barrier();
      BSTAMP(z1); BACC(d_bar, z1, z4);
      // step (c, 2): weights of (following chunk, 1); in the last chunk the tile is stored row by row as its rows complete
      OutState os;
      LaneStats ls;
      if (last) {
        os = out_state(cur);
        if (EARLY) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ls.s1[ct][r] = 0.f; ls.s2[ct][r] = 0.f; }
        }
      }
      compute(TB2{}, lastc, imgoff, [&](int k) __attribute__((always_inline)) {
        if (k < WPW && pre) issue_w1(wlane, cn, 1, k);
      }, [&](int m) __attribute__((always_inline)) {
        store_row(os, m);
        if (EARLY && want_stats) stats_row(ls, os, m);
      });
      BSTAMP(z3);
#ifdef CONV64_STAMPS
      if (last) d_fin += z3 - z1; else d_cmp += z3 - z1;
#endif
      // (last chunk) the tile stores are younger than every piece
      if (last) { if (pre) WAIT_W(WPW + NSTORE); else wait_vm<NSTORE>(); }
      else WAIT_W(WPW);
      BSTAMP(z4); BACC(d_wait, z4, z3);
      __builtin_amdgcn_s_barrier();
      BSTAMP(z1); BACC(d_bar, z1, z4);
#ifdef CONV64_STAMPS
      d_steps += 3;
#endif
      ++gchunk;
      if (last) {
        BSTAMP(z0);
        if (want_stats) tile_stats(cur, os, ls);
        BSTAMP(z1); BACC(d_epi, z1, z0);
      }
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk(c, F{});
    chunk(nchunks - 1, T{});
#ifdef CONV64_STAMPS
    d_tiles += 1;
    if (!has_next && (tid & 63) == 0 && blockIdx.x < 256) {
      unsigned long long* d = conv_bt_dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      d[0] = d_fin; d[1] = d_cmp; d[2] = d_wait; d[3] = d_bar; d[4] = d_epi; d[5] = d_prep; d[6] = d_steps; d[7] = d_tiles;
    }
#endif
    if (!has_next) break;
    cur = nxt; w = wnext; first = false;
  }
}

#undef WAIT_W

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

static int bt_num_cus() {  // one persistent workgroup per CU (device-properties cache: read once)
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

template <int WC, int NCT, int MT>
static void bt_launch(const ConvArgs& a, int nch, int order, int reserve, hipStream_t st) {
  const int tx32 = (a.tiles_x + 1) / 2;
  const int ptiles = a.N * a.tiles_y * tx32;
  const int nb = (a.o1 + a.o2) / nch;
  const int nwork = ptiles * nb;
  const int ncu = reserve > 0 ? persistent_cus(bt_num_cus(), reserve) : bt_num_cus();
  const int tile_major = (order != 0 && nb > 1 && ptiles % 8 == 0) ? nb : 0;
  hipLaunchKernelGGL((conv_bt_kernel<WC, NCT, MT>), dim3(nwork < ncu ? nwork : ncu), dim3(512), 0, st, a, ptiles, tx32, nwork, tile_major);
}

int conv_bt_launch(const ConvArgs& a, int order, int reserve, hipStream_t st) {
  const int nch = bt_nch(a);
  if (nch == 128) bt_launch<2, 4, 8>(a, nch, order, reserve, st);
  else if (nch == 96) bt_launch<2, 3, 8>(a, nch, order, reserve, st);
  else if (nch == 64) bt_launch<1, 4, 4>(a, nch, order, reserve, st);
  else return MIA_EARG;
  return MIA_OK;
}
