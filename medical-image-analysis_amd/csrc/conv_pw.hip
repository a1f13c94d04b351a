// ConvTranspose2d(kernel 2, stride 2) forward and input gradient for bf16, as ONE pointwise GEMM per launch on the LDS-DMA
// pattern of conv_bt.hip.  Reference layer: the decoder's upsampling `nn.ConvTranspose2d(c_below, c, 2, 2)`
// (src/models/unet/unet.py:142, applied at :212); its input gradient is what autograd runs for that module.
//
// A 2x2 / stride-2 transposed conv has no overlapping taps: every fine pixel (2i + ph, 2j + pw) is one tap (ph, pw) of the
// coarse pixel (i, j).  So
//   forward  (MODE_T2S2): out[p, tap * Cout + co] = bias[co] + sum_ci x[p, ci] W[tap][co][ci]        M = coarse pixels p = (n, i, j),
//                         K = Cin, N = 4 Cout; the store scatters column block `tap` to fine pixel (2i + ph, 2j + pw);
//   gradient (MODE_G2S2): dx[p, ci] = sum_tap sum_co dout[(n, 2i + ph, 2j + pw), co] W[tap][ci][co]  K = 4 Cout (the gather of the
//                         four fine pixels is a per-tap scalar offset of the row address), N = Cin.
//   strided-conv input gradient (MODE_T3S2, round 4): dx[(n, 2i + ph, 2j + pw), ci] = sum over the taps of parity class (ph, pw) of
//                         sum_co dy[(n, i + dh, j + dw), co] W[kh][kw][ci][co] -- the transposed form of the 3x3 / stride-2 conv
//                         (unet.py:54-66).  A class uses the window positions (dh, dw) with dh <= ph, dw <= pw (1 / 2 / 2 / 4 taps:
//                         kh = ph ? (dh ? 0 : 2) : 1, likewise kw): M = coarse pixels, N = 4 Cin columns class-major, K = positions
//                         of the block's classes x Cout.  A 128-column block is one class when Cin >= 128; at Cin = 64 it holds
//                         two classes and walks the union of their positions, the weight rows of a class that does not use a
//                         position arriving as zeros (out-of-range DMA source): 12 position-halves issued for 9 useful.  The
//                         store scatters to fine pixel (2i + ph, 2j + pw); with ConvArgs::acc_out it adds into the tensor (the
//                         skip gradient accumulated in place, mia_conv_mma_acc).
// The 256-thread tile kernel ran these as one-tap 32-channel chunks: 16 (forward) / 32 (gradient) MFMAs per wave between two
// barriers, the four parity classes as separate workgroups re-staging the same input tile.  Here ONE 512-thread workgroup per
// CU owns a 256-pixel x 128-column block and walks K in 64-wide stages through a ring of three LDS stages (A 32 KB + B 16 KB),
// every byte by LDS-DMA (`buffer_load_dwordx4 ... offen lds`, issued and counted in inline asm), one barrier per stage, the
// ring running across work items (the next item's first stages are in flight during the epilogue).  MFMA operands swapped
// (A = weights, B = pixels): a lane's accumulators are 8 consecutive channels of one pixel -> 16-byte stores straight from
// registers.  Per stage and wave: 6 DMA pieces, 16 ds_read_b128, 32 MFMAs.
//
// LDS images, 128 bytes per row (64 k), eight 16-byte units per row, unit u of row r stored at (u ^ f(r)); both are
// conflict-free for the 16-lane groups of ds_read_b128:
//   pixels : rows r .. r + 15 of one fragment: f(r) = (r >> 1) & 7
//   weights: fragment row i of channel fragment ct is local channel 32 (ct >> 1) + 8 (i >> 2) + 4 (ct & 1) + (i & 3) (so that the
//            accumulators of fragments (ct, ct + 1) are 8 consecutive channels): f(r) = ((r >> 1) & 1) | (((r >> 3) & 3) << 1)
// The DMA destination is lane-linear (one piece = 8 rows x 128 B), so the swizzle is applied on each lane's SOURCE address.
//
// Contract (conv_pw_eligible, otherwise the tile kernel runs): bf16, one source, one destination, no statistics; channels per
// tap of the input % 64 == 0, K >= 128, N % 128 == 0, npad / kpad unpadded, tensors < 4 GiB, 16-byte aligned pointers.
#include "conv_common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
#define PW_SENT 0xFFFFFFF0u /* always beyond num_records: loads return zero, stores are dropped */

__device__ __forceinline__ i32x4 pw_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long addr = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)addr);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(addr >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// One LDS-DMA piece: 64 lanes x 16 bytes, lane L lands at lds_dst + 16 L (see conv_bt.hip::dma16 for the M0 / s_nop notes).
__device__ __forceinline__ void pw_dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
  soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);        // wave-uniform by construction; tells hipcc so
  lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float pw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pw_pack(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pw_f32x2{a, b}, pw_bf16x2));
}

__device__ __forceinline__ float pw_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, left in every lane
  int iv;
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false));
  iv = __builtin_bit_cast(int, v); v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false));
  return v;
}

constexpr int PW_TM = 256, PW_TN = 128;
constexpr int PW_A = PW_TM * 128, PW_B = PW_TN * 128, PW_STAGE = PW_A + PW_B, PW_NSTAGE = 3;
constexpr int PW_BIAS = PW_NSTAGE * PW_STAGE;  // two 1 KB slots: the bias of the item being computed / being fetched
constexpr int PW_RED = PW_BIAS + 2 * 1024;     // [8 waves][64 channels][2] statistics exchange (strided 3x3 forward)
constexpr int PW_LDS = PW_RED + 8 * 64 * 8;
constexpr int PW_NSTORE = 8;                   // store instructions per wave and item

}  // namespace

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_pw_kernel(const ConvArgs a, int mtot, int wco, int nk, int mtiles, int nwork, int tile_major) {
  constexpr bool TR = (MODE == MODE_T2S2), S2 = (MODE == MODE_G3S2), T3 = (MODE == MODE_T3S2), BIAS = TR || S2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PW_LDS];
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;  // 64-pixel quarter, 64-column half of the block
  const int q = lane >> 4, r = lane & 15;
  const int cin = a.c1;                     // channels per tap of the A tensor
  const unsigned kpitch = (unsigned)a.kpad * 2u;
  const int cpt = cin >> 6;                 // 64-wide stages per tap (MODE_G2S2)

  const size_t in_bytes = (size_t)a.N * a.Hin * a.Win * a.c1 * 2, out_bytes = (size_t)a.N * a.Hout * a.Wout * a.o1 * 2;
  const i32x4 rsA = pw_rsrc_words(a.in1, (unsigned)in_bytes);
  const i32x4 rsW = pw_rsrc_words(a.wp, (unsigned)((size_t)4 * a.npad * a.kpad * 2));
  const i32x4 rsB = pw_rsrc_words(BIAS ? (const void*)a.bias : a.wp, BIAS ? (unsigned)(a.o1 * 4) : 0u);
  const i32x4 rsW9 = pw_rsrc_words(a.wp, (unsigned)((size_t)9 * a.npad * a.kpad * 2));  // S2: nine taps
  const rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.out1, 0, (int)(unsigned)out_bytes, 0x00020000);

  struct Item { int m0, n0; };
  // T3: parity classes of the block's two 64-column halves, the window positions they use (bit pos = 2 dh + dw) and the stage count
  auto t3_cls = [&](int n0, int half) __attribute__((always_inline)) -> int { return (n0 + 64 * half) / a.o1; };
  auto t3_posmask = [&](int n0) __attribute__((always_inline)) -> unsigned {
    unsigned m = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c = t3_cls(n0, half);
      m |= 1u | ((c & 1) ? 2u : 0u) | ((c & 2) ? 4u : 0u) | ((c == 3) ? 8u : 0u);
    }
    return m;
  };
  auto t3_nk = [&](int n0) __attribute__((always_inline)) -> int { return __builtin_popcount(t3_posmask(n0)) * cpt; };
  auto t3_tap = [&](int cls, int pos) __attribute__((always_inline)) -> int {  // weight tap of (class, position), -1 = the class skips it
    const int ph = cls >> 1, pw = cls & 1, dh = pos >> 1, dw = pos & 1;
    if (dh > ph || dw > pw) return -1;
    const int kh = ph ? (dh ? 0 : 2) : 1, kw = pw ? (dw ? 0 : 2) : 1;
    return kh * 3 + kw;
  };
  // T3: a workgroup owns whole pixel tiles (tile = blockIdx.x, + gridDim.x, ...) and walks ALL column blocks of a tile back to back:
  // the parity classes cost 1 / 2 / 2 / 4 taps, so any order that pins a workgroup to one column block (the XCD-grouped order
  // does: 32 slots per round, 16 blocks) leaves the four-tap workgroups running 1.8x longer than the average; here every workgroup
  // does every class, and a tile's dy rows are re-read from the same CU's L2 sixteen times in a row.  `tile_major` = column blocks.
  auto t3_live = [&](int w) __attribute__((always_inline)) -> bool {
    const int k = (w - (int)blockIdx.x) / (int)gridDim.x;
    return (int)blockIdx.x + (k / tile_major) * (int)gridDim.x < mtiles;
  };
  auto decode = [&](int w) __attribute__((always_inline)) -> Item {
    int cb, tile;
    if (T3) {
      const int k = (w - (int)blockIdx.x) / (int)gridDim.x, kt = k / tile_major;
      cb = k - kt * tile_major;
      tile = (int)blockIdx.x + kt * (int)gridDim.x;
    } else if (tile_major > 0) {  // the column blocks of a pixel tile take consecutive slots of one XCD (its A rows stay in that L2)
      const int slot = w >> 3, grp = slot / tile_major;
      cb = slot - grp * tile_major;
      tile = grp * 8 + (w & 7);
    } else {
      cb = w / mtiles;
      tile = w - cb * mtiles;
    }
    return Item{tile * PW_TM, cb * PW_TN};
  };

  // ---- issue side: the DMA lane offsets of the item being fetched
  const int lr = lane >> 3;
  const unsigned uA = (unsigned)((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));
  const unsigned uB = (unsigned)((lane & 7) ^ (((lane >> 4) & 1) | ((wave & 3) << 1)));
  unsigned va[4], vb[2], vbias = PW_SENT, vmask = 0;
  int i_n0 = 0, i_nk = nk, ipos = 0;  // T3: the fetched item's column base, stage count, current window position
  unsigned i_pm = 1;                  //     and remaining-position mask
  auto setup_issue = [&](int w) __attribute__((always_inline)) {
    const Item it = decode(w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = it.m0 + 8 * wave + 64 * k + lr;
      unsigned base;
      if (TR) base = (unsigned)p * (unsigned)(cin * 2);
      else if (S2) {  // output pixel (n, i, j) reads fine pixels (2i + ta - 1, 2j + tb - 1): base = pixel (2i, 2j), six validity bits
        const int t1 = p / wco, j = p - t1 * wco, n = t1 / a.Hout, i = t1 - n * a.Hout;
        base = (unsigned)((n * a.Hin + 2 * i) * a.Win + 2 * j) * (unsigned)(cin * 2);
        const unsigned bits = (i > 0 ? 1u : 0u) | 2u | (2 * i + 1 < a.Hin ? 4u : 0u) | (j > 0 ? 8u : 0u) | 16u | (2 * j + 1 < a.Win ? 32u : 0u);
        vmask = k == 0 ? bits : (vmask | (bits << (6 * k)));
      } else if (T3) {  // coarse pixel (n, i, j) itself; positions (dh, dw) add a row / a column: two validity bits
        const int t1 = p / wco, j = p - t1 * wco, i = t1 % a.Hin;
        base = (unsigned)p * (unsigned)(cin * 2);
        const unsigned bits = (i + 1 < a.Hin ? 1u : 0u) | (j + 1 < a.Win ? 2u : 0u);
        vmask = k == 0 ? bits : (vmask | (bits << (2 * k)));
      } else { const int j = p % wco; base = (unsigned)(4 * p - 2 * j) * (unsigned)(cin * 2); }  // fine pixel (n, 2i, 2j)
      va[k] = p < mtot ? base + uA * 16u : PW_SENT;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int row = it.n0 + 8 * wave + 64 * k + lr;
      if (T3) row -= t3_cls(it.n0, k) * a.o1;  // channel within its parity class (weights are [tap][Cin][Cout])
      vb[k] = (unsigned)row * kpitch + uB * 16u;
    }
    if (T3) { i_n0 = it.n0; i_nk = t3_nk(it.n0); i_pm = t3_posmask(it.n0); ipos = __builtin_ctz(i_pm); }
    if (BIAS) vbias = lane < 32 ? (unsigned)((it.n0 + 4 * lane) % a.o1) * 4u : PW_SENT;
  };
  int wi = blockIdx.x, ki = 0, itap = 0, icc = 0, ita = 0, itb = 0, islot = 0, ibias = 0;
  bool ihave = T3 ? t3_live(wi) : wi < nwork;
  if (ihave) setup_issue(wi);
  auto issue_next = [&]() __attribute__((always_inline)) -> int {
    if (!ihave) return 0;
    unsigned soffA, soffB;
    if (TR) { soffA = (unsigned)ki * 128u; soffB = soffA; }
    else if (S2) { soffA = 0u; soffB = (unsigned)itap * (unsigned)a.npad * kpitch + (unsigned)icc * 128u; }
    else if (T3) { soffA = (unsigned)(((ipos >> 1) * a.Win + (ipos & 1)) * (cin * 2)) + (unsigned)icc * 128u; soffB = (unsigned)icc * 128u; }
    else {
      soffA = (unsigned)((itap >> 1) * a.Win + (itap & 1)) * (unsigned)(cin * 2) + (unsigned)icc * 128u;
      soffB = (unsigned)itap * (unsigned)a.npad * kpitch + (unsigned)icc * 128u;
    }
    const unsigned dst = lds0 + (unsigned)islot * PW_STAGE + (unsigned)wave * 1024u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned v = va[k];
      if (S2) {  // tap (ita, itb): a signed pixel offset, rows / columns outside the image read zeros
        const int toff = ((ita - 1) * a.Win + (itb - 1)) * (cin * 2) + icc * 128;
        const bool ok = ((vmask >> (6 * k + ita)) & (vmask >> (6 * k + 3 + itb)) & 1u) != 0u && v != PW_SENT;
        v = ok ? (unsigned)((int)v + toff) : PW_SENT;
      }
      if (T3) {  // position (dh, dw): the row below / the column to the right must exist
        const unsigned vb2 = vmask >> (2 * k);
        const bool ok = (!(ipos & 2) || (vb2 & 1u)) && (!(ipos & 1) || (vb2 & 2u));
        v = ok ? v : PW_SENT;
      }
      pw_dma16(rsA, v, soffA, dst + (unsigned)k * 8192u);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (T3) {  // the half's class picks the tap of this position, or zeros when it has none
        const int tap = t3_tap(t3_cls(i_n0, k), ipos);
        pw_dma16(rsW9, tap >= 0 ? vb[k] : PW_SENT, soffB + (unsigned)(tap >= 0 ? tap : 0) * (unsigned)a.npad * kpitch, dst + PW_A + (unsigned)k * 8192u);
      } else pw_dma16(S2 ? rsW9 : rsW, vb[k], soffB, dst + PW_A + (unsigned)k * 8192u);
    }
    int cnt = 6;
    if (BIAS && ki == 0) { pw_dma16(rsB, vbias, 0u, lds0 + PW_BIAS + (unsigned)ibias * 1024u); ibias ^= 1; cnt = 7; }
    islot = islot == PW_NSTAGE - 1 ? 0 : islot + 1;
    ++ki;
    if (!TR) {
      if (++icc == cpt) {
        icc = 0; ++itap;
        if (S2 && ++itb == 3) { itb = 0; ++ita; }
        if (T3) { i_pm &= i_pm - 1; ipos = i_pm ? __builtin_ctz(i_pm) : 0; }
      }
    }
    if (ki == (T3 ? i_nk : nk)) {
      ki = 0; itap = 0; icc = 0; ita = 0; itb = 0;
      wi += gridDim.x;
      ihave = T3 ? t3_live(wi) : wi < nwork;
      if (ihave) setup_issue(wi);
    }
    return cnt;
  };

  // ---- compute side: fragment addresses (stage-relative); k-step 1 of a stage is the same address ^ 64
  const unsigned pfrag = (unsigned)((wm * 64 + r) * 128 + ((q ^ ((r >> 1) & 7)) * 16));
  const unsigned wfrag = (unsigned)(PW_A + (wn * 64 + 8 * (r >> 2) + (r & 3)) * 128 + ((q ^ (((r & 3) >> 1) | ((r >> 2) << 1))) * 16));

  issue_next();
  issue_next();  // nk >= 2: the first item's second stage, 6 pieces
  pw_wait_vm<6>();
  __builtin_amdgcn_s_barrier();

  int cslot = 0, cbias = 0, st_m0 = 0, st_n0 = 0;
  bool after_epilogue = false, stats_pending = false;
  for (int wcur = blockIdx.x; T3 ? t3_live(wcur) : wcur < nwork; wcur += gridDim.x) {
    const Item it = decode(wcur);
    f32x4 acc[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) acc[ct][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_cur = T3 ? t3_nk(it.n0) : nk;
    for (int kc = 0; kc < nk_cur; ++kc) {
      const int cnt = issue_next();
      const unsigned char* st = smem + cslot * PW_STAGE;
      u32x4 wf[2][4], pfr[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          wf[ks][ct] = *reinterpret_cast<const u32x4*>(st + ((wfrag + (unsigned)((32 * (ct >> 1) + 4 * (ct & 1)) * 128)) ^ (unsigned)(ks * 64)));
#pragma unroll
        for (int pf = 0; pf < 4; ++pf)
          pfr[ks][pf] = *reinterpret_cast<const u32x4*>(st + ((pfrag + (unsigned)(pf * 2048)) ^ (unsigned)(ks * 64)));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int pf = 0; pf < 4; ++pf)
            acc[ct][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][ct]), __builtin_bit_cast(bf16x8, pfr[ks][pf]),
                                                                  acc[ct][pf], 0, 0, 0);
      cslot = cslot == PW_NSTAGE - 1 ? 0 : cslot + 1;
      // own pieces of the stage after next may stay in flight (and the previous item's stores, issued between the two)
      if (cnt == 0) pw_wait_vm<0>();
      else if (after_epilogue) { if (cnt == 7) pw_wait_vm<7 + PW_NSTORE>(); else pw_wait_vm<6 + PW_NSTORE>(); }
      else { if (cnt == 7) pw_wait_vm<7>(); else pw_wait_vm<6>(); }
      after_epilogue = false;

      if (kc == nk_cur - 1) {  // ---- epilogue: 8 consecutive channels per lane and fragment pair, 16-byte stores
        const int ncol = it.n0 + wn * 64;              // first GEMM column of this wave
        int tap = 0, co0 = ncol;
        if (TR || T3) { tap = ncol / a.o1; co0 = ncol - tap * a.o1; }  // T3: `tap` = the wave's parity class (ph, pw)
        const unsigned tapoff = TR ? (unsigned)((tap >> 1) * a.Wout + (tap & 1)) : 0u;
        unsigned t3pix[4];
        if (T3) {
#pragma unroll
          for (int pf = 0; pf < 4; ++pf) {
            const int p = it.m0 + wm * 64 + pf * 16 + r;
            const int t1 = p / wco, j = p - t1 * wco, n = t1 / a.Hin, i = t1 - n * a.Hin;
            const int oy = 2 * i + (tap >> 1), ox = 2 * j + (tap & 1);
            t3pix[pf] = (p < mtot && oy < a.Hout && ox < a.Wout) ? (unsigned)((n * a.Hout + oy) * a.Wout + ox) : 0xFFFFFFFFu;
          }
        }
        // accumulate mode: all eight previous 16-byte units of this lane are loaded before the first store (the compiler cannot move a
        // load above a store itself).  The queue is drained first, so the compiler's own counted waits for these loads see only
        // operations it knows about (the LDS-DMA pieces are invisible to it).
        u32x4 prev[2][4];
        if (T3 && a.acc_out) {
          pw_wait_vm<0>();
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) {
              const unsigned vo = t3pix[pf] != 0xFFFFFFFFu ? (t3pix[pf] * (unsigned)a.o1 + (unsigned)(co0 + 32 * pr + 8 * q)) * 2u : PW_SENT;
              prev[pr][pf] = __builtin_amdgcn_raw_buffer_load_b128(rsO, (int)vo, 0, 0);
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float bv[8], s1[8], s2[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
          if (BIAS) {
            const float* bl = reinterpret_cast<const float*>(smem + PW_BIAS + cbias * 1024) + wn * 64 + 32 * pr + 8 * q;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl), b1 = *reinterpret_cast<const f32x4*>(bl + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
          }
#pragma unroll
          for (int pf = 0; pf < 4; ++pf) {
            const int p = it.m0 + wm * 64 + pf * 16 + r;
            unsigned pix;
            if (TR) { const int j = p % wco; pix = (unsigned)(4 * p - 2 * j) + tapoff; }
            else pix = (unsigned)p;
            unsigned voff = p < mtot ? (pix * (unsigned)a.o1 + (unsigned)(co0 + 32 * pr + 8 * q)) * 2u : PW_SENT;
            if (T3) voff = t3pix[pf] != 0xFFFFFFFFu ? (t3pix[pf] * (unsigned)a.o1 + (unsigned)(co0 + 32 * pr + 8 * q)) * 2u : PW_SENT;
            const f32x4 lo = acc[2 * pr][pf], hi = acc[2 * pr + 1][pf];
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bv[e]; v[4 + e] = hi[e] + bv[4 + e]; }
            if (T3 && a.acc_out) {  // out += result: the skip tensor's first gradient piece is already there
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += __builtin_bit_cast(float, prev[pr][pf][e] << 16);
                v[2 * e + 1] += __builtin_bit_cast(float, prev[pr][pf][e] & 0xFFFF0000u);
              }
            }
            if (S2 && a.stats != nullptr) {  // whole tiles only (contract): every pixel counts
#pragma unroll
              for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
            u32x4 d;
            d.x = pw_pack(v[0], v[1]); d.y = pw_pack(v[2], v[3]); d.z = pw_pack(v[4], v[5]); d.w = pw_pack(v[6], v[7]);
            __builtin_amdgcn_raw_buffer_store_b128(d, rsO, (int)voff, 0, 0);
          }
          if (S2 && a.stats != nullptr) {  // per-wave sums of its 64 pixels: DPP row sums over the 16 pixel lanes
            float* red = reinterpret_cast<float*>(smem + PW_RED) + (wave * 64 + 32 * pr + 8 * q) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t1 = pw_row16_sum(s1[e]), t2 = pw_row16_sum(s2[e]);
              if (r == 0) { red[2 * e] = t1; red[2 * e + 1] = t2; }
            }
          }
        }
        if (S2 && a.stats != nullptr) { stats_pending = true; st_m0 = it.m0; st_n0 = it.n0; }
        cbias ^= 1;
        after_epilogue = true;
      }
      __builtin_amdgcn_s_barrier();
      if (S2 && stats_pending) {
        // the host's statistics tiles are 8 x 16 pixels = 128 consecutive coarse pixels when whole rows tile evenly (contract): the
        // 256-pixel block is two entries, wave quarters {0, 1} and {2, 3}
        stats_pending = false;
        if (tid < 256) {
          const int half = tid >> 7, wnn = (tid >> 6) & 1, ch = tid & 63;
          const float* red = reinterpret_cast<const float*>(smem + PW_RED);
          const float* ra = red + ((wnn * 4 + 2 * half) * 64 + ch) * 2;
          const float* rb = ra + 64 * 2;
          typedef float f2_t __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(1))) f2_t gf2;
          gf2* dst = (gf2*)(a.stats + ((size_t)(st_m0 / 128 + half) * a.o1 + st_n0 + wnn * 64 + ch) * 2);
          *dst = f2_t{ra[0] + rb[0], ra[1] + rb[1]};
        }
      }
    }
  }
}

bool conv_pw_eligible(int mode, int dtype, const ConvArgs& a) {
  // (MODE_T3S2, the strided conv's input gradient as exact-tap GEMMs, round 4: correct, slower than the tile kernel at every level --
  // profiles/r04_ab_conv_pw_t3.txt -- and no longer instantiated; its branches stay in the kernel template as documentation of the attempt)
  if (dtype != MIA_BF16 || (mode != MODE_T2S2 && mode != MODE_G2S2 && mode != MODE_G3S2)) return false;
  if (a.c2 != 0 || a.o2 != 0 || !a.vec_in || !a.vec_out) return false;
  if (a.c1 % 64 != 0 || a.kpad != a.c1 || a.npad != a.o1) return false;
  const int ntot = (mode == MODE_T2S2 || mode == MODE_T3S2) ? 4 * a.o1 : a.o1;
  const int ktot = (mode == MODE_T2S2 || mode == MODE_T3S2) ? a.c1 : (mode == MODE_G3S2 ? 9 : 4) * a.c1;  // (T3: the one-tap class)
  if (ntot % PW_TN != 0 || ktot < 128 || a.o1 % 64 != 0) return false;
  if ((mode == MODE_T2S2 || mode == MODE_G3S2) != (a.bias != nullptr)) return false;  // the forwards carry the bias, the gradients none
  if (a.acc_out && mode != MODE_T3S2) return false;
  if (mode == MODE_T3S2 && !(a.Hin == (a.Hout + 1) / 2 && a.Win == (a.Wout + 1) / 2 && a.flip == 0)) return false;
  if (mode == MODE_G3S2) {
    // statistics: the finalize step only sums an image's entries, so any partition of its pixels into the host's tiles_y * tiles_x
    // entries will do -- here runs of 128 consecutive output pixels (two per block); needs whole 8 x 16 tilings and no block
    // straddling two images
    if (a.stats != nullptr && !(a.Wout % 16 == 0 && a.Hout % 8 == 0 && (a.Hout * a.Wout) % PW_TM == 0)) return false;
  } else if (a.stats != nullptr) return false;
  const size_t lim = ((size_t)1 << 32) - ((size_t)1 << 20);
  if ((size_t)a.N * a.Hin * a.Win * a.c1 * 2 >= lim || (size_t)a.N * a.Hout * a.Wout * a.o1 * 2 >= lim) return false;
  if ((size_t)9 * a.npad * a.kpad * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

static int pw_num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

int conv_pw_launch(int mode, const ConvArgs& a, int reserve, hipStream_t st) {
  const bool tr = mode == MODE_T2S2, t3 = mode == MODE_T3S2;
  const int hco = (tr || t3) ? a.Hin : a.Hout, wco = (tr || t3) ? a.Win : a.Wout;  // the coarse grid
  const int mtot = a.N * hco * wco;
  const int nk = ((tr || t3) ? a.c1 : (mode == MODE_G3S2 ? 9 : 4) * a.c1) / 64;  // (t3: per item, 1 .. 4 x this -- the kernel works it out)
  const int nblocks = ((tr || t3) ? 4 * a.o1 : a.o1) / PW_TN;
  const int mtiles = ceil_div(mtot, PW_TM);
  const int nwork = mtiles * nblocks;
  const int tile_major = t3 ? nblocks : ((nblocks > 1 && mtiles % 8 == 0) ? nblocks : 0);
  const int ncu = reserve > 0 ? persistent_cus(pw_num_cus(), reserve) : pw_num_cus();
  const dim3 grid(t3 ? (mtiles < ncu ? mtiles : ncu) : (nwork < ncu ? nwork : ncu));  // t3: one workgroup per pixel tile at most
  if (t3) { mia_set_error("conv_pw: MODE_T3S2 is not built"); return MIA_EUNSUPPORTED; }
  else if (mode == MODE_G3S2) hipLaunchKernelGGL(conv_pw_kernel<MODE_G3S2>, grid, dim3(512), 0, st, a, mtot, wco, nk, mtiles, nwork, tile_major);
  else if (tr) hipLaunchKernelGGL(conv_pw_kernel<MODE_T2S2>, grid, dim3(512), 0, st, a, mtot, wco, nk, mtiles, nwork, tile_major);
  else hipLaunchKernelGGL(conv_pw_kernel<MODE_G2S2>, grid, dim3(512), 0, st, a, mtot, wco, nk, mtiles, nwork, tile_major);
  return MIA_OK;
}
