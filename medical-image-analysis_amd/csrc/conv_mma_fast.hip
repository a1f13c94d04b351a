// Branch-free fast path of the implicit-GEMM conv (same math / tiling / LDS layout as conv_mma.hip).
//
// The generic kernel spends most of its issue slots on per-element predicates (image border, channel
// tails, two-source select) -- rocprofv3 showed SQ_ACTIVE_INST_ANY ~3x the MFMA issue time.  Here every
// global access is a raw buffer load/store against a per-image descriptor, so the image border and the
// tile tail become an out-of-range offset that the hardware turns into a zero (loads) or a dropped write
// (stores); channel chunks never straddle a source; all staging addresses are "thread constant +
// compile-time immediate"; the epilogue writes LDS at immediate offsets.  Contract (checked on the host,
// otherwise the generic kernel runs): c1 % KB == 0, c2 % KB == 0, o1 % EPU == 0, o2 % EPU == 0, 16-byte aligned
// pointers, per-image tensors < 2 GiB.
#include <type_traits>

#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define SENT 0xFFFFFFF0u /* always beyond num_records */

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// WC = 2 (stride-2 3x3 forward, bf16): 512 threads, the eight waves are 4 row groups x 2 halves of a 128-channel block on a
// 16-row tile -- each wave owns 4 x 4 accumulators (8 fragment reads per 16 MFMAs instead of 6 per 8: the 256-thread
// 8-row shape saturates the LDS pipe), one workgroup per CU.  The statistics stay in the host's 8-row tile layout.
// SPLIT (T = float only): fp32 tensors, products on the f16 matrix cores from two-part split operands (common.h SplitF16): commit()
// turns every staged fp32 unit, scaled by its tensor's power of two (a.amax_in1 / amax_in2 / amax_w: device pointers to max |x| as fp32
// bit patterns), into (h | l) words, an MFMA step is two 16x16x32 f16 instructions, the epilogue scales the accumulators back.
// SPLIT = 2 (even MT and NT): THREE products on separate h / l planes -- x w = h H + h L + l H, the dropped l L is <= 2^-24 |x w|, the size
// of the parts' own rounding.  A 16-channel chunk of one plane is exactly the K = 16 of v_mfma_f32_32x32x16_f16, so a wave's MT x NT
// 16 x 16 tiles become (MT / 2) x (NT / 2) 32 x 32 tiles (two image rows -- m and m + MT / 2 -- x 16 pixels by 32 output channels) and a chunk costs
// 3 x 32 matrix cycles per 32 x 32 outputs instead of 8 x 16: -25 %, with no VALU work in the loop (the interleaved form rotates every B
// fragment).  LDS holds the same bytes: unit planes [h ch 0-7 | h ch 8-15 | l ch 0-7 | l ch 8-15] instead of four 4-channel word planes.
template <typename T, int MODE, int MT, int NT, int WC = 1, int SPLIT = 0>
__global__ __launch_bounds__(256 * WC, (MT >= 8 || WC == 2 ? 1 : 2)) void conv_mma_fast_kernel(const ConvArgs a) {
  static_assert(!SPLIT || sizeof(T) == 4, "split mode is a mode of the fp32 kernel");
  constexpr bool P = SPLIT == 2;  // planar three-product form
  static_assert(!P || (MT % 2 == 0 && NT % 2 == 0 && WC == 1), "32 x 32 tiles");
  constexpr int MB = P ? MT / 2 : 1, NB = P ? NT / 2 : 1;
  using G = Geo<MODE, MT>;
  constexpr int NTHR = 256 * WC, PL = 64 * WC;  // threads; pixel lanes (x 4 channel groups) of a staging iteration
  constexpr int TH = G::TH, BN = 16 * NT * WC, EPU = Elem<T>::EPU, KB = 4 * EPU, ES = (int)sizeof(T);
  constexpr int PITCH = G::PITCH, S = G::S, IW = G::IW, IH = G::IH;
  constexpr int A_IT = (IH * IW + PL - 1) / PL;  // PL pixels (x 4 channel groups) per staging iteration
  constexpr int NPIX_ALLOC = (S == 1) ? (A_IT * PL > G::NPIX ? A_IT * PL : G::NPIX) : G::NPIX + 1;  // S==2: +1 dummy slot
  // (planar form: == 4 (mod 8) -- the two channel halves a staging quad writes with ds_write_b64 land 16 of the stores' 32 banks apart)
  constexpr int NPA = P ? ((NPIX_ALLOC + 3) / 8) * 8 + 4 : ((NPIX_ALLOC + 13) / 16) * 16 + 2;
  constexpr int NPB = P ? BN + 4 : BN + 2;
  constexpr int TPI = PL / BN;  // taps staged per iteration
  constexpr int B_IT = (G::MAXTAPS + TPI - 1) / TPI;
  constexpr int OSTR = BN + EPU;
  constexpr int A_UNITS = 4 * NPA, B_UNITS = B_IT * TPI * 4 * NPB;
  constexpr int STAGE_BYTES = (A_UNITS + B_UNITS) * 16;
  constexpr int OUT_BYTES = TH * 16 * OSTR * ES;
  constexpr int LDS_BYTES = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
  static_assert(PL % BN == 0 && TPI >= 1, "weight staging deals whole taps");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + 2 * 4 * BN * 4];
  u32x4* ldsA = reinterpret_cast<u32x4*>(smem);
  u32x4* ldsB = ldsA + A_UNITS;
  T* ldsO = reinterpret_cast<T*>(smem);
  float* ldsR = reinterpret_cast<float*>(smem + LDS_BYTES);
  __shared__ unsigned ldsM[2];  // block maxima of |out1| / |out2| (a.amax_out*)
  const bool want_amax = sizeof(T) == 4 && (a.amax_out1 != nullptr || a.amax_out2 != nullptr);  // uniform
  if (want_amax && threadIdx.x < 2) ldsM[threadIdx.x] = 0u;  // (the K loop's barriers order this before the epilogue's atomics)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3, wc = tid >> 8;  // row group; channel half (WC == 2)
  const int q = lane >> 4, r16 = lane & 15;
  const int pr = pi16(r16);
  const int g = tid & 3, p4 = tid >> 2;
  // planar form: MFMA row r32 of a 32 x 32 tile is pixel (image row r32 >> 4, column pi16(r32 & 15)): a ds_read_b128 is served in the
  // lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), i.e. eight lanes of each image row, and with the rows PITCH == 2 (mod 16)
  // slots apart the odd / even split of pi16 keeps a group's sixteen 16-byte slots distinct (the 16 x 16 form's reason for pi16 too);
  // kh = channel half of the chunk
  const int r32 = lane & 31, kh = lane >> 5;
  const int prow = r32 >> 4, ppx = pi16(r32 & 15);

  constexpr bool TMODE = (MODE == MODE_T3S2 || MODE == MODE_T2S2);
  // Block order.  Workgroups are dealt to the 8 XCDs round robin by linear id, and each XCD has its own L2: the blocks that
  // read the same input tile (the output-channel blocks of a tile, times the 4 output-parity classes of a transposed mode)
  // take consecutive slots of ONE XCD, so the tile is fetched into one L2 once instead of into up to eight (and, for the
  // parity classes, instead of once per pass over the whole grid).
  int bid = blockIdx.x, par = TMODE ? (int)blockIdx.y : 0, nb;
  if (a.xcd) {
    const int G = a.nblk_n * (TMODE ? 4 : 1);
    const int slot = bid >> 3, grp = slot / G, w = slot - grp * G;
    bid = grp * 8 + (bid & 7);
    if (bid >= a.N * a.tiles_x * a.tiles_y) return;
    nb = w % a.nblk_n; par = w / a.nblk_n;
  } else {
    nb = bid % a.nblk_n; bid /= a.nblk_n;
  }
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid;
  const int n0 = nb * BN;
  const int ph = par >> 1, pw = par & 1;
  const int oy0 = ty * TH, ox0 = tx * 16;
  const int iy0 = MODE == MODE_G3S1 ? oy0 - 1 : MODE == MODE_G3S2 ? 2 * oy0 - 1 : MODE == MODE_G2S2 ? 2 * oy0 : oy0;
  const int ix0 = MODE == MODE_G3S1 ? ox0 - 1 : MODE == MODE_G3S2 ? 2 * ox0 - 1 : MODE == MODE_G2S2 ? 2 * ox0 : ox0;

  const int nth = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : MODE == MODE_T3S2 ? (ph ? 2 : 1) : 1;
  const int ntw = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : MODE == MODE_T3S2 ? (pw ? 2 : 1) : 1;
  const int ntaps = nth * ntw;
  auto tap_w = [&](int ta, int tb) -> int {
    if (MODE == MODE_G3S1) { const int t = ta * 3 + tb; return a.flip ? 8 - t : t; }
    if (MODE == MODE_G3S2) return ta * 3 + tb;
    if (MODE == MODE_G2S2) return ta * 2 + tb;
    if (MODE == MODE_T3S2) { const int kh = ph ? (ta == 0 ? 0 : 2) : 1; const int kw = pw ? (tb == 0 ? 0 : 2) : 1; return kh * 3 + kw; }
    if (MODE == MODE_T2S2) return ph * 2 + pw;
    return 0;
  };
  auto tap_off = [&](int ta, int tb) -> int {
    if (MODE == MODE_G3S1) return ta * PITCH + tb;
    if (MODE == MODE_G3S2 || MODE == MODE_G2S2) return ta * PITCH + (tb & 1) * G::IWH + (tb >> 1);
    if (MODE == MODE_T3S2) { const int dh = ph ? (ta == 0 ? 1 : 0) : 0; const int dw = pw ? (tb == 0 ? 1 : 0) : 0; return dh * PITCH + dw; }
    return 0;
  };

  const int ctot = a.c1 + a.c2;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const size_t ipix = (size_t)a.Hin * a.Win;
  const rsrc_t rs1 = make_rsrc(in1 + (size_t)img * ipix * a.c1, (unsigned)(ipix * a.c1 * ES));
  const rsrc_t rs2 = make_rsrc(a.c2 ? in2 + (size_t)img * ipix * a.c2 : in1, (unsigned)(ipix * (a.c2 ? a.c2 : 0) * ES));
  constexpr int WTAPS = (MODE == MODE_G3S1 || MODE == MODE_G3S2 || MODE == MODE_T3S2) ? 9 : (MODE == MODE_G1 ? 1 : 4);
  const rsrc_t rsw = make_rsrc(a.wp, (unsigned)((size_t)WTAPS * a.npad * a.kpad * ES));

  // ---- thread-constant staging addresses
  unsigned a_pixoff[A_IT];
  unsigned a_valid = 0;
  int a_lds[S == 2 ? A_IT : 1];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int pix = p4 + PL * i;
    const int iy = pix / IW, ix = pix - iy * IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    const bool ok = pix < IH * IW && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    a_pixoff[i] = (unsigned)(gy * a.Win + gx);
    a_valid |= (ok ? 1u : 0u) << i;
    if (S == 2) a_lds[i] = pix < IH * IW ? iy * PITCH + (ix & 1) * G::IWH + (ix >> 1) : G::NPIX;
  }
  const int bn_ = p4 % BN, tsub = p4 / BN;
  // weight staging offsets.  3x3 / stride 1 with a 64-channel output block stages exactly one tap per iteration: the tap
  // stride is uniform and rides in the load's scalar offset, so one thread-dependent offset serves all nine loads
  constexpr bool LINB = ((MODE == MODE_G3S1 || MODE == MODE_G3S2) && TPI == 1);
  const int wtap_bytes = a.npad * a.kpad * ES;
  unsigned b_voff[LINB ? 1 : B_IT];
  if constexpr (LINB) {
    b_voff[0] = (unsigned)((((size_t)n0 + bn_) * a.kpad + g * EPU) * ES);
  } else {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int tl = i * TPI + tsub;
      const int ta = tl / ntw, tb = tl - ta * ntw;
      b_voff[i] = tl < ntaps ? (unsigned)((((size_t)tap_w(ta, tb) * a.npad + n0 + bn_) * a.kpad + g * EPU) * ES) : SENT;
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 accp[MB][NB];  // (planar form; acc[][] is unused there and vice versa)
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) accp[m][n][j] = 0.f;
  // the bias is loaded up front: at the top of the epilogue it would expose a full memory round trip per block
  const int nout = a.o1 + a.o2;
  float bv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int bi = P ? n0 + (n >> 1) * 32 + r32 : n0 + (wc * NT + n) * 16 + pr;  // (planar form: bv[2 * nb] is the lane's channel of block nb)
    bi = bi < nout ? bi : nout - 1;
    bv[n] = a.bias ? a.bias[bi] : 0.f;
  }

  u32x4 pa[A_IT], pb[B_IT];
  // split mode: operand scales 2^ea (activations: the larger maximum of the two sources, one reduction dimension), 2^eb (weights)
  // (the three scalar loads are issued here, their arithmetic sits behind the first tile's vector loads: a workgroup lives for one
  // tile, ~10 us, and a scalar round trip in front of its first fetch was 8 % of the launch)
  float sc_a = 1.f, sc_b = 1.f;
  int e_out = 0;
  unsigned raw_a = 0u, raw_a2 = 0u, raw_w = 0u;
  if constexpr (SPLIT) { raw_a = *a.amax_in1; raw_a2 = a.c2 ? *a.amax_in2 : 0u; raw_w = *a.amax_w; }
  auto fetch = [&](int c0) {
    const bool second = c0 >= a.c1;  // uniform: chunks never straddle the two sources
    const rsrc_t rs = second ? rs2 : rs1;
    const unsigned cs_es = (unsigned)((second ? a.c2 : a.c1) * ES);
    const int soff = (second ? c0 - a.c1 : c0) * ES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const unsigned voff = ((a_valid >> i) & 1u) ? a_pixoff[i] * cs_es + (unsigned)(g * 16) : SENT;
      pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if constexpr (LINB)
        pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)b_voff[0], c0 * ES + ((MODE == MODE_G3S1 && a.flip) ? 8 - i : i) * wtap_bytes, 0);
      else
        pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)b_voff[i], c0 * ES, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int at = g * NPA + (S == 2 ? a_lds[i] : p4 + PL * i);
      if constexpr (P) {
        u32x2 hp, lp;
        SplitF16::unit_planar(pa[i], sc_a, hp, lp);
        u32x2* d = reinterpret_cast<u32x2*>(ldsA + (g >> 1) * NPA + (at - g * NPA)) + (g & 1);
        d[0] = hp; d[2 * 2 * NPA] = lp;  // the l planes sit two unit planes behind the h planes
      } else if constexpr (SPLIT) ldsA[at] = SplitF16::unit(pa[i], sc_a);
      else ldsA[at] = pa[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int at = ((i * TPI + tsub) * 4 + g) * NPB + bn_;
      if constexpr (P) {
        u32x2 hp, lp;
        if (a.wsplit) SplitF16::planes_of(pb[i], hp, lp);  // (uniform: the weights arrive as (h | l) words, or are split here)
        else SplitF16::unit_planar(pb[i], sc_b, hp, lp);
        u32x2* d = reinterpret_cast<u32x2*>(ldsB + ((i * TPI + tsub) * 4 + (g >> 1)) * NPB + bn_) + (g & 1);
        d[0] = hp; d[2 * 2 * NPB] = lp;
      } else if constexpr (SPLIT) ldsB[at] = a.wsplit ? pb[i] : SplitF16::unit(pb[i], sc_b);  // (uniform: the weights arrive split, or are split here)
      else ldsB[at] = pb[i];
    }
  };
  // one step of the matrix loop: acc[m][n] += A(m) x B(n) for the wave's MT x NT accumulators.  Split mode forms the (L, H) copy of
  // a B fragment right before its MT MFMA pairs (4 v_alignbit per 2 * MT MFMAs; kept out of the double buffer: 16 more live registers
  // per buffered fragment would spill the 4 x 4 shape)
  auto mma_step = [&](auto&& a_of, const u32x4* bfr) {
    if constexpr (SPLIT == 1) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const u32x4 bs = SplitF16::swap_hl(bfr[n]);  // (H, L) as staged, and (L, H)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = SplitF16::mma(a_of(m), bfr[n], bs, acc[m][n]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = Mma<T>::run(a_of(m), bfr[n], acc[m][n]);
    }
  };

  fetch(0);
  if constexpr (SPLIT) {
    const unsigned ma = raw_a2 > raw_a ? raw_a2 : raw_a;
    const int ea = SplitF16::exp_of(ma & 0x7FFFFFFFu), eb = SplitF16::exp_of(raw_w & 0x7FFFFFFFu);
    sc_a = SplitF16::pow2(ea); sc_b = SplitF16::pow2(eb); e_out = -(ea + eb);
  }
  for (int c0 = 0; c0 < ctot; c0 += KB) {
    commit();
    __syncthreads();
    if (c0 + KB < ctot) fetch(c0 + KB);
    if constexpr (P) {
      // per tap: the (h, l) fragments of the wave's MB pixel blocks and (H, L) of its NB channel blocks, 3 * MB * NB MFMAs; fragments
      // double buffered in registers (tap t + 1's reads are issued in front of tap t's MFMAs)
      u32x4 af[2][MB][2], bf[2][NB][2];
      auto load_tap = [&](int tl, int toff, u32x4 (*afr)[2], u32x4 (*bfr)[2]) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) bfr[n][pl] = ldsB[(tl * 4 + pl * 2 + kh) * NPB + n * 32 + r32];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) afr[m][pl] = ldsA[(pl * 2 + kh) * NPA + S * (wave * MT + m + MB * prow) * PITCH + toff + ppx];
      };
      auto mma_tap = [&](const u32x4 (*afr)[2], const u32x4 (*bfr)[2]) {
#pragma unroll
        for (int pr3 = 0; pr3 < 3; ++pr3)  // l H, h L, h H
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              accp[m][n] = SplitF16::mfma32(afr[m][pr3 == 0 ? 1 : 0], bfr[n][pr3 == 1 ? 1 : 0], accp[m][n]);
      };
      if constexpr (MODE == MODE_T3S2) {  // run-time tap count (depends on the output parity)
        for (int ta = 0; ta < nth; ++ta)
          for (int tb = 0; tb < ntw; ++tb) {
            load_tap(ta * ntw + tb, tap_off(ta, tb), af[0], bf[0]);
            mma_tap(af[0], bf[0]);
          }
      } else {
        constexpr int KSW = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : 1;
        constexpr int NTAPS = KSW * KSW;
        load_tap(0, tap_off(0, 0), af[0], bf[0]);
#pragma unroll
        for (int tl = 0; tl < NTAPS; ++tl) {
          const int cur = tl & 1;
          if (tl + 1 < NTAPS) load_tap(tl + 1, tap_off((tl + 1) / KSW, (tl + 1) % KSW), af[cur ^ 1], bf[cur ^ 1]);
          __builtin_amdgcn_sched_barrier(0);
          mma_tap(af[cur], bf[cur]);
        }
      }
    } else if constexpr (MODE == MODE_T3S2) {  // run-time tap count (depends on the output parity)
      for (int ta = 0; ta < nth; ++ta) {
        for (int tb = 0; tb < ntw; ++tb) {
          const int tl = ta * ntw + tb;
          const int toff = tap_off(ta, tb);
          u32x4 bf[NT], af[MT];
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = ldsB[(tl * 4 + q) * NPB + (wc * NT + n) * 16 + pr];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = ldsA[q * NPA + S * (wave * MT + m) * PITCH + toff + pr];
          mma_step([&](int m) -> const u32x4& { return af[m]; }, bf);
        }
      }
    } else if constexpr (MODE == MODE_G3S1) {
      // stride-1 3x3: the A fragment of (row m, tap (ta, tb)) is the 16-pixel run of image row m + ta shifted by tb,
      // i.e. it depends on (m + ta, tb) only -- so per tb the wave reads its MT + 2 rows ONCE and slides the 3 vertical
      // taps over them in registers: 3 * (MT + 2) A reads per chunk instead of 9 * MT (18 vs 36 at MT = 4; the LDS pipe
      // is as busy as the matrix pipe at 72 reads per 144 MFMAs).  B fragments and the next tb's rows are fetched one
      // step ahead, spread over the three steps of a tb so every step issues the same number of ds_read_b128.
      constexpr int ROWS = MT + 2, RPS = (ROWS + 2) / 3;
      u32x4 fr[2][ROWS], bf[2][NT];
      auto load_row = [&](int j, int tb, u32x4* f) { f[j] = ldsA[q * NPA + (wave * MT + j) * PITCH + tb + pr]; };
      auto load_b = [&](int tl, u32x4* b) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = ldsB[(tl * 4 + q) * NPB + (wc * NT + n) * 16 + pr];
      };
#pragma unroll
      for (int j = 0; j < ROWS; ++j) load_row(j, 0, fr[0]);
      load_b(0, bf[0]);
#pragma unroll
      for (int step = 0; step < 9; ++step) {
        const int tb = step / 3, ta = step % 3;
        const int cf = tb & 1, cb = step & 1;
        if (step + 1 < 9) load_b(((step + 1) % 3) * 3 + (step + 1) / 3, bf[cb ^ 1]);
        if (tb < 2) {
#pragma unroll
          for (int j = ta * RPS; j < (ta + 1) * RPS && j < ROWS; ++j) load_row(j, tb + 1, fr[cf ^ 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_step([&](int m) -> const u32x4& { return fr[cf][m + ta]; }, bf[cb]);
      }
    } else {
      // compile-time taps: fragments are double buffered in registers -- tap t+1's ds_read_b128s are issued before
      // tap t's MT*NT MFMAs, so the LDS latency hides behind a full tap of matrix work
      constexpr int KSW = (MODE == MODE_G3S1 || MODE == MODE_G3S2) ? 3 : MODE == MODE_G2S2 ? 2 : 1;
      constexpr int NTAPS = KSW * KSW;
      u32x4 bf[2][NT], af[2][MT];
      auto load_tap = [&](int tl, u32x4* bfr, u32x4* afr) {
        const int toff = tap_off(tl / KSW, tl % KSW);
#pragma unroll
        for (int n = 0; n < NT; ++n) bfr[n] = ldsB[(tl * 4 + q) * NPB + (wc * NT + n) * 16 + pr];
#pragma unroll
        for (int m = 0; m < MT; ++m) afr[m] = ldsA[q * NPA + S * (wave * MT + m) * PITCH + toff + pr];
      };
      load_tap(0, bf[0], af[0]);
#pragma unroll
      for (int tl = 0; tl < NTAPS; ++tl) {
        const int cur = tl & 1;
        if (tl + 1 < NTAPS) load_tap(tl + 1, bf[cur ^ 1], af[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this tap's MFMAs (the scheduler would sink it)
        mma_step([&](int m) -> const u32x4& { return af[cur][m]; }, bf[cur]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue
  const int hd = TMODE ? (a.Hout - ph + 1) / 2 : a.Hout;
  const int wd = TMODE ? (a.Wout - pw + 1) / 2 : a.Wout;
  int rowoff[4];
  float cmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int px = pi16(4 * q + r);
    rowoff[r] = (wave * MT * 16 + px) * OSTR + pr;
    cmask[r] = (ox0 + px < wd) ? 1.f : 0.f;
  }
  const bool full = (oy0 + TH <= hd) && (ox0 + 16 <= wd);
  float s1[NT], s2[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { s1[n] = 0.f; s2[n] = 0.f; }
  if constexpr (SPLIT == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = SplitF16::unscale(acc[m][n][r], e_out);
  }
  unsigned am1 = 0u, am2 = 0u;
  if constexpr (P) {
    // a lane's 16 values of block (mb, nb): channel n0 + nb * 32 + r32, image rows mb + MB * (j >> 3) of the wave's MT, pixels
    // px8[j & 7] = pi16 of the MFMA row (j & 3) + 8 * ((j >> 2) & 1) + 4 * kh
    float cm8[8];
    int px8[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      px8[jj] = pi16((jj & 3) + 8 * (jj >> 2) + 4 * kh);
      cm8[jj] = (ox0 + px8[jj] < wd) ? 1.f : 0.f;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) accp[m][n][j] = SplitF16::unscale(accp[m][n][j], e_out) + bv[2 * n];
    if (want_amax) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int col = n0 + n * 32 + r32;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const bool rok = oy0 + wave * MT + m + MB * (j >> 3) < hd;
            const float v = accp[m][n][j];  // (a copy: __builtin_bit_cast of the vector ELEMENT read element 0 for every j)
            const unsigned b = (rok && cm8[j & 7] != 0.f && col < nout) ? (__builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu) : 0u;
            if (col < a.o1) am1 = b > am1 ? b : am1; else am2 = b > am2 ? b : am2;
          }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float v = accp[m][n][j];
          if (a.stats != nullptr) {
            const float rm = (oy0 + wave * MT + m + MB * (j >> 3) < hd) ? 1.f : 0.f;
            const float vm = full ? v : v * (rm * cm8[j & 7]);
            s1[n] += vm; s2[n] += vm * v;
          }
          ldsO[((wave * MT + m + MB * (j >> 3)) * 16 + px8[j & 7]) * OSTR + n * 32 + r32] = v;
        }
    if (a.stats != nullptr) {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float t1 = s1[n], t2 = s2[n];
        t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        if (kh == 0) { ldsR[(0 * 4 + wave) * BN + n * 32 + r32] = t1; ldsR[(1 * 4 + wave) * BN + n * 32 + r32] = t2; }
      }
    }
  }
  if constexpr (sizeof(T) == 4 && !P) {
    if (want_amax) {  // ONE uniform branch around the whole pass (inside the store loop below it became a branch per element).
      // Stored elements only: rows / pixels beyond the image and channels beyond nout do not count.
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bool rok = oy0 + wave * MT + m < hd;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int col = n0 + (wc * NT + n) * 16 + pr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned b = (rok && cmask[r] != 0.f && col < nout) ? (__builtin_bit_cast(unsigned, acc[m][n][r] + bv[n]) & 0x7FFFFFFFu) : 0u;
            if (col < a.o1) am1 = b > am1 ? b : am1; else am2 = b > am2 ? b : am2;
          }
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < (P ? 0 : MT); ++m) {
    const float rm = (oy0 + wave * MT + m < hd) ? 1.f : 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[m][n][r] + bv[n];
        if (a.stats != nullptr) {
          const float vm = full ? v : v * (rm * cmask[r]);
          s1[n] += vm; s2[n] += vm * v;
        }
        ldsO[rowoff[r] + m * 16 * OSTR + (wc * NT + n) * 16] = Elem<T>::cvt(v);
      }
    }
  }
  if (!P && a.stats != nullptr) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float t1 = s1[n], t2 = s2[n];
      t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) { ldsR[(0 * 4 + wave) * BN + (wc * NT + n) * 16 + pr] = t1; ldsR[(1 * 4 + wave) * BN + (wc * NT + n) * 16 + pr] = t2; }
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (want_amax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned t1 = (unsigned)__shfl_xor((int)am1, o, 64), t2 = (unsigned)__shfl_xor((int)am2, o, 64);
        am1 = t1 > am1 ? t1 : am1; am2 = t2 > am2 ? t2 : am2;
      }
      if (lane == 0) { if (am1) atomicMax(&ldsM[0], am1); if (am2) atomicMax(&ldsM[1], am2); }
    }
  }
  __syncthreads();
  if constexpr (sizeof(T) == 4) {
    if (want_amax && tid < 2) {
      unsigned* dst = tid == 0 ? a.amax_out1 : a.amax_out2;
      const unsigned b = ldsM[tid];
      if (dst != nullptr && b > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, b);
    }
  }
  typedef __attribute__((address_space(1))) float gfloat;  // global (not flat) stores: a flat access makes the compiler drain every outstanding memory operation around it
  if constexpr (WC == 1) {
    if (a.stats != nullptr && tid < BN && n0 + tid < nout) {
      const float t1 = ldsR[0 * BN + tid] + ldsR[1 * BN + tid] + ldsR[2 * BN + tid] + ldsR[3 * BN + tid];
      const float t2 = ldsR[4 * BN + tid] + ldsR[5 * BN + tid] + ldsR[6 * BN + tid] + ldsR[7 * BN + tid];
      const size_t tile = (size_t)img * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx;
      gfloat* dst = (gfloat*)(a.stats + (tile * nout + n0 + tid) * 2);
      dst[0] = t1; dst[1] = t2;
    }
  } else {
    // the statistics keep the host's 8-row tile layout (mia_conv_mma_tiles): row groups {0, 1} and {2, 3} of the 16-row
    // tile are two statistics tiles
    static_assert(WC == 1 || (TH == 16 && 2 * BN <= NTHR), "two 8-row statistics tiles per 16-row tile");
    const int half = tid / BN, ch = tid - half * BN;
    if (a.stats != nullptr && tid < 2 * BN && n0 + ch < nout && 2 * ty + half < a.st_tiles_y) {
      const float t1 = ldsR[(0 + 2 * half) * BN + ch] + ldsR[(1 + 2 * half) * BN + ch];
      const float t2 = ldsR[(4 + 2 * half) * BN + ch] + ldsR[(5 + 2 * half) * BN + ch];
      const size_t tile = (size_t)img * (a.tiles_x * a.st_tiles_y) + (2 * ty + half) * a.tiles_x + tx;
      gfloat* dst = (gfloat*)(a.stats + (tile * nout + n0 + ch) * 2);
      dst[0] = t1; dst[1] = t2;
    }
  }
  // coalesced 16-byte stores.  A block's channel range normally lies in one destination; when the split point o1 is not
  // a multiple of BN the straddling block issues every store twice, once per destination, each masked to its own lanes.
  constexpr int UPP = BN / EPU, PPI = NTHR / UPP, O_IT = TH * 16 / PPI;
  const size_t opix = (size_t)a.Hout * a.Wout;
  const int cu = tid % UPP, pl0 = tid / UPP;
  const int y0 = pl0 >> 4, px = pl0 & 15;
  const int ox = TMODE ? 2 * (ox0 + px) + pw : ox0 + px;
  const int ch = n0 + cu * EPU;
  const bool pixok = (ox0 + px < wd) && (ch < nout);
  auto store_to = [&](bool second) {
    const int cn = second ? a.o2 : a.o1;
    T* obase = second ? static_cast<T*>(a.out2) : static_cast<T*>(a.out1);
    const rsrc_t rso = make_rsrc(obase + (size_t)img * opix * cn, (unsigned)(opix * cn * ES));
    const int nloc = second ? ch - a.o1 : ch;
    const bool colok = pixok && ((ch >= a.o1) == second);
    unsigned voffs[O_IT];
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
      const int y = y0 + i * (PPI / 16);
      const int oy = TMODE ? 2 * (oy0 + y) + ph : oy0 + y;
      voffs[i] = (colok && oy0 + y < hd) ? (unsigned)((((size_t)oy * a.Wout + ox) * cn + nloc) * ES) : SENT;
    }
    if (a.acc_out) {  // out += result: every previous value is loaded before the first store (the compiler cannot reorder them itself)
      u32x4 prev[O_IT];
#pragma unroll
      for (int i = 0; i < O_IT; ++i) prev[i] = __builtin_amdgcn_raw_buffer_load_b128(rso, (int)voffs[i], 0, 0);
#pragma unroll
      for (int i = 0; i < O_IT; ++i) {
        alignas(16) T dv[EPU]; alignas(16) T pv[EPU];
        *reinterpret_cast<u32x4*>(dv) = *reinterpret_cast<const u32x4*>(ldsO + (pl0 + i * PPI) * OSTR + cu * EPU);
        *reinterpret_cast<u32x4*>(pv) = prev[i];
#pragma unroll
        for (int e = 0; e < EPU; ++e) dv[e] = Elem<T>::cvt(Elem<T>::ld(pv + e) + Elem<T>::ld(dv + e));
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(dv), rso, (int)voffs[i], 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
      const u32x4 d = *reinterpret_cast<const u32x4*>(ldsO + (pl0 + i * PPI) * OSTR + cu * EPU);
      __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)voffs[i], 0, 0);
    }
  };
  const bool first_part = n0 < a.o1, second_part = a.o2 > 0 && n0 + BN > a.o1;  // uniform
  if (first_part) store_to(false);
  if (second_part) store_to(true);
}

template <typename T, int MODE, int MT, int NT, int WC = 1, int SPLIT = 0>
static void flaunch(const ConvArgs& a, int grid_y, hipStream_t st) {
  int grid_x = a.N * a.tiles_x * a.tiles_y * a.nblk_n;
  if (a.xcd) {  // groups of (channel blocks x parity classes) per tile, tiles rounded up to a multiple of 8
    grid_x = ((a.N * a.tiles_x * a.tiles_y + 7) / 8) * 8 * a.nblk_n * grid_y;
    grid_y = 1;
  }
  hipLaunchKernelGGL((conv_mma_fast_kernel<T, MODE, MT, NT, WC, SPLIT>), dim3(grid_x, grid_y), dim3(256 * WC), 0, st, a);
}
template <typename T, int MODE, int MT>
static void flaunch_nt(const ConvArgs& a, int nt, int grid_y, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    if (a.split) {  // fp32 tensors, two-part split f16 products: three on planes (32 x 32 tiles) or four on interleaved words
      if (a.split == 2 && nt == 4) flaunch<T, MODE, MT, 4, 1, 2>(a, grid_y, st);
      else if (a.split == 2 && nt == 2) flaunch<T, MODE, MT, 2, 1, 2>(a, grid_y, st);
      else if (nt == 4) flaunch<T, MODE, MT, 4, 1, 1>(a, grid_y, st);
      else if (nt == 2) flaunch<T, MODE, MT, 2, 1, 1>(a, grid_y, st);
      else flaunch<T, MODE, MT, 1, 1, 1>(a, grid_y, st);
      return;
    }
  }
  if (nt == 4) flaunch<T, MODE, MT, 4>(a, grid_y, st);
  else if (nt == 2) flaunch<T, MODE, MT, 2>(a, grid_y, st);
  else flaunch<T, MODE, MT, 1>(a, grid_y, st);
}
template <typename T, int MODE>
static void flaunch_mt(const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st) {
  if constexpr (MODE == MODE_G3S2 || MODE == MODE_G2S2) {
    if constexpr (MODE == MODE_G3S2 && sizeof(T) == 2) {
      if (mt == 4) { flaunch<T, MODE, 4, 4, 2>(a, grid_y, st); return; }  // 512-thread 16-row tile, 128-channel blocks
    }
    flaunch_nt<T, MODE, 2>(a, nt, grid_y, st);
  } else {
    if (mt >= 4) flaunch_nt<T, MODE, 4>(a, nt, grid_y, st);
    else flaunch_nt<T, MODE, 2>(a, nt, grid_y, st);
  }
}
template <typename T>
static int fdispatch(int mode, const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st) {
  switch (mode) {
    case MODE_G3S1: flaunch_mt<T, MODE_G3S1>(a, mt, nt, grid_y, st); break;
    case MODE_G3S2: flaunch_mt<T, MODE_G3S2>(a, mt, nt, grid_y, st); break;
    case MODE_G2S2: flaunch_mt<T, MODE_G2S2>(a, mt, nt, grid_y, st); break;
    case MODE_T3S2: flaunch_mt<T, MODE_T3S2>(a, mt, nt, grid_y, st); break;
    case MODE_T2S2: flaunch_mt<T, MODE_T2S2>(a, mt, nt, grid_y, st); break;
    case MODE_G1: flaunch_mt<T, MODE_G1>(a, mt, nt, grid_y, st); break;
    default: return MIA_EARG;
  }
  return MIA_OK;
}

bool conv_mma_fast_eligible(int dtype, const ConvArgs& a, int nt) {
  const int epu = dtype == MIA_BF16 ? 8 : 4, kb = 4 * epu, es = dtype == MIA_BF16 ? 2 : 4, bn = 16 * nt;
  if (!a.vec_in || !a.vec_out) return false;
  if (a.c1 % kb != 0 || a.c2 % kb != 0) return false;
  const size_t lim = (size_t)1 << 31;
  const size_t ipix = (size_t)a.Hin * a.Win, opix = (size_t)a.Hout * a.Wout;
  if (ipix * (a.c1 > a.c2 ? a.c1 : a.c2) * es >= lim) return false;
  if (opix * (a.o1 > a.o2 ? a.o1 : a.o2) * es >= lim) return false;
  if ((size_t)9 * a.npad * a.kpad * es >= lim) return false;
  return true;
}

int conv_mma_fast_launch(int mode, int dtype, const ConvArgs& a, int mt, int nt, int grid_y, hipStream_t st) {
  return dtype == MIA_BF16 ? fdispatch<bf16_t>(mode, a, mt, nt, grid_y, st) : fdispatch<float>(mode, a, mt, nt, grid_y, st);
}
