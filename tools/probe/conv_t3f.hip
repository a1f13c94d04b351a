// PROBE (round 5, measured and not shipped -- profiles/r05_ab_conv_t3_fused.txt): built into the library as option conv_t3_fused for the
// A/B (csrc/ copy + conv_common.h declarations + `if (opt.conv_t3_fused && fast && conv_t3_fused_eligible(..)) rc = conv_t3_fused_launch(a, st);`
// in front of conv_mma_run's dispatch chain), bit-identical to the tile kernel, +-0 over the step's four launches.
// Input gradient of the 3x3 / stride-2 conv (MODE_T3S2, bf16) with the four output-parity classes FUSED in one workgroup.
//
// dx[2y + ph][2x + pw] = sum over the taps (kh, kw) with kh == 1 (ph = 0) or kh in {0, 2} (ph = 1), likewise kw / pw, of
// dy[y + (kh == 0)][x + (kw == 0)] . W[kh][kw]: 1 / 2 / 2 / 4 taps per class.  The tile kernel (conv_mma_fast.hip) runs one workgroup per
// class: each stages its own copy of the 17 x 17 dy tile and pays two barriers per 32-channel chunk for 16 / 32 / 32 / 64 MFMAs per wave
// -- the launch furthest below its roof in the step (0.26 of the matrix peak; 4 x 0.47 ms per cfg3 step, 5 x 1.19 ms per cfg5 step).
// Here a workgroup owns an 8 x 16 tile of CLASS pixels (= 16 x 32 output pixels) for all four classes: the 9 x 17 dy tile is staged once
// per chunk, all nine taps of the weights are staged, the six A fragments a wave needs ((2 + 1) rows x 2 column shifts) are read once and
// meet nine B-fragment sets: 72 MFMAs per wave and barrier pair, every wave the same.  Four accumulator sets of 2 x NT tiles
// (128 registers at NT = 4), two workgroups per CU.  The epilogue writes the classes one after the other through the same LDS image
// (16-byte stores of 8 channels at output pixel (2y + ph, 2x + pw)); accumulate mode (out += result) and two destinations as in the
// tile kernel.  Same products in the same order per output element as the tile kernel (chunks ascending, taps ascending within a
// class): bit-identical results.
// Contract (checked on the host, else the tile kernel runs): bf16, one source, c1 % 32 == 0, o1 % 8 == 0, o2 % 8 == 0, 16-byte aligned
// pointers, per-image tensors < 2 GiB, no statistics, >= 32 output channels.
#include "conv_common.h"

typedef __amdgpu_buffer_rsrc_t t3rsrc_t;
#define T3SENT 0xFFFFFFF0u /* always beyond num_records */

__device__ __forceinline__ t3rsrc_t t3_make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_t3_fused_kernel(const ConvArgs a) {
  constexpr int MT = 2, TH = 4 * MT, BN = 16 * NT, EPU = 8, KB = 32, ES = 2;
  constexpr int IH = TH + 1, IW = 17, PITCH = IW, NPIX = IH * PITCH;
  constexpr int PL = 64, A_IT = (NPIX + PL - 1) / PL;
  constexpr int NPA = ((A_IT * PL + 13) / 16) * 16 + 2, NPB = BN + 2;
  constexpr int TPI = PL / BN, B_IT = (9 + TPI - 1) / TPI;
  constexpr int OSTR = BN + EPU;
  constexpr int A_UNITS = 4 * NPA, B_UNITS = B_IT * TPI * 4 * NPB;
  constexpr int STAGE_BYTES = (A_UNITS + B_UNITS) * 16, OUT_BYTES = TH * 16 * OSTR * ES;
  constexpr int LDS_BYTES = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  u32x4* ldsA = reinterpret_cast<u32x4*>(smem);
  u32x4* ldsB = ldsA + A_UNITS;
  bf16_t* ldsO = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r16 = lane & 15, pr = pi16(r16);
  const int g = tid & 3, p4 = tid >> 2;

  // block order: the output-channel blocks of a tile take consecutive slots of ONE XCD (conv_mma_fast.hip)
  int bid = blockIdx.x, nb;
  if (a.xcd) {
    const int slot = bid >> 3, grp = slot / a.nblk_n;
    nb = slot - grp * a.nblk_n;
    bid = grp * 8 + (bid & 7);
    if (bid >= a.N * a.tiles_x * a.tiles_y) return;
  } else {
    nb = bid % a.nblk_n; bid /= a.nblk_n;
  }
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid;
  const int n0 = nb * BN, oy0 = ty * TH, ox0 = tx * 16;

  const bf16_t* in1 = static_cast<const bf16_t*>(a.in1);
  const size_t ipix = (size_t)a.Hin * a.Win;
  const t3rsrc_t rs1 = t3_make_rsrc(in1 + (size_t)img * ipix * a.c1, (unsigned)(ipix * a.c1 * ES));
  const t3rsrc_t rsw = t3_make_rsrc(a.wp, (unsigned)((size_t)9 * a.npad * a.kpad * ES));
  const unsigned cs_es = (unsigned)(a.c1 * ES);

  unsigned a_voff[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int pix = p4 + PL * i;
    const int iy = pix / IW, ix = pix - iy * IW;
    const int gy = oy0 + iy, gx = ox0 + ix;
    const bool ok = pix < NPIX && gy < a.Hin && gx < a.Win;
    a_voff[i] = ok ? (unsigned)(gy * a.Win + gx) * cs_es + (unsigned)(g * 16) : T3SENT;
  }
  const int bn_ = p4 % BN, tsub = p4 / BN;
  const int wtap_bytes = a.npad * a.kpad * ES;
  const unsigned b_voff0 = (unsigned)((((size_t)n0 + bn_) * a.kpad + g * EPU) * ES);

  f32x4 acc[4][MT][NT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[c][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nout = a.o1 + a.o2;

  u32x4 pa[A_IT], pb[B_IT];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)a_voff[i], c0 * ES, 0);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int tl = i * TPI + tsub;
      pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(tl < 9 ? b_voff0 : T3SENT), c0 * ES + tl * wtap_bytes, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ldsA[g * NPA + p4 + PL * i] = pa[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) ldsB[((i * TPI + tsub) * 4 + g) * NPB + bn_] = pb[i];
  };

  fetch(0);
  for (int c0 = 0; c0 < a.c1; c0 += KB) {
    commit();
    __syncthreads();
    if (c0 + KB < a.c1) fetch(c0 + KB);
    // the wave's rows r = 0 .. 2 (its two class rows + the one below) at column shifts 0 / 1: read once, used by every tap.  The
    // 9 * NT B fragments stream through a ring of six registers sets, fetched four fragments (eight MFMAs) ahead of their use (a full
    // double buffer of NT fragments does not fit beside the 128 accumulator registers)
    constexpr int NF = 9 * NT, RING = 6, AHEAD = 4;
    u32x4 af[MT + 1][2], bf[RING];
    auto load_b = [&](int f) { bf[f % RING] = ldsB[((f / NT) * 4 + q) * NPB + (f % NT) * 16 + pr]; };
#pragma unroll
    for (int f = 0; f < AHEAD; ++f) load_b(f);
#pragma unroll
    for (int r = 0; r < MT + 1; ++r)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) af[r][dw] = ldsA[q * NPA + (wave * MT + r) * PITCH + dw + pr];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int t = f / NT, n = f % NT, kh = t / 3, kw = t % 3;
      const int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0), dh = kh == 0 ? 1 : 0, dw = kw == 0 ? 1 : 0;
      if (f + AHEAD < NF) load_b(f + AHEAD);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of this fragment's MFMAs
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[cls][m][n] = Mma<bf16_t>::run(af[m + dh][dw], bf[f % RING], acc[cls][m][n]);
    }
    __syncthreads();
  }

  // ---- epilogue: one class after the other through the same LDS image
  float bv[NT];  // (loaded here, not up front: the main loop has no register to spare, and an input gradient has no bias)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int bi = n0 + n * 16 + pr;
    bi = bi < nout ? bi : nout - 1;
    bv[n] = a.bias ? a.bias[bi] : 0.f;
  }
  constexpr int UPP = BN / EPU, PPI = 256 / UPP, O_IT = TH * 16 / PPI;
  const size_t opix = (size_t)a.Hout * a.Wout;
  const int cu = tid % UPP, pl0 = tid / UPP;
  const int y0 = pl0 >> 4, px = pl0 & 15;
  const int ch = n0 + cu * EPU;
  const bool first_part = n0 < a.o1, second_part = a.o2 > 0 && n0 + BN > a.o1;  // uniform
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    const int hd = (a.Hout - ph + 1) / 2, wd = (a.Wout - pw + 1) / 2;
    if (cls) __syncthreads();  // the previous class's image has been read
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ldsO[((wave * MT + m) * 16 + pi16(4 * q + r)) * OSTR + n * 16 + pr] = f2bf(acc[cls][m][n][r] + bv[n]);
    __syncthreads();
    const int ox = 2 * (ox0 + px) + pw;
    const bool pixok = (ox0 + px < wd) && (ch < nout);
    auto store_to = [&](bool second) {
      const int cn = second ? a.o2 : a.o1;
      bf16_t* obase = second ? static_cast<bf16_t*>(a.out2) : static_cast<bf16_t*>(a.out1);
      const t3rsrc_t rso = t3_make_rsrc(obase + (size_t)img * opix * cn, (unsigned)(opix * cn * ES));
      const int nloc = second ? ch - a.o1 : ch;
      const bool colok = pixok && ((ch >= a.o1) == second);
      unsigned voffs[O_IT];
#pragma unroll
      for (int i = 0; i < O_IT; ++i) {
        const int y = y0 + i * (PPI / 16);
        const int oy = 2 * (oy0 + y) + ph;
        voffs[i] = (colok && oy0 + y < hd) ? (unsigned)((((size_t)oy * a.Wout + ox) * cn + nloc) * ES) : T3SENT;
      }
      if (a.acc_out) {  // out += result: every previous value is loaded before the first store
        u32x4 prev[O_IT];
#pragma unroll
        for (int i = 0; i < O_IT; ++i) prev[i] = __builtin_amdgcn_raw_buffer_load_b128(rso, (int)voffs[i], 0, 0);
#pragma unroll
        for (int i = 0; i < O_IT; ++i) {
          alignas(16) bf16_t dv[EPU]; alignas(16) bf16_t pv[EPU];
          *reinterpret_cast<u32x4*>(dv) = *reinterpret_cast<const u32x4*>(ldsO + (pl0 + i * PPI) * OSTR + cu * EPU);
          *reinterpret_cast<u32x4*>(pv) = prev[i];
#pragma unroll
          for (int e = 0; e < EPU; ++e) dv[e] = f2bf(bf2f(pv[e]) + bf2f(dv[e]));
          store_data_fence();
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(dv), rso, (int)voffs[i], 0, 0);
          store_data_pad();
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < O_IT; ++i) {
        const u32x4 d = *reinterpret_cast<const u32x4*>(ldsO + (pl0 + i * PPI) * OSTR + cu * EPU);
        __builtin_amdgcn_raw_buffer_store_b128(d, rso, (int)voffs[i], 0, 0);
      }
    };
    if (first_part) store_to(false);
    if (second_part) store_to(true);
  }
}

bool conv_t3_fused_eligible(int mode, int dtype, const ConvArgs& a) {
  if (mode != MODE_T3S2 || dtype != MIA_BF16) return false;
  if (!a.vec_in || !a.vec_out || a.c2 != 0 || a.c1 % 32 != 0 || a.stats != nullptr || a.nl_scale != nullptr || a.cr_y != nullptr) return false;
  if (a.o1 + a.o2 <= 16) return false;
  const size_t lim = (size_t)1 << 31;
  if ((size_t)a.Hin * a.Win * a.c1 * 2 >= lim || (size_t)a.Hout * a.Wout * (a.o1 > a.o2 ? a.o1 : a.o2) * 2 >= lim) return false;
  if ((size_t)9 * a.npad * a.kpad * 2 >= lim) return false;
  return true;
}

// (sets its own tile grid: 8 class rows x 16 class pixels per workgroup)
int conv_t3_fused_launch(ConvArgs a, hipStream_t st) {
  const int hd = (a.Hout + 1) / 2, wd = (a.Wout + 1) / 2;
  a.tiles_y = (hd + 7) / 8; a.tiles_x = (wd + 15) / 16;
  const int nout = a.o1 + a.o2, nt = nout > 32 ? 4 : 2;
  a.nblk_n = (nout + 16 * nt - 1) / (16 * nt);
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  const int grid = a.xcd ? ((ntiles + 7) / 8) * 8 * a.nblk_n : ntiles * a.nblk_n;
  if (nt == 4) hipLaunchKernelGGL(conv_t3_fused_kernel<4>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_t3_fused_kernel<2>, dim3(grid), dim3(256), 0, st, a);
  return MIA_OK;
}
