// Sustained matrix-core rate of gfx950 under the board's power limit, as a function of MFMA shape and of the LDS operand
// traffic beside it (tools/probe/mfma_power.py builds and runs this).  Not part of libmia_hip.
//   SHAPE 0: v_mfma_f32_16x16x32_bf16 (16 Kflop / 16 cycles)    SHAPE 1: v_mfma_f32_32x32x16_bf16 (32 Kflop / 32 cycles)
//   LDSR   : ds_read_b128 per GROUP of 8 MFMAs (0, 2, 4, 8, 16); the values read replace the A operands of the group
//   HBM    : one 16-byte global load per lane every HBM-th group (0 = none), streaming over a big buffer: 2048 * HBM / 16 flop per byte
//            (the canonical 64 -> 64 block of the benchmark: 288 flop per byte, 0.75 LDS reads per MFMA)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE, int LDSR, int HBM, int VALU = 0>
__global__ __launch_bounds__(256, 2) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ out, int iters, size_t src_units,
                                                    unsigned long long* __restrict__ clocks) {
  __shared__ u32x4 lds[4096];  // 64 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) % src_units];
  __syncthreads();
  u32x4 a[8], b[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = lds[(tid * 8 + i) & 4095];
  b[0] = lds[(tid + 17) & 4095]; b[1] = lds[(tid + 99) & 4095];
  constexpr int NACC = SHAPE == 0 ? 8 : 4;
  f32x4 c4[8];
  f32x16 c16[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
  unsigned off = tid;
  size_t gp = ((size_t)blockIdx.x * 256 + tid) % src_units;
  const size_t gstride = (size_t)gridDim.x * 256;
  u32x4 gacc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  typedef float vf2 __attribute__((ext_vector_type(2)));
  vf2 vacc[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
  const vf2 vk = {0.999f, 1.001f};
  u32x4 n[8], ring[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { n[i] = a[i]; ring[i] = a[i]; }
  auto body = [&](u32x4* cur, u32x4* nxt, int u) {  // reads for the NEXT group are in flight while this group's MFMAs run
    if (HBM > 0 && u % (HBM > 0 ? HBM : 1) == 0) {  // eight loads in flight per thread: the oldest is consumed, a new one issued
      gacc[0] ^= ring[0][0]; gacc[1] ^= ring[0][1]; gacc[2] ^= ring[0][2]; gacc[3] ^= ring[0][3];
#pragma unroll
      for (int i = 0; i < 7; ++i) ring[i] = ring[i + 1];
      ring[7] = __builtin_nontemporal_load(src + gp);
      gp += gstride; if (gp >= src_units) gp -= src_units;
    }
#pragma unroll
    for (int r = 0; r < LDSR; ++r) nxt[r & 7] = lds[(off + 64 * r) & 4095];
    off += 37;
    if constexpr (VALU > 0) {  // VALU packed-fp32 FMAs per group on their own registers (the normalise-on-load transform: ~2 per MFMA)
#pragma unroll
      for (int i = 0; i < VALU; ++i) vacc[i & 3] = __builtin_elementwise_fma(vacc[i & 3], vk, vacc[(i + 1) & 3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHAPE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[i]), __builtin_bit_cast(bf16x8, b[i & 1]), c4[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[2 * i]), __builtin_bit_cast(bf16x8, b[i & 1]), c16[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { body(a, n, 2 * u); body(n, a, 2 * u + 1); }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = (float)(gacc[0] ^ gacc[1] ^ gacc[2] ^ gacc[3]) + vacc[0][0] + vacc[1][1] + vacc[2][0] + vacc[3][1];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c4[i][0] + c4[i][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][15];
  if (s == 123.456f) out[tid] = s;
  if (tid == 0 && blockIdx.x < 64) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int LDSR, int HBM, int VALU = 0>
static int run(const void* src, void* out, int iters, size_t units, void* clocks, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((mfma_loop<SHAPE, LDSR, HBM, VALU>), dim3(blocks), dim3(256), 0, st, (const u32x4*)src, (float*)out, iters, units, (unsigned long long*)clocks);
  return (int)hipGetLastError();
}

extern "C" int mfma_power_run(int shape, int ldsr, int hbm, int valu, const void* src, void* out, int iters, size_t units, void* clocks, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define CASE(S, L, H) if (shape == S && ldsr == L && hbm == H) return run<S, L, H>(src, out, iters, units, clocks, blocks, st);
#define CASES(S, H) CASE(S, 0, H) CASE(S, 2, H) CASE(S, 4, H) CASE(S, 6, H) CASE(S, 8, H)
  if (shape == 0 && ldsr == 4 && hbm == 2 && valu == 8) return run<0, 4, 2, 8>(src, out, iters, units, clocks, blocks, st);
  if (shape == 0 && ldsr == 4 && hbm == 2 && valu == 16) return run<0, 4, 2, 16>(src, out, iters, units, clocks, blocks, st);
  if (shape == 0 && ldsr == 4 && hbm == 2 && valu == 24) return run<0, 4, 2, 24>(src, out, iters, units, clocks, blocks, st);
  if (valu != 0) return -2;
  CASES(0, 0) CASES(1, 0) CASES(0, 1) CASES(1, 1) CASES(0, 2) CASES(0, 4) CASES(0, 8)
  return -1;
}
