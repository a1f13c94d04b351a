// Probe (gfx950): buffer_load_dwordx4 ... offen lds -- destination beyond 64 KB through M0, zero fill of out-of-range lanes,
// completion through vmcnt.  Build: hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned nbytes, unsigned* out, unsigned lds_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // poison
  for (unsigned i = tid; i < (lds_off + 4096) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  const i32x4 r = make_rsrc(src, nbytes);
  // lane L reads source unit (63 - L) [reversed: shows the source address is per lane]; lanes >= 48 out of range
  const unsigned voff = lane < 48 ? (unsigned)((wave * 64 + 63 - lane) * 16) : 0xFFFFFFF0u;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + lds_off + wave * 1024);
  dma16(r, voff, base);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < 1024; i += 256) out[i] = reinterpret_cast<unsigned*>(smem + lds_off)[i];
}

int main() {
  const unsigned n = 4096;  // bytes
  std::vector<unsigned> h(n / 4);
  for (unsigned i = 0; i < n / 4; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, n); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  for (unsigned lds_off : {0u, 32768u, 70000u / 16 * 16, 131072u}) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipMemset(o, 0, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), lds_off + 4096, 0, d, n, o, lds_off);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
          const unsigned got = r[w * 256 + l * 4 + k];
          const unsigned want = l < 48 ? (unsigned)((w * 64 + 63 - l) * 4 + k) : 0u;
          if (got != want) { if (bad < 4) printf("  off %u wave %d lane %d k %d: got %08x want %08x\n", lds_off, w, l, k, got, want); ++bad; }
        }
    printf("lds_off %6u: err %d, mismatches %d\n", lds_off, (int)e, bad);
  }
  return 0;
}
