// Probe: does v_mfma_f32_16x16x32_f16 keep fp16 DENORMAL inputs (needed by the fp16 two-part split of fp32 operands: the low part of a
// small element is an fp16 denormal)?  Also v_cvt_pk_f16_f32 (RNE, denormal results).  Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f16_denorm mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ void k(float av, float bv, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float avs[] = {1.0f, 6.103515625e-05f /*2^-14 min normal*/, 3.0517578125e-05f /*2^-15*/, 9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24 min denormal*/};
  for (float av : avs) {
    k<<<1, 64>>>(av, 1024.f, d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.6e (as f16 -> %.6e)  b = 1024: mfma sum of 32 products = %.6e  expected %.6e  %s\n", av, h[1], h[0], 32.0 * av * 1024.0,
           h[0] == (float)(32.0 * av * 1024.0) ? "KEPT" : "FLUSHED/DIFFERENT");
  }
  return 0;
}
