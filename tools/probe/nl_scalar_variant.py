"""A/B build: the normalise-on-load transform of conv64_persist_kernel<NL> / wgrad_bf16_2wg_kernel<NL> with SCALAR fp32 VALU ops
(v_fma_f32 / v_mul_f32 via inline asm, so -O3 cannot SLP-pack them) instead of v_pk_fma_f32 / v_pk_mul_f32 -- MI355X_MICROARCH.md prices
packed f32 VALU beside MFMAs as an anti-lever.  Same arithmetic, bit-identical results.  Builds tools/ab/libmia_nlscalar.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "medical-image-analysis_amd", "csrc")
OBJ = os.path.join(ROOT, "medical-image-analysis_amd", "mia_hip", "_obj")
AB = os.path.join(ROOT, "tools", "ab")
HELP = '''
__device__ __forceinline__ unsigned nl_pair_scalar(unsigned w, float sc0, float sc1, float sh0, float sh1, float sl) {
  const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
  float v0, v1, m0, m1;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(sc0), "v"(x0), "v"(sh0));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v1) : "v"(sc1), "v"(x1), "v"(sh1));
  asm("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(v0), "v"(sl));
  asm("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(v1), "v"(sl));
  typedef float f2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{__builtin_fmaxf(v0, m0), __builtin_fmaxf(v1, m1)}, b2_));
}
'''
c64 = open(os.path.join(CSRC, "conv64.hip")).read()
old = c64[c64.index("      const f32x2_t x = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};"):
          c64.index("      o[d] = MASK ? (r & keep) : r;")]
c64 = c64.replace(old, "      const unsigned r = nl_pair_scalar(w, sc[0], sc[1], sh[0], sh[1], a.nl_slope);\n")
c64 = c64.replace("template <bool NL, bool CR>\n__global__", HELP + "template <bool NL, bool CR>\n__global__", 1)
wg = open(os.path.join(CSRC, "conv_wgrad.hip")).read()
old = wg[wg.index("            const nl_f32x2 x = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};"):
         wg.index("          }\n        }\n        __builtin_amdgcn_sched_barrier(0);\n      }\n      if (!interior) {  // border tile: halo units outside the image go back to zero")]
wg = wg.replace(old, "            px[i][2 * hf + d] = nl_pair_scalar(w, sc[2 * d], sc[2 * d + 1], sh[2 * d], sh[2 * d + 1], a.nl_slope);\n")
wg = wg.replace("template <int TH, bool NL = false>", HELP + "template <int TH, bool NL = false>", 1)
os.makedirs(AB, exist_ok=True)
objs = []
for name, text in (("conv64", c64), ("conv_wgrad", wg)):
    path = os.path.join(CSRC, f"_nls_{name}.hip")
    open(path, "w").write(text)
    try:
        o = os.path.join(AB, f"nls_{name}.o")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage",
                            "-c", path, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        nm = None
        for line in r.stderr.splitlines():
            if "Function Name:" in line:
                nm = line.split("Function Name:")[1].split()[0]
            if nm and ("conv64_persist_kernelILb1ELb0" in nm or "wgrad_bf16_2wg_kernelILi8ELb1" in nm) and ("VGPRs:" in line or "ScratchSize" in line):
                print(nm, line.split("remark:")[1].strip())
        objs.append(o)
    finally:
        os.remove(path)
rest = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f not in ("conv64.o", "conv_wgrad.o")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(AB, "libmia_nlscalar.so")] + objs + rest)
print("built tools/ab/libmia_nlscalar.so")
