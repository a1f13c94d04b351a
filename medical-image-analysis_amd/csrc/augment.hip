// Batched augmentation / resize / normalisation kernels (gfx950), per-sample parameters.
//
// Reference: src/transforms/joint_transform.py (RandomAffine :158-206, RandomRotation :100-127,
// JointResize :11-38, RandomRotation90 :40-65, MirrorTransform :67-97), image_transform.py (RandomGamma
// :15-44, RandomContrast/"RandomBrightness" :47-106, RandomGaussianNoise :109-142, RandomGaussianBlur
// :145-193, SimulateLowRes :196-236), normalization.py:9-26.  The reference runs these per sample on
// the CPU inside DataLoader workers through torchvision; here every stage is one (or two, where a
// per-sample statistic is needed) HBM-streaming launch over the whole batch: images [B,C,H,W] fp32,
// labels [B,H,W] int64, parameters in small per-sample device arrays.  `apply[b]`: > 0 = transform sample b, 0 = pass it
// through (copy in -> out), < 0 = SKIP it: nothing of sample b is read or written (round 4: the batched pipeline runs the
// element-wise stages in place and the neighbourhood stages into a scratch buffer + mia_copy_selected, so a stage that was drawn
// for 4 of 32 samples streams 4 samples, not 32).
// Geometric ops move image and label in ONE launch.  All are bandwidth-bound (read once, write once).
#include "common.h"

__device__ __forceinline__ bool on(const int* apply, int b) { return apply == nullptr || apply[b] != 0; }
__device__ __forceinline__ bool skipped(const int* apply, int b) { return apply != nullptr && apply[b] < 0; }

// ---------------------------------------------------------------- inverse-affine nearest warp (image + label)
// mats[b] = the 6 entries of torchvision's inverse affine matrix (centre frame).  The source pixel of every output pixel
// is INDEX arithmetic and must equal torchvision's tensor path bit for bit (F.affine -> _gen_affine_grid ->
// grid_sample(nearest, zeros, align_corners=False), joint_transform.py:189-190), so the fp32 operations are pinned one by
// one, contraction off:
//   base   = (x + 0.5 - W/2, y + 0.5 - H/2, 1)                    linspace with step exactly 1
//   theta' = theta / (0.5 W, 0.5 H)                               one correctly rounded division per entry
//   g      = base @ theta'^T as the CPU sgemm accumulates it:     acc = x*t0; acc = fma(y, t1, acc); acc = fma(1, t2, acc)
//   f      = ((g + 1) * size - 1) / 2                             grid_sampler unnormalize, align_corners=False
//   src    = round-half-even(f), zero outside [0, size-1]
// (checked against torch-CPU on 4e7 pixels of random rotations / scales: 0 differing indices; the orders
// (x*t0 + y*t1) + t2 without fma and fma(x, t0, fma(y, t1, t2)) differ on ~1e-6 of the pixels.)
struct AffineSrc { int idx; bool inside; };

__device__ __forceinline__ AffineSrc affine_src(const float* __restrict__ m, int x, int y, int w, int h) {
#pragma clang fp contract(off)
  const float hx = 0.5f * (float)w, hy = 0.5f * (float)h;
  const float xb = ((float)x + 0.5f) - hx, yb = ((float)y + 0.5f) - hy;
  const float t0 = m[0] / hx, t1 = m[1] / hx, t2 = m[2] / hx, t3 = m[3] / hy, t4 = m[4] / hy, t5 = m[5] / hy;
  float gx = xb * t0, gy = xb * t3;
  gx = __builtin_fmaf(yb, t1, gx); gy = __builtin_fmaf(yb, t4, gy);
  gx = gx + t2; gy = gy + t5;
  const float fx = (((gx + 1.f) * (float)w) - 1.f) * 0.5f, fy = (((gy + 1.f) * (float)h) - 1.f) * 0.5f;
  const float rx = __builtin_rintf(fx), ry = __builtin_rintf(fy);
  AffineSrc s;
  s.inside = (rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1));
  s.idx = s.inside ? (int)ry * w + (int)rx : 0;
  return s;
}

// VEC = 4: a thread owns four consecutive output pixels of one row (W % 4 == 0): the gathers stay scalar (nearest
// sampling reads arbitrary source pixels) but every store is 16 bytes -- one for the image, two for the int64 labels.
template <int VEC>
__global__ void affine_nearest_kernel(const float* __restrict__ img_in, float* __restrict__ img_out,
                                      const long long* __restrict__ lab_in, long long* __restrict__ lab_out, int nb, int c,
                                      int h, int w, const float* __restrict__ mats, const int* __restrict__ apply) {
  const int64_t hw = (int64_t)h * w, groups = hw / VEC, total = (int64_t)nb * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / groups);
    const int64_t p = (i - (int64_t)b * groups) * VEC;
    const int y = (int)(p / w), x0 = (int)(p - (int64_t)y * w);
    if (skipped(apply, b)) continue;
    const bool act = on(apply, b);
    int src[VEC];
    bool ins[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (act) {
        const AffineSrc s = affine_src(mats + b * 6, x0 + j, y, w, h);
        src[j] = s.idx; ins[j] = s.inside;
      } else {
        src[j] = (int)p + j; ins[j] = true;
      }
    }
    if (img_in) {
      for (int ch = 0; ch < c; ++ch) {
        const float* pi = img_in + ((int64_t)b * c + ch) * hw;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = ins[j] ? pi[src[j]] : 0.f;
        float* po = img_out + ((int64_t)b * c + ch) * hw + p;
        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(po) = f32x4{v[0], v[1], v[2], v[3]};
        else po[0] = v[0];
      }
    }
    if (lab_in) {
      const long long* pl = lab_in + (int64_t)b * hw;
      long long v[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = ins[j] ? pl[src[j]] : 0;
      long long* po = lab_out + (int64_t)b * hw + p;
      if constexpr (VEC == 4) {
        typedef __attribute__((ext_vector_type(2))) long long i64x2;
        reinterpret_cast<i64x2*>(po)[0] = i64x2{v[0], v[1]};
        reinterpret_cast<i64x2*>(po)[1] = i64x2{v[2], v[3]};
      } else {
        po[0] = v[0];
      }
    }
  }
}

extern "C" int mia_affine_nearest(const float* img_in, float* img_out, const long long* lab_in, long long* lab_out, int nb,
                                  int c, int h, int w, const float* mats, const int* apply, void* stream) {
  MIA_CHECK_ARG((img_in || lab_in) && mats && nb > 0 && h > 0 && w > 0, "mia_affine_nearest: bad arguments");
  MIA_CHECK_ARG((img_in == nullptr) == (img_out == nullptr) && (lab_in == nullptr) == (lab_out == nullptr), "mia_affine_nearest: in/out mismatch");
  MIA_CHECK_ARG((int64_t)h * w < ((int64_t)1 << 31), "mia_affine_nearest: image too large");
  const bool vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(img_out) | reinterpret_cast<uintptr_t>(lab_out)) & 15) == 0;
  const int64_t total = (int64_t)nb * h * w / (vec ? 4 : 1);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec) hipLaunchKernelGGL(affine_nearest_kernel<4>, dim3(blocks), dim3(256), 0, st, img_in, img_out, lab_in, lab_out, nb, c, h, w, mats, apply);
  else hipLaunchKernelGGL(affine_nearest_kernel<1>, dim3(blocks), dim3(256), 0, st, img_in, img_out, lab_in, lab_out, nb, c, h, w, mats, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- elastic deformation (image + label)
// north_star names an "elastic" augmentation; the reference has none, so this kernel follows its OWN spec
// (docs: transforms/hip/joint_transform.py::RandomElastic, restated on the CPU in oracle/transforms_ref.py::apply_elastic):
// the U-Net paper's scheme -- random displacement vectors on a coarse (gh x gw) grid of control points spanning the image
// corner to corner, per-pixel displacement by bilinear interpolation of the grid, image sampled bilinearly (zero outside),
// label sampled at the nearest source pixel (round half even, zero outside).  Every fp32 operation below is pinned
// (contraction off) so label maps are bit-exact against the restatement:
//   u = x * ((gw-1)/(W-1)), j0 = min(int(u), gw-2), tu = u - j0      (same for v, i0, tv along y)
//   d = (1-tv) * ((1-tu)*D[i0][j0] + tu*D[i0][j0+1]) + tv * ((1-tu)*D[i0+1][j0] + tu*D[i0+1][j0+1])     per component
//   (sx, sy) = (x + d_x, y + d_y)
// disp: [B][2][gh][gw] fp32, component 0 = x displacement, 1 = y displacement, in pixels.
template <int VEC>
__global__ void elastic_warp_kernel(const float* __restrict__ img_in, float* __restrict__ img_out,
                                    const long long* __restrict__ lab_in, long long* __restrict__ lab_out, int nb, int c, int h,
                                    int w, const float* __restrict__ disp, int gh, int gw, const int* __restrict__ apply) {
#pragma clang fp contract(off)
  const int64_t hw = (int64_t)h * w, groups = hw / VEC, total = (int64_t)nb * groups;
  const float su = w > 1 ? (float)(gw - 1) / (float)(w - 1) : 0.f, sv = h > 1 ? (float)(gh - 1) / (float)(h - 1) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / groups);
    const int64_t p = (i - (int64_t)b * groups) * VEC;
    const int y = (int)(p / w), x0 = (int)(p - (int64_t)y * w);
    if (skipped(apply, b)) continue;
    const bool act = on(apply, b);
    const float* D = disp + (int64_t)b * 2 * gh * gw;
    const float v = (float)y * sv;
    int i0 = (int)v; i0 = i0 < gh - 2 ? i0 : gh - 2; i0 = i0 < 0 ? 0 : i0;
    const float tv = v - (float)i0;
    float iv[VEC];
    long long lv[VEC];
    for (int ch = -1; ch < c; ++ch) {  // ch == -1: the label plane
      if (ch < 0 && !lab_in) continue;
      if (ch >= 0 && !img_in) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int x = x0 + j;
        float sx = (float)x, sy = (float)y;
        if (act) {
          const float u = (float)x * su;
          int j0 = (int)u; j0 = j0 < gw - 2 ? j0 : gw - 2; j0 = j0 < 0 ? 0 : j0;
          const float tu = u - (float)j0;
          float d[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float* g = D + (int64_t)k * gh * gw + (int64_t)i0 * gw + j0;
            const float top = (1.f - tu) * g[0] + tu * g[1];
            const float bot = (1.f - tu) * g[gw] + tu * g[gw + 1];
            d[k] = (1.f - tv) * top + tv * bot;
          }
          sx = (float)x + d[0];
          sy = (float)y + d[1];
        }
        if (ch < 0) {
          const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
          const bool in = rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1);
          lv[j] = in ? lab_in[(int64_t)b * hw + (int64_t)(int)ry * w + (int)rx] : 0;
        } else {
          const float fx = __builtin_floorf(sx), fy = __builtin_floorf(sy);
          const float ax = sx - fx, ay = sy - fy;
          const int xi = (int)fx, yi = (int)fy;
          const float* pi = img_in + ((int64_t)b * c + ch) * hw;
          auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? pi[(int64_t)yy * w + xx] : 0.f; };
          const float top = (1.f - ax) * at(yi, xi) + ax * at(yi, xi + 1);
          const float bot = (1.f - ax) * at(yi + 1, xi) + ax * at(yi + 1, xi + 1);
          iv[j] = (1.f - ay) * top + ay * bot;
        }
      }
      if (ch < 0) {
        long long* po = lab_out + (int64_t)b * hw + p;
#pragma unroll
        for (int j = 0; j < VEC; ++j) po[j] = lv[j];
      } else {
        float* po = img_out + ((int64_t)b * c + ch) * hw + p;
        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(po) = f32x4{iv[0], iv[1], iv[2], iv[3]};
        else po[0] = iv[0];
      }
    }
  }
}

extern "C" int mia_elastic_warp(const float* img_in, float* img_out, const long long* lab_in, long long* lab_out, int nb, int c,
                                int h, int w, const float* disp, int gh, int gw, const int* apply, void* stream) {
  MIA_CHECK_ARG((img_in || lab_in) && disp && nb > 0 && h > 0 && w > 0 && gh >= 2 && gw >= 2, "mia_elastic_warp: bad arguments");
  MIA_CHECK_ARG((img_in == nullptr) == (img_out == nullptr) && (lab_in == nullptr) == (lab_out == nullptr), "mia_elastic_warp: in/out mismatch");
  MIA_CHECK_ARG((int64_t)h * w < ((int64_t)1 << 31), "mia_elastic_warp: image too large");
  const bool vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(img_out) | reinterpret_cast<uintptr_t>(lab_out)) & 15) == 0;
  const int64_t total = (int64_t)nb * h * w / (vec ? 4 : 1);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec) hipLaunchKernelGGL(elastic_warp_kernel<4>, dim3(blocks), dim3(256), 0, st, img_in, img_out, lab_in, lab_out, nb, c, h, w, disp, gh, gw, apply);
  else hipLaunchKernelGGL(elastic_warp_kernel<1>, dim3(blocks), dim3(256), 0, st, img_in, img_out, lab_in, lab_out, nb, c, h, w, disp, gh, gw, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- rot90 (k quarter turns, torch.rot90 on (-2,-1)) + flips
// out[b][.][y][x] = in[b][.][sy][sx].  Square or non-square: for odd k the output is W x H.
template <typename T>
__global__ void rot_flip_kernel(const T* __restrict__ in, T* __restrict__ out, int nb, int c, int h, int w, int k, int flip_h,
                                int flip_w) {
  const int oh = (k & 1) ? w : h, ow = (k & 1) ? h : w;
  const int64_t ohw = (int64_t)oh * ow, total = (int64_t)nb * c * ohw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bc = i / ohw, p = i - bc * ohw;
    int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    int sy, sx;
    // torch.rot90(x, k, (H, W)): k=1 -> out[y][x] = in[x][W-1-y]
    switch (k & 3) {
      case 0: sy = y; sx = x; break;
      case 1: sy = x; sx = w - 1 - y; break;
      case 2: sy = h - 1 - y; sx = w - 1 - x; break;
      default: sy = h - 1 - x; sx = y; break;
    }
    if (flip_h) sy = h - 1 - sy;
    if (flip_w) sx = w - 1 - sx;
    out[i] = in[bc * (int64_t)h * w + (int64_t)sy * w + sx];
  }
}

extern "C" int mia_rot90_flip(const void* in, void* out, int elem_bytes, int nb, int c, int h, int w, int k, int flip_h,
                              int flip_w, void* stream) {
  MIA_CHECK_ARG(in && out && nb > 0 && c > 0 && h > 0 && w > 0, "mia_rot90_flip: bad arguments");
  MIA_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "mia_rot90_flip: element size %d not 4 or 8", elem_bytes);
  const int64_t total = (int64_t)nb * c * h * w;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(rot_flip_kernel<unsigned int>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned int*>(in), static_cast<unsigned int*>(out), nb, c, h, w, k, flip_h, flip_w);
  else
    hipLaunchKernelGGL(rot_flip_kernel<unsigned long long>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned long long*>(in), static_cast<unsigned long long*>(out), nb, c, h, w, k, flip_h, flip_w);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- crop window (RandomCrop2D, joint_transform.py:130-155)
// out[b][.][y][x] = in[b][.][top[b] + y][left[b] + x]; F.crop with an in-range window (RandomCrop.get_params never pads).
template <typename T>
__global__ void crop_kernel(const T* __restrict__ in, T* __restrict__ out, int nb, int c, int h, int w, int oh, int ow,
                            const int* __restrict__ top, const int* __restrict__ left) {
  const int64_t ohw = (int64_t)oh * ow, per = (int64_t)c * ohw, total = (int64_t)nb * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const int64_t r = i - (int64_t)b * per;
    const int ch = (int)(r / ohw);
    const int64_t p = r - (int64_t)ch * ohw;
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    out[i] = in[(((int64_t)b * c + ch) * h + top[b] + y) * w + left[b] + x];
  }
}

extern "C" int mia_crop(const void* in, void* out, int elem_bytes, int nb, int c, int h, int w, int oh, int ow, const int* top,
                        const int* left, void* stream) {
  MIA_CHECK_ARG(in && out && top && left && nb > 0 && c > 0 && h > 0 && w > 0, "mia_crop: bad arguments");
  MIA_CHECK_ARG(oh > 0 && ow > 0 && oh <= h && ow <= w, "mia_crop: window %dx%d does not fit %dx%d", oh, ow, h, w);
  MIA_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "mia_crop: element size %d not 4 or 8", elem_bytes);
  const int64_t total = (int64_t)nb * c * oh * ow;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(crop_kernel<unsigned int>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned int*>(in), static_cast<unsigned int*>(out), nb, c, h, w, oh, ow, top, left);
  else
    hipLaunchKernelGGL(crop_kernel<unsigned long long>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned long long*>(in), static_cast<unsigned long long*>(out), nb, c, h, w, oh, ow, top, left);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- gaussian blur (k x k outer-product kernel, reflect pad)
#define BLUR_MAXK 9
__global__ void gaussian_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int nb, int c, int h, int w,
                                     const float* __restrict__ sigma, const int* __restrict__ ksize, const int* __restrict__ apply) {
  const int64_t hw = (int64_t)h * w, total = (int64_t)nb * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bc = i / hw, p = i - bc * hw;
    const int b = (int)(bc / c);
    if (skipped(apply, b)) continue;
    if (!on(apply, b)) { out[i] = in[i]; continue; }
    const int k = ksize[b], r = k / 2;
    const float s = sigma[b];
    float k1[BLUR_MAXK];
    float ksum = 0.f;
#pragma unroll
    for (int j = 0; j < BLUR_MAXK; ++j) {
      const float xx = (float)(j - r) / s;
      k1[j] = (j < k) ? expf(-0.5f * xx * xx) : 0.f;
      ksum += k1[j];
    }
#pragma unroll
    for (int j = 0; j < BLUR_MAXK; ++j) k1[j] /= ksum;
    const int y = (int)(p / w), x = (int)(p - (int64_t)y * w);
    const float* src = in + bc * hw;
    float acc = 0.f;
    for (int a = 0; a < k; ++a) {
      int yy = y + a - r;
      yy = yy < 0 ? -yy : (yy >= h ? 2 * h - 2 - yy : yy);
      for (int bb = 0; bb < k; ++bb) {
        int xx = x + bb - r;
        xx = xx < 0 ? -xx : (xx >= w ? 2 * w - 2 - xx : xx);
        acc += (k1[a] * k1[bb]) * src[(int64_t)yy * w + xx];
      }
    }
    out[i] = acc;
  }
}

extern "C" int mia_gaussian_blur(const float* in, float* out, int nb, int c, int h, int w, const float* sigma, const int* ksize,
                                 int max_ksize, const int* apply, void* stream) {
  MIA_CHECK_ARG(in && out && sigma && ksize && nb > 0 && c > 0, "mia_gaussian_blur: bad arguments");
  MIA_CHECK_ARG(max_ksize >= 1 && max_ksize <= BLUR_MAXK && (max_ksize & 1), "mia_gaussian_blur: kernel size %d not odd in [1,%d]", max_ksize, BLUR_MAXK);
  MIA_CHECK_ARG(h > max_ksize / 2 && w > max_ksize / 2, "mia_gaussian_blur: reflect padding needs H,W > k/2");
  const int64_t total = (int64_t)nb * c * h * w;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(gaussian_blur_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, nb, c, h, w, sigma, ksize, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- per-sample statistics over C*H*W: stats[b] = (sum, sumsq) in double
// gray=1 with c==3: statistics of 0.2989 r + 0.587 g + 0.114 b (torchvision rgb_to_grayscale) over H*W.
__global__ void sample_stats_partial_kernel(const float* __restrict__ in, int c, int64_t hw, int gray, int slabs,
                                            float* __restrict__ part, const int* __restrict__ apply) {
  __shared__ float red[16];
  const int b = blockIdx.x / slabs, s = blockIdx.x % slabs;
  if (skipped(apply, b)) {  // nobody reads this sample's statistics: leave zeros (uniform per block)
    if (threadIdx.x == 0) { part[(size_t)blockIdx.x * 2] = 0.f; part[(size_t)blockIdx.x * 2 + 1] = 0.f; }
    return;
  }
  const int64_t n = (gray && c == 3) ? hw : (int64_t)c * hw;
  const int64_t per = (n + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < n ? r0 + per : n;
  const float* src = in + (int64_t)b * c * hw;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
    const float v = (gray && c == 3) ? (0.2989f * src[i] + 0.587f * src[hw + i] + 0.114f * src[2 * hw + i]) : src[i];
    s1 += v; s2 += v * v;
  }
  float r = block_sum(s1, red); if (threadIdx.x == 0) part[(size_t)blockIdx.x * 2] = r;
  r = block_sum(s2, red); if (threadIdx.x == 0) part[(size_t)blockIdx.x * 2 + 1] = r;
}

__global__ void sample_stats_final_kernel(const float* __restrict__ part, int nb, int slabs, int64_t n, float* __restrict__ out) {
  // out[b] = (mean, unbiased std)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double s1 = 0, s2 = 0;
  for (int s = 0; s < slabs; ++s) { s1 += part[((size_t)b * slabs + s) * 2]; s2 += part[((size_t)b * slabs + s) * 2 + 1]; }
  const double mean = s1 / (double)n;
  double var = n > 1 ? (s2 - (double)n * mean * mean) / (double)(n - 1) : 0.0;
  if (var < 0) var = 0;
  out[b * 2] = (float)mean;
  out[b * 2 + 1] = (float)sqrt(var);
}

#define STAT_SLABS 64
extern "C" int mia_sample_stats_workspace(int nb) { return nb * STAT_SLABS * 2; }

static int sample_stats_run(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std, const int* apply, void* stream);

extern "C" int mia_sample_stats(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std, void* stream) {
  return sample_stats_run(in, nb, c, hw, gray, workspace, mean_std, nullptr, stream);
}

// statistics of the samples with apply[b] >= 0 only (the others are not read; their rows of mean_std are zero)
extern "C" int mia_sample_stats_sel(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std,
                                    const int* apply, void* stream) {
  return sample_stats_run(in, nb, c, hw, gray, workspace, mean_std, apply, stream);
}

// out[b] = in[b] for every sample with apply[b] > 0 (16-byte units; the copy-back half of a neighbourhood stage that ran into
// a scratch buffer on the selected samples only)
__global__ void copy_selected_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t units, int nb, const int* __restrict__ apply) {
  const int64_t total = units * nb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / units);
    if (apply[b] > 0) out[i] = in[i];
  }
}

extern "C" int mia_copy_selected(const void* in, void* out, int64_t bytes_per_sample, int nb, const int* apply, void* stream) {
  MIA_CHECK_ARG(in && out && apply && nb > 0 && bytes_per_sample > 0 && bytes_per_sample % 16 == 0, "mia_copy_selected: bad arguments");
  MIA_CHECK_ARG(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "mia_copy_selected: unaligned tensors");
  const int64_t total = bytes_per_sample / 16 * nb;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(copy_selected_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u32x4*>(in),
                     static_cast<u32x4*>(out), bytes_per_sample / 16, nb, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

static int sample_stats_run(const float* in, int nb, int c, int64_t hw, int gray, float* workspace, float* mean_std, const int* apply, void* stream) {
  MIA_CHECK_ARG(in && workspace && mean_std && nb > 0 && c > 0 && hw > 0, "mia_sample_stats: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sample_stats_partial_kernel, dim3(nb * STAT_SLABS), dim3(256), 0, st, in, c, hw, gray, STAT_SLABS, workspace, apply);
  const int64_t n = (gray && c == 3) ? hw : (int64_t)c * hw;
  hipLaunchKernelGGL(sample_stats_final_kernel, dim3(ceil_div(nb, 64)), dim3(64), 0, st, workspace, nb, STAT_SLABS, n, mean_std);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- elementwise intensity ops (in-place safe)
#define EW_GAMMA 0     /* out = pow(in, p0[b])                                   image_transform.py:31 */
#define EW_CONTRAST 1  /* out = clamp(p0[b]*in + (1-p0[b])*mean[b], 0, 1)         torchvision _blend    */
#define EW_NOISE 2     /* out = clamp(in + aux, 0, 1) with explicit noise tensor  image_transform.py:130-132 */
#define EW_ZSCORE 3    /* out = (in - mean[b]) / max(std[b], 1e-8)                normalization.py:17-21 */
__global__ void elementwise_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t per_sample, int nb, int op,
                                   const float* __restrict__ p0, const float* __restrict__ mean_std, const float* __restrict__ aux,
                                   const int* __restrict__ apply) {
  const int64_t total = per_sample * nb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_sample);
    if (skipped(apply, b)) continue;
    float v = in[i];
    if (on(apply, b)) {
      if (op == EW_GAMMA) v = powf(v, p0[b]);
      else if (op == EW_CONTRAST) { const float f = p0[b]; v = f * v + (1.f - f) * mean_std[b * 2]; v = fminf(fmaxf(v, 0.f), 1.f); }
      else if (op == EW_NOISE) { v = v + aux[i]; v = fminf(fmaxf(v, 0.f), 1.f); }
      else { const float sd = fmaxf(mean_std[b * 2 + 1], 1e-8f); v = (v - mean_std[b * 2]) / sd; }
    }
    out[i] = v;
  }
}

extern "C" int mia_elementwise(const float* in, float* out, int64_t per_sample, int nb, int op, const float* p0,
                               const float* mean_std, const float* aux, const int* apply, void* stream) {
  MIA_CHECK_ARG(in && out && per_sample > 0 && nb > 0 && op >= 0 && op <= EW_ZSCORE, "mia_elementwise: bad arguments");
  MIA_CHECK_ARG((op != EW_GAMMA && op != EW_CONTRAST) || p0, "mia_elementwise: missing parameter array");
  MIA_CHECK_ARG((op != EW_CONTRAST && op != EW_ZSCORE) || mean_std, "mia_elementwise: missing statistics");
  MIA_CHECK_ARG(op != EW_NOISE || aux, "mia_elementwise: missing noise tensor");
  const int64_t total = per_sample * nb;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(elementwise_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, per_sample, nb, op, p0, mean_std, aux, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- gaussian noise with a counter-based generator (Philox4x32-10)
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ void noise_clip_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t per_sample, int nb,
                                  const float* __restrict__ sigma, unsigned long long seed, unsigned long long offset,
                                  const int* __restrict__ apply) {
  const int64_t total = per_sample * nb, quads = (total + 3) / 4;
  for (int64_t qd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += (int64_t)gridDim.x * blockDim.x) {
    if (apply != nullptr) {  // a quad spans at most two samples: nothing to do when neither is selected for anything
      const int64_t last = qd * 4 + 3 < total ? qd * 4 + 3 : total - 1;
      if (apply[(int)(qd * 4 / per_sample)] < 0 && apply[(int)(last / per_sample)] < 0) continue;
    }
    unsigned c0 = (unsigned)qd, c1 = (unsigned)(qd >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((c0 >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((c1 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((c2 >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((c3 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    const float z[4] = {r0 * cosf(6.2831853f * u1), r0 * sinf(6.2831853f * u1), r1 * cosf(6.2831853f * u3), r1 * sinf(6.2831853f * u3)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = qd * 4 + e;
      if (i >= total) break;
      const int b = (int)(i / per_sample);
      if (skipped(apply, b)) continue;
      float v = in[i];
      if (on(apply, b)) { v += sigma[b] * z[e]; v = fminf(fmaxf(v, 0.f), 1.f); }
      out[i] = v;
    }
  }
}

extern "C" int mia_noise_clip(const float* in, float* out, int64_t per_sample, int nb, const float* sigma, uint64_t seed,
                              uint64_t offset, const int* apply, void* stream) {
  MIA_CHECK_ARG(in && out && sigma && per_sample > 0 && nb > 0, "mia_noise_clip: bad arguments");
  const int64_t quads = (per_sample * nb + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
  hipLaunchKernelGGL(noise_clip_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, per_sample, nb, sigma,
                     (unsigned long long)seed, (unsigned long long)offset, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// ---------------------------------------------------------------- resize
__device__ __forceinline__ float src_index(float scale, int dst) {  // area_pixel_compute_source_index, align_corners=False
  const float s = scale * ((float)dst + 0.5f) - 0.5f;
  return s < 0.f ? 0.f : s;
}

// bilinear (no antialias) with optional per-sample low-res simulation: when lowres != null the source image is
// first viewed through a nearest-exact downsample to (lh[b], lw[b]) (SimulateLowRes, image_transform.py:218-225).
__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int nb, int c, int h, int w, int oh,
                                       int ow, const int* __restrict__ lowres, const int* __restrict__ apply) {
  const int64_t ohw = (int64_t)oh * ow, total = (int64_t)nb * c * ohw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bc = i / ohw, p = i - bc * ohw;
    const int b = (int)(bc / c);
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    const float* src = in + bc * (int64_t)h * w;
    if (lowres != nullptr && skipped(apply, b)) continue;
    if (lowres != nullptr && !on(apply, b)) { out[i] = src[(int64_t)y * w + x]; continue; }
    const int lh = lowres ? lowres[b * 2] : h, lw = lowres ? lowres[b * 2 + 1] : w;
    const float sy = src_index((float)lh / (float)oh, y), sx = src_index((float)lw / (float)ow, x);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    auto fetch = [&](int yy, int xx) -> float {
      if (lowres) {  // nearest-exact: src = min(floor((dst+0.5)*in/out), in-1)
        const int iy = min((int)floorf(((float)yy + 0.5f) * ((float)h / (float)lh)), h - 1);
        const int ix = min((int)floorf(((float)xx + 0.5f) * ((float)w / (float)lw)), w - 1);
        return src[(int64_t)iy * w + ix];
      }
      return src[(int64_t)yy * w + xx];
    };
    const float top = (1.f - lx) * fetch(y0, x0) + lx * fetch(y0, x1);
    const float bot = (1.f - lx) * fetch(y1, x0) + lx * fetch(y1, x1);
    out[i] = (1.f - ly) * top + ly * bot;
  }
}

extern "C" int mia_resize_bilinear(const float* in, float* out, int nb, int c, int h, int w, int oh, int ow, const int* lowres_hw,
                                   const int* apply, void* stream) {
  MIA_CHECK_ARG(in && out && nb > 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "mia_resize_bilinear: bad arguments");
  MIA_CHECK_ARG(lowres_hw == nullptr || (oh == h && ow == w), "mia_resize_bilinear: low-res simulation keeps the size");
  const int64_t total = (int64_t)nb * c * oh * ow;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, nb, c, h, w, oh, ow, lowres_hw, apply);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// adjoint of resize_bilinear (no low-res): din += scatter of dout.  Used by the deep-supervision heads' Upsample.
__global__ void resize_bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int nb, int c, int h, int w,
                                           int oh, int ow) {
  const int64_t ohw = (int64_t)oh * ow, total = (int64_t)nb * c * ohw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bc = i / ohw, p = i - bc * ohw;
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    const float sy = src_index((float)h / (float)oh, y), sx = src_index((float)w / (float)ow, x);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, g = dout[i];
    float* dst = din + bc * (int64_t)h * w;
    atomicAdd(dst + (int64_t)y0 * w + x0, (1.f - ly) * (1.f - lx) * g);
    atomicAdd(dst + (int64_t)y0 * w + x1, (1.f - ly) * lx * g);
    atomicAdd(dst + (int64_t)y1 * w + x0, ly * (1.f - lx) * g);
    atomicAdd(dst + (int64_t)y1 * w + x1, ly * lx * g);
  }
}

extern "C" int mia_resize_bilinear_bwd(const float* dout, float* din_zeroed, int nb, int c, int h, int w, int oh, int ow, void* stream) {
  MIA_CHECK_ARG(dout && din_zeroed && nb > 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "mia_resize_bilinear_bwd: bad arguments");
  const int64_t total = (int64_t)nb * c * oh * ow;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dout, din_zeroed, nb, c, h, w, oh, ow);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// antialiased bilinear, one separable pass along W (axis=1) or H (axis=0): torch _upsample_bilinear2d_aa
// (triangle filter, support = max(scale,1), weights normalised per output index).
__global__ void resize_aa_pass_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t planes, int h, int w, int osize,
                                      int axis) {
  const int isize = axis ? w : h;
  const int oh = axis ? h : osize, ow = axis ? osize : w;
  const float scale = (float)isize / (float)osize;
  const float support = scale >= 1.f ? scale : 1.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const int64_t ohw = (int64_t)oh * ow, total = planes * ohw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / ohw, p = i - pl * ohw;
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    const int o = axis ? x : y;
    const float center = scale * ((float)o + 0.5f);
    int lo = (int)(center - support + 0.5f); if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5f); if (hi > isize) hi = isize;
    float wsum = 0.f, acc = 0.f;
    const float* src = in + pl * (int64_t)h * w;
    for (int j = lo; j < hi; ++j) {
      float t = ((float)j - center + 0.5f) * invscale;
      t = t < 0.f ? -t : t;
      const float wt = t < 1.f ? 1.f - t : 0.f;
      wsum += wt;
      acc += wt * (axis ? src[(int64_t)y * w + j] : src[(int64_t)j * w + x]);
    }
    out[i] = wsum != 0.f ? acc / wsum : 0.f;
  }
}

extern "C" int mia_resize_bilinear_aa(const float* in, float* tmp, float* out, int nb, int c, int h, int w, int oh, int ow, void* stream) {
  MIA_CHECK_ARG(in && tmp && out && nb > 0 && c > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "mia_resize_bilinear_aa: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t planes = (int64_t)nb * c;
  // horizontal pass: [h][w] -> tmp [h][ow]; vertical pass: [h][ow] -> out [oh][ow]
  int64_t total = planes * h * ow;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_aa_pass_kernel, dim3(blocks), dim3(256), 0, st, in, tmp, planes, h, w, ow, 1);
  total = planes * oh * ow;
  blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_aa_pass_kernel, dim3(blocks), dim3(256), 0, st, tmp, out, planes, h, ow, oh, 0);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// nearest (legacy "nearest": src = min(floor(dst * in/out), in-1)), 4- or 8-byte elements (float images, int64 labels)
template <typename T>
__global__ void resize_nearest_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t planes, int h, int w, int oh, int ow) {
  const int64_t ohw = (int64_t)oh * ow, total = planes * ohw;
  const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / ohw, p = i - pl * ohw;
    const int y = (int)(p / ow), x = (int)(p - (int64_t)y * ow);
    const int sy = min((int)floorf((float)y * sh), h - 1), sx = min((int)floorf((float)x * sw), w - 1);
    out[i] = in[pl * (int64_t)h * w + (int64_t)sy * w + sx];
  }
}

extern "C" int mia_resize_nearest(const void* in, void* out, int elem_bytes, int64_t planes, int h, int w, int oh, int ow, void* stream) {
  MIA_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "mia_resize_nearest: bad arguments");
  MIA_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "mia_resize_nearest: element size %d not 4 or 8", elem_bytes);
  const int64_t total = planes * oh * ow;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(resize_nearest_kernel<unsigned int>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned int*>(in), static_cast<unsigned int*>(out), planes, h, w, oh, ow);
  else
    hipLaunchKernelGGL(resize_nearest_kernel<unsigned long long>, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned long long*>(in), static_cast<unsigned long long*>(out), planes, h, w, oh, ow);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
