// Flat-buffer optimizer step with clip-by-global-norm folded in (gfx950).
//
// Reference: ALTrainer.train_step (src/training/al_trainer.py:1374-1379): zero_grad, backward,
// clip_grad_norm_(params, 10.0), optimizer.step() with torch.optim.Adam / AdamW(betas=(0.9,0.999)) or
// SGD(momentum=0.9) (al_trainer.py:744-761).  All parameters / gradients / moments live in ONE flat
// fp32 buffer each, so the global L2 norm is one two-stage reduction and the update is one 16-byte
// vectorised stream (4 reads + 3 writes per element) with the clip coefficient read from device
// memory -- no host synchronisation anywhere in the step.
#include "common.h"

__global__ void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  const int64_t n4 = n / 4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = x4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += x[i] * x[i];
  const float r = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// out[0] = L2 norm of grad_scale*g, out[1] = clip coefficient min(1, max_norm/(norm+1e-6))  (torch clip_grad_norm_)
__global__ void norm_final_kernel(const float* __restrict__ part, int nblk, float max_norm, float grad_scale, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < blockDim.x; ++i) t += red[i];
    const float nrm = grad_scale * (float)sqrt(t);
    float coef = max_norm / (nrm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    out[0] = nrm;
    out[1] = max_norm > 0.f ? coef : 1.f;
  }
}

#define GN_BLOCKS 1024
extern "C" int mia_grad_norm_workspace(void) { return GN_BLOCKS; }

extern "C" int mia_grad_norm(const float* grad, int64_t n, float max_norm, float grad_scale, float* workspace, float* out, void* stream) {
  MIA_CHECK_ARG(grad && workspace && out && n > 0, "mia_grad_norm: bad arguments");
  MIA_CHECK_ARG((reinterpret_cast<uintptr_t>(grad) & 15) == 0, "mia_grad_norm: gradient buffer must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = (int)((n / 4 + 255) / 256 < GN_BLOCKS ? (n / 4 + 255) / 256 + 1 : GN_BLOCKS);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, grad, n, workspace);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, workspace, blocks, max_norm, grad_scale, out);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

#define OPT_ADAM 0
#define OPT_ADAMW 1
#define OPT_SGD 2

// hp (device, fp32): [0]=lr [1]=bias_correction1 [2]=bias_correction2 (host computes 1-beta^t in double)
// dyn (device, fp32[4]; nullptr = use the arguments): lr, bias_correction1, bias_correction2, first_step != 0 -- the per-step
// scalars of a step that is replayed from a captured hipGraph (mia_optim_step_dyn), where kernel arguments are frozen
__global__ void optim_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, int64_t n, int kind, float lr, float beta1, float beta2, float eps,
                                  float wd, float bc1, float bc2, int first_step, const float* __restrict__ clip,
                                  float grad_scale, const float* __restrict__ dyn) {
  if (dyn != nullptr) { lr = dyn[0]; bc1 = dyn[1]; bc2 = dyn[2]; first_step = dyn[3] != 0.f; }
  const float cs = (clip ? clip[1] : 1.f) * grad_scale;
  const float sq_bc2 = sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * cs, pi = p[i];
    if (kind == OPT_SGD) {
      if (wd != 0.f) gi += wd * pi;
      const float buf = first_step ? gi : beta1 * m[i] + gi;
      m[i] = buf;
      p[i] = pi - lr * buf;
    } else {
      if (kind == OPT_ADAMW) pi *= (1.f - lr * wd);
      else if (wd != 0.f) gi += wd * pi;
      const float mi = beta1 * m[i] + (1.f - beta1) * gi;       // torch: lerp(m, g, 1-beta1)
      const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;  // torch: v*beta2 + (1-beta2)*g*g
      m[i] = mi; v[i] = vi;
      const float denom = sqrtf(vi) / sq_bc2 + eps;
      p[i] = pi - (lr / bc1) * (mi / denom);
    }
  }
}

extern "C" int mia_optim_step(float* param, const float* grad, float* m, float* v, int64_t n, int kind, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, int first_step,
                              const float* clip, float grad_scale, void* stream) {
  MIA_CHECK_ARG(param && grad && m && n > 0, "mia_optim_step: bad arguments");
  MIA_CHECK_ARG(kind == OPT_SGD || v != nullptr, "mia_optim_step: Adam needs second-moment buffer");
  MIA_CHECK_ARG(kind >= OPT_ADAM && kind <= OPT_SGD, "mia_optim_step: unknown optimizer %d", kind);
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(optim_step_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, m, v, n, kind,
                     lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, first_step, clip, grad_scale, (const float*)nullptr);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// The 32 bytes a captured step reads its per-iteration scalars from (fp32 {lr, bc1, bc2, first}, u64 {seed, base offset}), written
// by a one-thread kernel whose ARGUMENTS carry the values: they are copied when the launch is enqueued, so a host that runs many
// replays ahead of the device cannot overwrite a staging buffer the device has not read yet (a pinned async copy could).
__global__ void step_dyn_set_kernel(float* __restrict__ f, float lr, float bc1, float bc2, float first, unsigned long long seed,
                                    unsigned long long offset) {
  f[0] = lr; f[1] = bc1; f[2] = bc2; f[3] = first;
  unsigned long long* u = reinterpret_cast<unsigned long long*>(f + 4);
  u[0] = seed; u[1] = offset;
}
extern "C" int mia_step_dyn_set(void* dyn32, float lr, float bias_corr1, float bias_corr2, int first_step, uint64_t seed, uint64_t offset,
                                void* stream) {
  MIA_CHECK_ARG(dyn32 && (reinterpret_cast<uintptr_t>(dyn32) & 15) == 0, "mia_step_dyn_set: needs a 16-byte aligned 32-byte device buffer");
  hipLaunchKernelGGL(step_dyn_set_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<float*>(dyn32), lr, bias_corr1,
                     bias_corr2, first_step ? 1.f : 0.f, (unsigned long long)seed, (unsigned long long)offset);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_optim_step_dyn(float* param, const float* grad, float* m, float* v, int64_t n, int kind, float beta1, float beta2,
                                  float eps, float weight_decay, const float* dyn, const float* clip, float grad_scale, void* stream) {
  MIA_CHECK_ARG(param && grad && m && n > 0 && dyn, "mia_optim_step_dyn: bad arguments");
  MIA_CHECK_ARG(kind == OPT_SGD || v != nullptr, "mia_optim_step_dyn: Adam needs second-moment buffer");
  MIA_CHECK_ARG(kind >= OPT_ADAM && kind <= OPT_SGD, "mia_optim_step_dyn: unknown optimizer %d", kind);
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(optim_step_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, m, v, n, kind,
                     0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 0, clip, grad_scale, dyn);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// x *= clip[1]  (clip_grad_norm_ as a stand-alone op for callers that keep torch.optim)
__global__ void scale_by_dev_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ clip) {
  const float c = clip[1];
  if (c == 1.f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= c;
}

extern "C" int mia_scale_by_clip(float* x, int64_t n, const float* clip, void* stream) {
  MIA_CHECK_ARG(x && clip && n > 0, "mia_scale_by_clip: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(scale_by_dev_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, clip);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
