// Validation / selection reductions over logits (gfx950): the forward-only consumers of the model.
//
// * mia_argmax_dice: `prob = output.softmax(1); pred = prob.argmax(1)` (src/training/al_trainer.py:1430-1431) and the
//   per-class hard Dice of `calculate_metric_percase` (:1539-1556: medpy.metric.dc = 2|A&B|/(|A|+|B|), 0 when the
//   prediction is empty) as one pass: label map out + per-(image, class) counts |P&G|, |P|, |G|.  softmax is monotone,
//   so argmax is taken on the logits (first maximum wins, like torch.argmax).
// * mia_selector_scores: the per-image acquisition scores of the active-learning selectors, fused softmax + reduction:
//   entropy    mean_{c,h,w}( -p * log2(p + smooth) )     (src/activelearning/entropy_selector.py:42-49)
//   confidence mean_{h,w}( -max_c p )                    (confidence_selector.py:42-47)
//   margin     mean_{h,w}( -(p_top1 - p_top2) )          (margin_selector.py:42-48)
// Both read K1 logits (+ one label) per pixel once; wave-shuffle + LDS block reductions, no float atomics.
#include "common.h"

#define MAXK 8
struct MGeom { int64_t sn, sk, sp; };

__global__ void argmax_dice_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, long long* __restrict__ pred,
                                   int64_t hw, int k1, MGeom g, int slabs, float* __restrict__ part /*[B][slabs][K1][3]*/) {
  __shared__ float red[16];
  const int b = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  float ci[MAXK], cp[MAXK], cg[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) { ci[k] = 0.f; cp[k] = 0.f; cg[k] = 0.f; }
  const float* base = logits + b * g.sn;
  for (int64_t p = r0 + threadIdx.x; p < r1; p += blockDim.x) {
    float best = -INFINITY;
    int arg = 0;
    if (logits != nullptr) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < k1) { const float v = base[p * g.sp + k * g.sk]; if (v > best) { best = v; arg = k; } }
      if (pred) pred[(int64_t)b * hw + p] = arg;
    } else {
      arg = (int)pred[(int64_t)b * hw + p];  // label-map mode: `pred` is an INPUT (e.g. the post-processed prediction)
    }
    const int lab = labels ? (int)labels[(int64_t)b * hw + p] : -1;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float isp = arg == k ? 1.f : 0.f, isg = lab == k ? 1.f : 0.f;
        ci[k] += isp * isg; cp[k] += isp; cg[k] += isg;
      }
  }
  if (part == nullptr) return;
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
    if (k < k1) {
      float* dst = part + (((size_t)b * slabs + s) * k1 + k) * 3;
      float r;
      r = block_sum(ci[k], red); if (threadIdx.x == 0) dst[0] = r;
      r = block_sum(cp[k], red); if (threadIdx.x == 0) dst[1] = r;
      r = block_sum(cg[k], red); if (threadIdx.x == 0) dst[2] = r;
    }
}

// counts[b][k][3] = (|P&G|, |P|, |G|), dice[b][k] = |P| > 0 ? 2I/(|P|+|G|) : 0
__global__ void dice_finalize_kernel(const float* __restrict__ part, int nb, int slabs, int k1, float* __restrict__ counts,
                                     float* __restrict__ dice) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * k1) return;
  const int b = i / k1, k = i % k1;
  double a0 = 0, a1 = 0, a2 = 0;
  for (int s = 0; s < slabs; ++s) {
    const float* p = part + (((size_t)b * slabs + s) * k1 + k) * 3;
    a0 += p[0]; a1 += p[1]; a2 += p[2];
  }
  counts[i * 3] = (float)a0; counts[i * 3 + 1] = (float)a1; counts[i * 3 + 2] = (float)a2;
  dice[i] = a1 > 0 ? (float)(2.0 * a0 / (a1 + a2)) : 0.f;
}

extern "C" int mia_argmax_dice_workspace(int nb, int k1, int slabs) { return nb * slabs * k1 * 3; }

extern "C" int mia_argmax_dice(const float* logits, const long long* labels, long long* pred, int nb, int64_t hw, int k1, int64_t sn,
                               int64_t sk, int64_t sp, int slabs, float* workspace, float* counts, float* dice, void* stream) {
  MIA_CHECK_ARG(nb > 0 && hw > 0 && slabs > 0, "mia_argmax_dice: bad arguments");
  MIA_CHECK_ARG(logits || (pred && labels), "mia_argmax_dice: label-map mode (logits == NULL) needs pred (input) and labels");
  MIA_CHECK_ARG(k1 >= 1 && k1 <= MAXK, "mia_argmax_dice: k1=%d not in [1,%d]", k1, MAXK);
  MIA_CHECK_ARG((labels == nullptr) == (counts == nullptr) && (counts == nullptr) == (dice == nullptr) &&
                (counts == nullptr || workspace != nullptr), "mia_argmax_dice: labels, workspace, counts and dice go together");
  MIA_CHECK_ARG(pred || labels, "mia_argmax_dice: nothing to compute");
  hipStream_t st = static_cast<hipStream_t>(stream);
  MGeom g{sn, sk, sp};
  hipLaunchKernelGGL(argmax_dice_kernel, dim3(nb * slabs), dim3(256), 0, st, logits, labels, pred, hw, k1, g, slabs, labels ? workspace : nullptr);
  if (labels) hipLaunchKernelGGL(dice_finalize_kernel, dim3(ceil_div(nb * k1, 64)), dim3(64), 0, st, workspace, nb, slabs, k1, counts, dice);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// scores partials: part[b][slab][3] = (sum entropy terms, sum -max p, sum -(p1 - p2))
__global__ void selector_scores_kernel(const float* __restrict__ logits, int64_t hw, int k1, MGeom g, int slabs, float smooth, float* __restrict__ part) {
  __shared__ float red[16];
  const int b = blockIdx.x / slabs, s = blockIdx.x % slabs;
  const int64_t per = (hw + slabs - 1) / slabs, r0 = s * per, r1 = r0 + per < hw ? r0 + per : hw;
  const float* base = logits + b * g.sn;
  float se = 0.f, sc = 0.f, sm = 0.f;
  for (int64_t p = r0 + threadIdx.x; p < r1; p += blockDim.x) {
    float v[MAXK], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { v[k] = base[p * g.sp + k * g.sk]; mx = fmaxf(mx, v[k]); }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) { v[k] = expf(v[k] - mx); sum += v[k]; }
    const float inv = 1.f / sum;
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < k1) {
        const float pk = v[k] * inv;
        se += -pk * log2f(pk + smooth);
        if (pk > p1) { p2 = p1; p1 = pk; } else if (pk > p2) p2 = pk;
      }
    sc += -p1;
    sm += -(p1 - p2);
  }
  float* dst = part + ((size_t)b * slabs + s) * 3;
  float r;
  r = block_sum(se, red); if (threadIdx.x == 0) dst[0] = r;
  r = block_sum(sc, red); if (threadIdx.x == 0) dst[1] = r;
  r = block_sum(sm, red); if (threadIdx.x == 0) dst[2] = r;
}

__global__ void selector_finalize_kernel(const float* __restrict__ part, int nb, int slabs, int k1, int64_t hw, float* __restrict__ scores) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double e = 0, c = 0, m = 0;
  for (int s = 0; s < slabs; ++s) { const float* p = part + ((size_t)b * slabs + s) * 3; e += p[0]; c += p[1]; m += p[2]; }
  scores[b * 3 + 0] = (float)(e / ((double)hw * k1));
  scores[b * 3 + 1] = (float)(c / (double)hw);
  scores[b * 3 + 2] = (float)(m / (double)hw);
}

extern "C" int mia_selector_scores_workspace(int nb, int slabs) { return nb * slabs * 3; }

extern "C" int mia_selector_scores(const float* logits, int nb, int64_t hw, int k1, int64_t sn, int64_t sk, int64_t sp, float smooth, int slabs,
                                   float* workspace, float* scores, void* stream) {
  MIA_CHECK_ARG(logits && workspace && scores && nb > 0 && hw > 0 && slabs > 0, "mia_selector_scores: bad arguments");
  MIA_CHECK_ARG(k1 >= 2 && k1 <= MAXK, "mia_selector_scores: k1=%d not in [2,%d]", k1, MAXK);
  hipStream_t st = static_cast<hipStream_t>(stream);
  MGeom g{sn, sk, sp};
  hipLaunchKernelGGL(selector_scores_kernel, dim3(nb * slabs), dim3(256), 0, st, logits, hw, k1, g, slabs, smooth, workspace);
  hipLaunchKernelGGL(selector_finalize_kernel, dim3(ceil_div(nb, 64)), dim3(64), 0, st, workspace, nb, slabs, k1, hw, scores);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}
