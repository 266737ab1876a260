// Optimizer tail of the training step (scripts/train_net.py:134-143): clip_grad_norm_ -> AdamW.step ->
// update_ema (engine/optimizer.py:5-22), as TWO multi-tensor launches over a device table of parameter tensors
// instead of ~4 x 600 small PyTorch kernels:
//   (1) grad_sqnorm_kernel : sum of squares of every gradient -> one device scalar (no host sync);
//   (2) adamw_ema_kernel   : clip coefficient from that scalar, decoupled weight decay, moment updates,
//                            bias-corrected step, and the EMA copy of the updated parameter — one pass over
//                            p, g, m, v, ema (HBM-bound: 5 reads + 4 writes of 4 B per element).
// Work is cut into fixed chunks; chunk c covers elements [chunk_off[c], chunk_off[c] + CHUNK) of tensor
// chunk_tensor[c].  Parameter groups (base / backbone / text / temporal-decoder learning rates,
// engine/optimizer.py:38-43) are indices into by-value hyper-parameter arrays, so the per-step learning-rate
// schedule (engine/lr_scheduler.py:212-252) costs no upload.
#pragma once
#include "stcat_platform.h"

#define STCAT_OPT_MAX_GROUPS 8

struct OptTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  float* ema;  // may be null
  long n;
  int group;
  int pad_;
};

struct OptHyper {
  float lr[STCAT_OPT_MAX_GROUPS];
  float wd[STCAT_OPT_MAX_GROUPS];
  float beta1, beta2, eps;
  float bc1, bc2;      // 1 - beta^t
  float max_norm;      // <= 0: no clipping
  float ema_decay;
  int skip_nonfinite;  // 1: a non-finite gradient norm skips the update (the fp16-plane mode's loss-scale policy); 0: the
                       // reference's behaviour — clip_grad_norm_ multiplies by max_norm / inf = 0 and inf * 0 = NaN propagates
};

__global__ void __launch_bounds__(256) grad_sqnorm_kernel(const OptTensor* tab, const int* chunk_tensor,
                                                          const long* chunk_off, int chunk, float* out) {
  __shared__ float part[4];
  const OptTensor t = tab[chunk_tensor[blockIdx.x]];
  const long lo = chunk_off[blockIdx.x];
  const long hi = lo + chunk < t.n ? lo + chunk : t.n;
  float acc = 0.f;
  const bool vec = (((unsigned long)(t.g + lo)) & 15u) == 0;
  if (vec) {
    const long nv = lo + ((hi - lo) / 4) * 4;
    for (long i = lo + (long)threadIdx.x * 4; i < nv; i += 256 * 4) {
      const float4 g4 = stcat_ld4(t.g + i);
      acc += g4.x * g4.x + g4.y * g4.y + g4.z * g4.z + g4.w * g4.w;
    }
    if (threadIdx.x == 0) {
      for (long j = nv; j < hi; ++j) acc += t.g[j] * t.g[j];
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) acc += t.g[i] * t.g[i];
  }
  acc = stcat_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

static __device__ __forceinline__ void stcat_adamw_one(float& p, float g, float& m, float& v, float* ema, float coef,
                                                       float lr, float wd, const OptHyper& h) {
  g *= coef;
  p *= 1.f - lr * wd;                                   // decoupled weight decay (torch.optim.AdamW)
  m += (g - m) * (1.f - h.beta1);                       // exp_avg.lerp_(grad, 1 - beta1)
  v = v * h.beta2 + (1.f - h.beta2) * g * g;
  const float denom = sqrtf(v) / sqrtf(h.bc2) + h.eps;
  p -= (lr / h.bc1) * (m / denom);
  if (ema) *ema = *ema * h.ema_decay + (1.f - h.ema_decay) * p;   // engine/optimizer.py:22
}

__global__ void __launch_bounds__(256) adamw_ema_kernel(const OptTensor* tab, const int* chunk_tensor,
                                                        const long* chunk_off, int chunk, const float* sqnorm,
                                                        OptHyper h) {
  const OptTensor t = tab[chunk_tensor[blockIdx.x]];
  const long lo = chunk_off[blockIdx.x];
  const long hi = lo + chunk < t.n ? lo + chunk : t.n;
  float coef = 1.f;
  if (h.max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float sq = *sqnorm;
    // Mode f16x3p only (skip_nonfinite): a non-finite global gradient norm (an overflow: fp16 planes with too large a loss
    // scale) SKIPS the whole update, as torch.cuda.amp.GradScaler.step does — clipping alone would turn inf * 0 into NaN
    // weights.  The caller sees it in the squared norm AdamW.step() returns (optim.PlaneLossScale reads it).  Every other
    // mode does what the reference does: the NaN propagates.
    if (h.skip_nonfinite && !(sq <= 3.0e38f)) return;
    coef = h.max_norm / (sqrtf(sq) + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  const float lr = h.lr[t.group], wd = h.wd[t.group];
  const bool vec = ((((unsigned long)(t.p + lo)) | ((unsigned long)(t.g + lo)) | ((unsigned long)(t.m + lo)) |
                     ((unsigned long)(t.v + lo)) | ((unsigned long)(t.ema ? t.ema + lo : nullptr))) & 15u) == 0;
  if (vec) {
    const long nv = lo + ((hi - lo) / 4) * 4;
    for (long i = lo + (long)threadIdx.x * 4; i < nv; i += 256 * 4) {
      float4 p4 = stcat_ld4(t.p + i), m4 = stcat_ld4(t.m + i), v4 = stcat_ld4(t.v + i);
      const float4 g4 = stcat_ld4(t.g + i);
      float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t.ema) e4 = stcat_ld4(t.ema + i);
      stcat_adamw_one(p4.x, g4.x, m4.x, v4.x, t.ema ? &e4.x : nullptr, coef, lr, wd, h);
      stcat_adamw_one(p4.y, g4.y, m4.y, v4.y, t.ema ? &e4.y : nullptr, coef, lr, wd, h);
      stcat_adamw_one(p4.z, g4.z, m4.z, v4.z, t.ema ? &e4.z : nullptr, coef, lr, wd, h);
      stcat_adamw_one(p4.w, g4.w, m4.w, v4.w, t.ema ? &e4.w : nullptr, coef, lr, wd, h);
      stcat_st4(t.p + i, p4); stcat_st4(t.m + i, m4); stcat_st4(t.v + i, v4);
      if (t.ema) stcat_st4(t.ema + i, e4);
    }
    if (threadIdx.x == 0) {  // ragged tail (< 4 elements)
      for (long j = nv; j < hi; ++j)
        stcat_adamw_one(t.p[j], t.g[j], t.m[j], t.v[j], t.ema ? t.ema + j : nullptr, coef, lr, wd, h);
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256)
      stcat_adamw_one(t.p[i], t.g[i], t.m[i], t.v[i], t.ema ? t.ema + i : nullptr, coef, lr, wd, h);
  }
}

// standalone clip_grad_norm_: g *= min(1, max_norm / (sqrt(sqnorm) + 1e-6)) over the tensor table, in place and
// layout-agnostic (flat element order of each gradient buffer: row-major and channels_last alike)
__global__ void __launch_bounds__(256) grad_clip_scale_kernel(const OptTensor* tab, const int* chunk_tensor,
                                                              const long* chunk_off, int chunk, const float* sqnorm,
                                                              float max_norm) {
  const OptTensor t = tab[chunk_tensor[blockIdx.x]];
  const long lo = chunk_off[blockIdx.x];
  const long hi = lo + chunk < t.n ? lo + chunk : t.n;
  float coef = max_norm / (sqrtf(*sqnorm) + 1e-6f);
  if (!(coef < 1.f)) return;  // norm within bounds: nothing to do (torch clamps the coefficient to 1)
  float* g = const_cast<float*>(t.g);
  for (long i = lo + threadIdx.x; i < hi; i += 256) g[i] *= coef;
}

// w_ema = w_ema * decay + (1 - decay) * w over a tensor table (update_ema for state that is not a trained
// parameter; trained parameters get their EMA inside adamw_ema_kernel)
__global__ void __launch_bounds__(256) ema_kernel(const OptTensor* tab, const int* chunk_tensor, const long* chunk_off,
                                                  int chunk, float decay) {
  const OptTensor t = tab[chunk_tensor[blockIdx.x]];
  const long lo = chunk_off[blockIdx.x];
  const long hi = lo + chunk < t.n ? lo + chunk : t.n;
  for (long i = lo + threadIdx.x; i < hi; i += 256) t.ema[i] = t.ema[i] * decay + (1.f - decay) * t.p[i];
}
