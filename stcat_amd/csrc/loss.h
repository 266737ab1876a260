// VideoSTGLoss (models/criterion.py:11-208) for ALL decoder layers in one launch, and its gradient in one more.
//
// The loss works on the decoders' tiny outputs — boxes [layers][rows][4], start/end logits [layers][b][T][2], the time
// decoder's head-mean attention [layers][b][T][T], actioness logits [layers][b][T] — and is ~100 element-wise /
// reduction ops per evaluation when written with tensor ops (and ~130 more in its backward).  Everything derived
// from the annotations alone (GT rows, Gaussian span targets, masks: LossPlan in pipeline.py) comes in precomputed.
// One workgroup per decoder layer; vec[k][l] = the un-weighted loss k of layer l (k: bbox L1, GIoU, span KL, guided
// attention, actioness BCE).  HBM-latency-bound by construction (~0.6 MB of operands).
#pragma once
#include "stcat_platform.h"

struct StgLossParams {
  const float* boxes;                // [nl][rows_total][4]  (cx, cy, w, h) after the sigmoid
  const long* rows;                  // [nbox] rows of the GT span (criterion.py:168-171)
  const float* tgt;                  // [nbox][4] target boxes (cx, cy, w, h)
  const float* sted;                 // [nl][b][T][2]
  const float* dist;                 // [b][T][2] normalised Gaussian span targets (criterion.py:98-111)
  const unsigned char* time_mask;    // [b][T] 1 = frame exists
  const float* w;                    // [nl][b][T][T]
  const unsigned char* pos_or_pad;   // [b][T] 1 = inside the GT span or padding
  const float* nb_neg;               // [b]
  const float* act;                  // [nl][b][T] or NULL
  const float* act_tgt;              // [b][T]
  const float* act_w;                // [b][T]
  const float* num_boxes_dev;        // device scalar (data-parallel: the all-reduced box count) or NULL
  float num_boxes;
  int nl, rows_total, nbox, b, T;
  const float* wmat;                 // [5][nl] loss weights (weight_dict) or NULL
  float* vec;                        // [5][nl]
  float* total;                      // caller-zeroed scalar: += sum_k,l wmat * vec, or NULL
  // backward
  const float* gvec;                 // [5][nl] or NULL
  const float* gtotal;               // device scalar or NULL
  float* d_boxes; float* d_sted; float* d_w; float* d_act;
};

static __device__ __forceinline__ float stcat_block_sum(float v, float* red) {
  v = stcat_wave_sum(v);
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
static __device__ __forceinline__ float stcat_block_max(float v, float* red) {
  v = stcat_wave_max(v);
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
  return s;
}

// generalized IoU of one matched pair (utils/box_utils.py:90-113) and, when G != nullptr, d giou / d (cx, cy, w, h)
// of the prediction with torch's sub-gradient conventions (minimum / maximum: ties split in half; clamp(min=0): the
// gradient passes at x >= 0).
static __device__ __forceinline__ float stcat_giou_pair(const float4 p, const float4 t, float* G) {
  const float ax0 = p.x - 0.5f * p.z, ay0 = p.y - 0.5f * p.w, ax1 = p.x + 0.5f * p.z, ay1 = p.y + 0.5f * p.w;
  const float bx0 = t.x - 0.5f * t.z, by0 = t.y - 0.5f * t.w, bx1 = t.x + 0.5f * t.z, by1 = t.y + 0.5f * t.w;
  const float aw = ax1 - ax0, ah = ay1 - ay0;
  const float area_a = aw * ah, area_b = (bx1 - bx0) * (by1 - by0);
  const float iwr = fminf(ax1, bx1) - fmaxf(ax0, bx0), ihr = fminf(ay1, by1) - fmaxf(ay0, by0);
  const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
  const float inter = iw * ih;
  const float uni = area_a + area_b - inter;
  const float cwr = fmaxf(ax1, bx1) - fminf(ax0, bx0), chr_ = fmaxf(ay1, by1) - fminf(ay0, by0);
  const float cw = fmaxf(cwr, 0.f), ch = fmaxf(chr_, 0.f);
  const float hull = cw * ch;
  const float giou = inter / uni - (hull - uni) / hull;
  if (G) {
    const float dU = -inter / (uni * uni) + 1.f / hull;        // d giou / d union
    const float gI = 1.f / uni - dU;                            // d giou / d inter (union = areas - inter)
    const float gH = -uni / (hull * hull);
    const float g_iw = (iwr >= 0.f ? gI * ih : 0.f), g_ih = (ihr >= 0.f ? gI * iw : 0.f);
    const float g_cw = (cwr >= 0.f ? gH * ch : 0.f), g_ch = (chr_ >= 0.f ? gH * cw : 0.f);
#define STCAT_LT(a, b) ((a) < (b) ? 1.f : ((a) == (b) ? 0.5f : 0.f))
    // x
    const float g_ax1 = g_iw * STCAT_LT(ax1, bx1) + g_cw * STCAT_LT(bx1, ax1) + dU * ah;
    const float g_ax0 = -g_iw * STCAT_LT(bx0, ax0) - g_cw * STCAT_LT(ax0, bx0) - dU * ah;
    const float g_ay1 = g_ih * STCAT_LT(ay1, by1) + g_ch * STCAT_LT(by1, ay1) + dU * aw;
    const float g_ay0 = -g_ih * STCAT_LT(by0, ay0) - g_ch * STCAT_LT(ay0, by0) - dU * aw;
#undef STCAT_LT
    G[0] = g_ax0 + g_ax1;
    G[1] = g_ay0 + g_ay1;
    G[2] = 0.5f * (g_ax1 - g_ax0);
    G[3] = 0.5f * (g_ay1 - g_ay0);
  }
  return giou;
}

__global__ void __launch_bounds__(256) stg_loss_fwd_kernel(StgLossParams p) {
  __shared__ float red[8];
  const int l = blockIdx.x, tid = threadIdx.x;
  const int T = p.T, b = p.b;
  const float nb = p.num_boxes_dev ? *p.num_boxes_dev : p.num_boxes;
  float out[5];
  // ---- boxes: L1 + GIoU on the GT-span rows (criterion.py:38-66)
  {
    float l1 = 0.f, gl = 0.f;
    for (int i = tid; i < p.nbox; i += blockDim.x) {
      const float4 q = stcat_ld4(p.boxes + ((long)l * p.rows_total + p.rows[i]) * 4);
      const float4 t = stcat_ld4(p.tgt + (long)i * 4);
      l1 += fabsf(q.x - t.x) + fabsf(q.y - t.y) + fabsf(q.z - t.z) + fabsf(q.w - t.w);
      gl += 1.f - stcat_giou_pair(q, t, nullptr);
    }
    out[0] = stcat_block_sum(l1, red) / nb;
    out[1] = stcat_block_sum(gl, red) / nb;
  }
  // ---- start / end distributions: KL(softmax over time || Gaussian target) (criterion.py:68-124)
  {
    float tot = 0.f;
    for (int bc = 0; bc < b * 2; ++bc) {
      const int bi = bc >> 1, c = bc & 1;
      const float* x = p.sted + ((long)(l * b + bi) * T) * 2 + c;
      const unsigned char* tm = p.time_mask + (long)bi * T;
      float m = STCAT_NEG_INF;
      for (int t = tid; t < T; t += blockDim.x) if (tm[t]) m = fmaxf(m, x[t * 2]);
      m = stcat_block_max(m, red);
      float s = 0.f;
      for (int t = tid; t < T; t += blockDim.x) if (tm[t]) s += expf(x[t * 2] - m);
      s = stcat_block_sum(s, red);
      float kl = 0.f;
      for (int t = tid; t < T; t += blockDim.x)
        if (tm[t]) {
          const float pr = expf(x[t * 2] - m) / s;
          kl += pr * logf((pr + 1e-6f) / p.dist[((long)bi * T + t) * 2 + c]);
        }
      tot += stcat_block_sum(kl, red);
    }
    out[2] = tot / (float)(b * T);
  }
  // ---- guided attention: -log(1 - w) on the query rows outside the GT span (criterion.py:126-145)
  {
    float tot = 0.f;
    for (int bi = 0; bi < b; ++bi) {
      const float* w = p.w + (long)(l * b + bi) * T * T;
      const unsigned char* pp = p.pos_or_pad + (long)bi * T;
      float a = 0.f;
      for (int i = tid; i < T * T; i += blockDim.x)
        if (!pp[i / T]) a += -logf(1.f - w[i] + 1e-6f);
      tot += stcat_block_sum(a, red) / p.nb_neg[bi];
    }
    out[3] = tot / (float)b;
  }
  // ---- actioness: weighted BCE with logits (criterion.py:147-158)
  out[4] = 0.f;
  if (p.act) {
    float a = 0.f;
    for (int i = tid; i < b * T; i += blockDim.x)
      if (p.time_mask[i]) {
        const float x = p.act[(long)l * b * T + i], y = p.act_tgt[i];
        a += p.act_w[i] * (fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
      }
    out[4] = stcat_block_sum(a, red) / (float)(b * T);
  }
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < 5; ++k) {
      p.vec[k * p.nl + l] = out[k];
      if (p.wmat) tot += p.wmat[k * p.nl + l] * out[k];
    }
    if (p.total && p.wmat) atomicAdd(p.total, tot);
  }
}

__global__ void __launch_bounds__(256) stg_loss_bwd_kernel(StgLossParams p) {
  __shared__ float red[8];
  const int l = blockIdx.x, tid = threadIdx.x;
  const int T = p.T, b = p.b;
  const float nb = p.num_boxes_dev ? *p.num_boxes_dev : p.num_boxes;
  float g[5];
  for (int k = 0; k < 5; ++k) {
    g[k] = p.gvec ? p.gvec[k * p.nl + l] : 0.f;
    if (p.gtotal && p.wmat) g[k] += *p.gtotal * p.wmat[k * p.nl + l];
  }
  // ---- boxes
  {
    float* d = p.d_boxes + (long)l * p.rows_total * 4;
    for (int i = tid; i < p.rows_total * 4; i += blockDim.x) d[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < p.nbox; i += blockDim.x) {
      const long r = p.rows[i];
      const float4 q = stcat_ld4(p.boxes + ((long)l * p.rows_total + r) * 4);
      const float4 t = stcat_ld4(p.tgt + (long)i * 4);
      float G[4];
      stcat_giou_pair(q, t, G);
      const float qa[4] = {q.x, q.y, q.z, q.w}, ta[4] = {t.x, t.y, t.z, t.w};
      for (int c = 0; c < 4; ++c) {
        const float df = qa[c] - ta[c];
        const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        d[r * 4 + c] = (g[0] * sg - g[1] * G[c]) / nb;
      }
    }
  }
  // ---- start / end distributions
  for (int bc = 0; bc < b * 2; ++bc) {
    const int bi = bc >> 1, c = bc & 1;
    const float* x = p.sted + ((long)(l * b + bi) * T) * 2 + c;
    float* dx = p.d_sted + ((long)(l * b + bi) * T) * 2 + c;
    const unsigned char* tm = p.time_mask + (long)bi * T;
    const float sc = g[2] / (float)(b * T);
    float m = STCAT_NEG_INF;
    for (int t = tid; t < T; t += blockDim.x) if (tm[t]) m = fmaxf(m, x[t * 2]);
    m = stcat_block_max(m, red);
    float s = 0.f;
    for (int t = tid; t < T; t += blockDim.x) if (tm[t]) s += expf(x[t * 2] - m);
    s = stcat_block_sum(s, red);
    float dot = 0.f;
    for (int t = tid; t < T; t += blockDim.x)
      if (tm[t]) {
        const float pr = expf(x[t * 2] - m) / s;
        const float gp = sc * (logf((pr + 1e-6f) / p.dist[((long)bi * T + t) * 2 + c]) + pr / (pr + 1e-6f));
        dot += pr * gp;
      }
    dot = stcat_block_sum(dot, red);
    for (int t = tid; t < T; t += blockDim.x) {
      float v = 0.f;
      if (tm[t]) {
        const float pr = expf(x[t * 2] - m) / s;
        const float gp = sc * (logf((pr + 1e-6f) / p.dist[((long)bi * T + t) * 2 + c]) + pr / (pr + 1e-6f));
        v = pr * (gp - dot);
      }
      dx[t * 2] = v;
    }
  }
  // ---- guided attention
  for (int bi = 0; bi < b; ++bi) {
    const float* w = p.w + (long)(l * b + bi) * T * T;
    float* dw = p.d_w + (long)(l * b + bi) * T * T;
    const unsigned char* pp = p.pos_or_pad + (long)bi * T;
    const float sc = g[3] / ((float)b * p.nb_neg[bi]);
    for (int i = tid; i < T * T; i += blockDim.x) dw[i] = pp[i / T] ? 0.f : sc / (1.f - w[i] + 1e-6f);
  }
  // ---- actioness
  if (p.act) {
    const float sc = g[4] / (float)(b * T);
    for (int i = tid; i < b * T; i += blockDim.x) {
      float v = 0.f;
      if (p.time_mask[i]) {
        const float x = p.act[(long)l * b * T + i];
        v = sc * p.act_w[i] * (1.f / (1.f + expf(-x)) - p.act_tgt[i]);
      }
      p.d_act[(long)l * b * T + i] = v;
    }
  }
}
