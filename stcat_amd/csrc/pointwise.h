// HBM-bound kernels of the hot path: LayerNorm (+residual) forward/backward, column
// reductions (bias / affine gradients), FrozenBN+ReLU backward masks, element-wise glue,
// sine embeddings, NHWC max-pool and the T x T temporal-map argmax.
// All are wave64 kernels with 16-byte accesses where the layout allows it.
#pragma once
#include "stcat_platform.h"
#include "stcat_rng.h"

// ---------------------------------------------------------------------------------
// LayerNorm over D = 256 (torch.nn.LayerNorm(256), eps 1e-5: modal_encoder.py:218-219,
// query_decoder.py:296-299, 573-576).  One wave per row, one float4 per lane.
// ---------------------------------------------------------------------------------
// y = LayerNorm(res + dropout(x)): the dropout of the residual branch (modal_encoder.py:237-240 and friends) is
// applied in-register from its counter-based mask; drop.thresh == 0 means no dropout
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* x, const float* res, const float* gamma,
                                                           const float* beta, float* y, float* mean, float* rstd,
                                                           int M, float eps, DropParams drop) {
  drop = stcat_drop_resolve(drop);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4 g = stcat_ld4(gamma + lane * 4), bt = stcat_ld4(beta + lane * 4);
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    float4 v = stcat_ld4(x + (long)row * 256 + lane * 4);
    if (drop.thresh) {
      const unsigned long long c0 = (unsigned long long)row * 256 + lane * 4;
      v.x *= stcat_drop_mul(drop, c0); v.y *= stcat_drop_mul(drop, c0 + 1);
      v.z *= stcat_drop_mul(drop, c0 + 2); v.w *= stcat_drop_mul(drop, c0 + 3);
    }
    if (res) {
      const float4 r = stcat_ld4(res + (long)row * 256 + lane * 4);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    const float mu = stcat_wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float var = stcat_wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rs = 1.f / sqrtf(var + eps);
    stcat_st4(y + (long)row * 256 + lane * 4,
              make_float4(dx * rs * g.x + bt.x, dy * rs * g.y + bt.y, dz * rs * g.z + bt.z, dw * rs * g.w + bt.w));
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += dy * xhat, dbeta += dy
// with dropout: dz is the gradient of the residual input, dx = mask * dz the gradient of the dropped branch
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* dy, const float* x, const float* res,
                                                           const float* gamma, const float* mean, const float* rstd,
                                                           float* dz, float* dx, float* dgamma, float* dbeta, int M,
                                                           DropParams drop) {
  drop = stcat_drop_resolve(drop);
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4 g = stcat_ld4(gamma + lane * 4);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    float4 v = stcat_ld4(x + (long)row * 256 + lane * 4);
    float4 dm = make_float4(1.f, 1.f, 1.f, 1.f);
    if (drop.thresh) {
      const unsigned long long c0 = (unsigned long long)row * 256 + lane * 4;
      dm = make_float4(stcat_drop_mul(drop, c0), stcat_drop_mul(drop, c0 + 1), stcat_drop_mul(drop, c0 + 2),
                       stcat_drop_mul(drop, c0 + 3));
      v.x *= dm.x; v.y *= dm.y; v.z *= dm.z; v.w *= dm.w;
    }
    if (res) {
      const float4 r = stcat_ld4(res + (long)row * 256 + lane * 4);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    const float4 d = stcat_ld4(dy + (long)row * 256 + lane * 4);
    const float mu = mean[row], rs = rstd[row];
    const float hx = (v.x - mu) * rs, hy = (v.y - mu) * rs, hz = (v.z - mu) * rs, hw = (v.w - mu) * rs;
    const float gx = d.x * g.x, gy = d.y * g.y, gz = d.z * g.z, gw = d.w * g.w;
    const float c1 = stcat_wave_sum(gx + gy + gz + gw) * (1.f / 256.f);
    const float c2 = stcat_wave_sum(gx * hx + gy * hy + gz * hz + gw * hw) * (1.f / 256.f);
    const float4 dzv = make_float4(rs * (gx - c1 - hx * c2), rs * (gy - c1 - hy * c2), rs * (gz - c1 - hz * c2),
                                   rs * (gw - c1 - hw * c2));
    stcat_st4(dz + (long)row * 256 + lane * 4, dzv);
    if (dx) stcat_st4(dx + (long)row * 256 + lane * 4, make_float4(dzv.x * dm.x, dzv.y * dm.y, dzv.z * dm.z, dzv.w * dm.w));
    ag.x += d.x * hx; ag.y += d.y * hy; ag.z += d.z * hz; ag.w += d.w * hw;
    ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
  }
  float* r0 = &red[0][w][lane * 4];
  float* r1 = &red[1][w][lane * 4];
  r0[0] = ag.x; r0[1] = ag.y; r0[2] = ag.z; r0[3] = ag.w;
  r1[0] = ab.x; r1[1] = ab.y; r1[2] = ab.z; r1[3] = ab.w;
  __syncthreads();
  const int c = threadIdx.x;
  atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
  atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
}

// out[n] += sum_m a[m][n] * (b ? b[m][n] : 1)      (bias gradients; affine gradients)
__global__ void __launch_bounds__(256) colsum_kernel(const float* a, const float* b, float* out, int M, int N,
                                                    int rows_per_block) {
  const int n = blockIdx.y * 256 + threadIdx.x;
  if (n >= N) return;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc = 0.f;
  for (int m = m0; m < m1; ++m) {
    const float v = a[(long)m * N + n];
    acc += b ? v * b[(long)m * N + n] : v;
  }
  atomicAdd(out + n, acc);
}

// ---------------------------------------------------------------------------------
// FrozenBN fold (backbone.py:56-66): scale = w * rsqrt(rv + eps), bias = b - rm * scale
// ---------------------------------------------------------------------------------
__global__ void frozen_bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                                      float* bias, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = w[c] * (1.f / sqrtf(rv[c] + eps));
    scale[c] = s;
    bias[c] = b[c] - rm[c] * s;
  }
}

// backward of y = relu(scale*conv + bias (+ res)):  dz = dy * [y > 0];  G = dz * scale[c];  dres = dz
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* dy, const float* y, const float* scale, float* G,
                                                     float* dres, long n4, int C, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 d = stcat_ld4(dy + i * 4);
    if (relu) {
      const float4 yy = stcat_ld4(y + i * 4);
      d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
      d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
    }
    if (dres) stcat_st4(dres + i * 4, d);
    if (G) {
      if (scale) {
        const float4 s = stcat_ld4(scale + (int)((i * 4) % C));
        d.x *= s.x; d.y *= s.y; d.z *= s.z; d.w *= s.w;
      }
      stcat_st4(G + i * 4, d);
    }
  }
}

// ---------------------------------------------------------------------------------
// element-wise glue.  b is indexed modulo bmod (row broadcast when bmod == D).
// ---------------------------------------------------------------------------------
enum { EW_ADD = 0, EW_MUL = 1, EW_SIGMOID = 2, EW_TANH = 3, EW_RELU = 4, EW_INVSIG = 5, EW_SIGMOID_BWD = 6,
       EW_TANH_BWD = 7, EW_INVSIG_BWD = 8, EW_ADD3 = 9, EW_AXPBY = 10, EW_COPY = 11 };

__global__ void __launch_bounds__(256) ew_kernel(int op, const float* a, const float* b, const float* c, float* out,
                                                long n, long bmod, float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = a[i];
    float r;
    switch (op) {
      case EW_ADD: r = x + b[i % bmod]; break;
      case EW_MUL: r = x * b[i % bmod]; break;
      case EW_ADD3: r = x + b[i] + c[i]; break;
      case EW_AXPBY: r = alpha * x + beta * b[i % bmod]; break;
      case EW_SIGMOID: r = 1.f / (1.f + expf(-x)); break;
      case EW_TANH: r = tanhf(x); break;
      case EW_RELU: r = fmaxf(x, 0.f); break;
      case EW_INVSIG: {  // models/net_utils.py:59-63
        const float xc = fminf(fmaxf(x, 0.f), 1.f);
        r = logf(fmaxf(xc, 1e-3f) / fmaxf(1.f - xc, 1e-3f));
      } break;
      case EW_SIGMOID_BWD: { const float y = b[i]; r = x * y * (1.f - y); } break;   // a = dy, b = y
      case EW_TANH_BWD: { const float y = b[i]; r = x * (1.f - y * y); } break;      // a = dy, b = y
      case EW_INVSIG_BWD: {                                                          // a = dy, b = x
        const float xx = b[i];
        float d = 0.f;
        if (xx >= 0.f && xx <= 1.f) {
          if (xx > 1e-3f) d += 1.f / xx;
          if (1.f - xx > 1e-3f) d += 1.f / (1.f - xx);
        }
        r = x * d;
      } break;
      default: r = x; break;
    }
    out[i] = r;
  }
}

// row-strided form of the two-operand ops (column blocks of wider matrices: a layer's slice of a layer-batched projection,
// the first half of the anchor sine embedding): out[r][c] = op(a[r][c], b[r][c]) with independent leading dimensions.
__global__ void __launch_bounds__(256) ew2d_kernel(int op, const float* a, long lda, const float* b, long ldb, float* out,
                                                  long ldo, long rows, int cols, float alpha, float beta) {
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    const float x = a[r * lda + c];
    float v;
    switch (op) {
      case EW_ADD: v = x + b[r * ldb + c]; break;
      case EW_MUL: v = x * b[r * ldb + c]; break;
      case EW_AXPBY: v = alpha * x + beta * b[r * ldb + c]; break;
      default: v = x; break;   // EW_COPY
    }
    out[r * ldo + c] = v;
  }
}

// ---------------------------------------------------------------------------------
// anchor sine embedding (net_utils.py:29-56): anchors [M][4] (x,y,w,h) -> [M][512] (y,x,w,h blocks)
// dimt: 128 divisors 10000^(2*floor(i/2)/128), precomputed in fp32 by the host.
// ---------------------------------------------------------------------------------
__global__ void sine_embed_fwd_kernel(const float* anchor, const float* dimt, float* out, int M) {
  const long n = (long)M * 512;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i >> 9), j = (int)(i & 511), blk = j >> 7, k = j & 127;
    const int coord = blk == 0 ? 1 : (blk == 1 ? 0 : blk);
    const float e = anchor[m * 4 + coord] * 6.283185307179586f / dimt[k];
    out[i] = (k & 1) ? cosf(e) : sinf(e);
  }
}
__global__ void sine_embed_bwd_kernel(const float* anchor, const float* dimt, const float* dout, float* danchor,
                                      int M) {
  // one wave per anchor row, lanes over the 512 outputs
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int m = blockIdx.x * 4 + w; m < M; m += gridDim.x * 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < 512; j += 64) {
      const int blk = j >> 7, k = j & 127;
      const int coord = blk == 0 ? 1 : (blk == 1 ? 0 : blk);
      const float sc = 6.283185307179586f / dimt[k];
      const float e = anchor[m * 4 + coord] * 6.283185307179586f / dimt[k];
      const float d = ((k & 1) ? -sinf(e) : cosf(e)) * sc * dout[(long)m * 512 + j];
      acc[0] += coord == 0 ? d : 0.f; acc[1] += coord == 1 ? d : 0.f;
      acc[2] += coord == 2 ? d : 0.f; acc[3] += coord == 3 ? d : 0.f;
    }
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      const float s = stcat_wave_sum(acc[c]);
      if (lane == 0) danchor[m * 4 + c] = s;
    }
  }
}

// 2-D sine position embedding (vision_model/position_encoding.py:70-94, normalize=True)
// mask [n][h][w] (1 = pad) -> pos [n][h*w][256] (token-major; channels = [pos_y(128) | pos_x(128)])
__global__ void pos_sine_2d_kernel(const unsigned char* mask, const float* dimt, float* pos, int n, int h, int w) {
  const long total = (long)n * h * w * 256;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i & 255);
    const long pix = i >> 8;
    const int xw = (int)(pix % w), yh = (int)((pix / w) % h), f = (int)(pix / ((long)w * h));
    const unsigned char* mk = mask + (long)f * h * w;
    float cum = 0.f, tot = 0.f;
    if (c < 128) {  // y: cumulative count of valid rows in this column
      for (int y = 0; y < h; ++y) {
        const float v = mk[y * w + xw] ? 0.f : 1.f;
        tot += v;
        if (y <= yh) cum += v;
      }
    } else {
      for (int x = 0; x < w; ++x) {
        const float v = mk[yh * w + x] ? 0.f : 1.f;
        tot += v;
        if (x <= xw) cum += v;
      }
    }
    const int k = c & 127;
    const float e = cum / (tot + 1e-6f) * 6.283185307179586f / dimt[k];
    pos[i] = (k & 1) ? cosf(e) : sinf(e);
  }
}

// 3x3 stride-2 pad-1 max-pool, NHWC (torchvision ResNet stem; backbone.py:115-119)
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float* x, float* y, int n, int H, int W, int C,
                                                          int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)n * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const long pix = i / c4n;
    const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH), f = (int)(pix / ((long)OW * OH));
    float4 m = make_float4(STCAT_NEG_INF, STCAT_NEG_INF, STCAT_NEG_INF, STCAT_NEG_INF);
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = oh * 2 - 1 + kh;
      if (hh < 0 || hh >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = ow * 2 - 1 + kw;
        if (ww < 0 || ww >= W) continue;
        const float4 v = stcat_ld4(x + (((long)f * H + hh) * W + ww) * C + c4 * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    stcat_st4(y + i * 4, m);
  }
}

// ---------------------------------------------------------------------------------
// live "2D temporal map" (models/post_processor.py:30-53): masked upper-triangular
// log_softmax(start)[s] + log_softmax(end)[e], flat argmax with first-max tie-break.
// One workgroup per video; sted [b][T][2]; out [b][2] = (start_idx, end_idx).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) temporal_map_argmax_kernel(const float* sted, const int* durations, int* out,
                                                                 int T) {
  __shared__ float ls[1024], le[1024];
  __shared__ float redv[256];
  __shared__ int redi[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* sp = sted + (long)b * T * 2;
  // log-softmax of both columns (256 threads cooperate)
  float m0 = STCAT_NEG_INF, m1 = STCAT_NEG_INF;
  for (int i = t; i < T; i += 256) { m0 = fmaxf(m0, sp[i * 2]); m1 = fmaxf(m1, sp[i * 2 + 1]); }
  redv[t] = m0; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) redv[t] = fmaxf(redv[t], redv[t + s]); __syncthreads(); }
  m0 = redv[0]; __syncthreads();
  redv[t] = m1; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) redv[t] = fmaxf(redv[t], redv[t + s]); __syncthreads(); }
  m1 = redv[0]; __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  for (int i = t; i < T; i += 256) { s0 += expf(sp[i * 2] - m0); s1 += expf(sp[i * 2 + 1] - m1); }
  redv[t] = s0; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) redv[t] += redv[t + s]; __syncthreads(); }
  s0 = redv[0]; __syncthreads();
  redv[t] = s1; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) redv[t] += redv[t + s]; __syncthreads(); }
  s1 = redv[0]; __syncthreads();
  const float l0 = logf(s0), l1 = logf(s1);
  for (int i = t; i < T; i += 256) { ls[i] = sp[i * 2] - m0 - l0; le[i] = sp[i * 2 + 1] - m1 - l1; }
  __syncthreads();
  const int dur = durations[b];
  float best = -1e32f;  // value of every masked cell; flat index 0 wins when all are masked
  int besti = 0;
  for (int s = t; s < T && s < dur; s += 256) {
    for (int e = s + 1; e < dur; ++e) {
      const float v = ls[s] + le[e];
      const int idx = s * T + e;
      if (v > best || (v == best && idx < besti)) { best = v; besti = idx; }
    }
  }
  redv[t] = best; redi[t] = besti; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      const float v = redv[t + s];
      const int ix = redi[t + s];
      if (v > redv[t] || (v == redv[t] && ix < redi[t])) { redv[t] = v; redi[t] = ix; }
    }
    __syncthreads();
  }
  if (t == 0) { out[b * 2] = redi[0] / T; out[b * 2 + 1] = redi[0] % T; }
}

// ---------------------------------------------------------------------------------
// narrow Linear layers (N <= 16): the 4/2/1-wide prediction heads and the anchor projection
// (pipeline.py:42-47 last MLP layers, query_decoder.py:448).  One wave per output row.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) small_linear_fwd_kernel(const float* x, const float* w, const float* bias,
                                                              float* y, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int m = blockIdx.x * 4 + wv; m < M; m += gridDim.x * 4) {
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      for (int k = lane * 4; k < K; k += 256) {
        const float4 a = stcat_ld4(x + (long)m * K + k), b = stcat_ld4(w + (long)n * K + k);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
      acc = stcat_wave_sum(acc);
      if (lane == 0) y[(long)m * N + n] = acc + (bias ? bias[n] : 0.f);
    }
  }
}
// dx[m][k] = sum_n g[m][n] w[n][k]
__global__ void __launch_bounds__(256) small_linear_dx_kernel(const float* g, const float* w, float* dx, int M, int N,
                                                             int K) {
  const long total = (long)M * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / K), k = (int)(i % K);
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += g[(long)m * N + n] * w[(long)n * K + k];
    dx[i] = acc;
  }
}
// dw[n][k] = sum_m g[m][n] x[m][k];  db[n] = sum_m g[m][n]   (plain stores).  One workgroup per (n, 32 columns of k):
// 8 row slices x 32 columns, reduced through LDS — the serial loop over the M = layers x T rows is 8x shorter than
// with one thread per output (the kernel is pure latency: 0.4 MFLOP).
__global__ void __launch_bounds__(256) small_linear_dw_kernel(const float* g, const float* x, float* dw, float* db,
                                                             int M, int N, int K) {
  __shared__ float part[2][8][32];
  const int t = threadIdx.x, kk = t & 31, ms = t >> 5;
  const int n = blockIdx.y, k = blockIdx.x * 32 + kk;
  float acc = 0.f, accb = 0.f;
  if (k < K) {
    for (int m = ms; m < M; m += 8) {
      const float gv = g[(long)m * N + n];
      acc += gv * x[(long)m * K + k];
      accb += gv;
    }
  }
  part[0][ms][kk] = acc;
  part[1][ms][kk] = accb;
  __syncthreads();
  if (ms == 0 && k < K) {
    float a = 0.f, b = 0.f;
    STCAT_UNROLL
    for (int i = 0; i < 8; ++i) { a += part[0][i][kk]; b += part[1][i][kk]; }
    dw[(long)n * K + k] = a;
    if (db && k == 0) db[n] = b;
  }
}

// ---------------------------------------------------------------------------------
// weight transpose for the data-gradient GEMM: W[Cout][taps][Cin] (OHWI / [N][K]) -> Wt[taps][Cin][Cout],
// so that dgrad's reduction index (tap, co) is contiguous and it can use the forward kernel's staging path.
// One 32x32 LDS tile per block; ~0.15 ms per step for all 83 M hot-path weights.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) weight_transpose_kernel(const float* w, float* wt, int Cout, int taps, int Cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long)co * taps + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wt[((long)tap * Cin + ci) * Cout + co] = tile[tx][r];
  }
}

// y = res + dropout(x)   (res may be null; the same launch on dY is the backward of the x branch)
__global__ void dropout_kernel(const float* x, const float* res, float* y, long n, DropParams d) {
  d = stcat_drop_resolve(d);
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 v = stcat_ld4(x + i);
      v.x *= stcat_drop_mul(d, (unsigned long long)i);
      v.y *= stcat_drop_mul(d, (unsigned long long)i + 1);
      v.z *= stcat_drop_mul(d, (unsigned long long)i + 2);
      v.w *= stcat_drop_mul(d, (unsigned long long)i + 3);
      if (res) {
        const float4 r = stcat_ld4(res + i);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      stcat_st4(y + i, v);
    } else {
      for (long j = i; j < n; ++j) y[j] = x[j] * stcat_drop_mul(d, (unsigned long long)j) + (res ? res[j] : 0.f);
    }
  }
}

// The same transpose for MANY weights in one launch (every conv of a backward pass): entry e owns blocks
// [blk0, blk0 + nbx*nby*taps) of the grid; a block finds its entry by binary search in the device table.
struct WtEntry {
  const float* w;
  float* wt;
  int Cout, taps, Cin;
  int blk0, nbx, nby;
};
__global__ void __launch_bounds__(256) weight_transpose_multi_kernel(const WtEntry* tab, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last entry with blk0 <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WtEntry e = tab[lo];
  const int rel = blockIdx.x - e.blk0;
  const int bx = rel % e.nbx, by = (rel / e.nbx) % e.nby, tap = rel / (e.nbx * e.nby);
  const int ci0 = bx * 32, co0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < e.Cout && ci < e.Cin) ? e.w[((long)co * e.taps + tap) * e.Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < e.Cin && co < e.Cout) e.wt[((long)tap * e.Cin + ci) * e.Cout + co] = tile[tx][r];
  }
}



// ---------------------------------------------------------------------------------------------------
// 2D temporal map head (models/map2d_head.py) — optional op (the reference never wires it into a loss); forward + backward.
// Gen2DMap (:9-62) = adaptive pooling of the T frame features to N steps, then 39 cascaded MaxPool1d layers written on
// sparse diagonals.  In closed form every valid cell (i, j) holds the RANGE MAXIMUM of the pooled sequence over [i, j]
// (verified equal to the cascade, tests/golden/map2d.npz), so the cascade becomes two small kernels.
// ---------------------------------------------------------------------------------------------------
// x [b][T][D] -> pooled [b][N][D]:  T > N: adaptive_avg_pool1d (then adaptive_max_pool1d N -> N = identity);
//                                   T <= N: adaptive_max_pool1d.  Window of step n: [floor(n T / N), ceil((n+1) T / N))
__global__ void __launch_bounds__(256) map2d_pool_kernel(const float* x, float* pooled, int b, int T, int N, int D) {
  const long total = (long)b * N * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D), n = (int)((i / D) % N), bb = (int)(i / ((long)D * N));
    const int s = (n * T) / N, e = ((n + 1) * T + N - 1) / N;
    const float* src = x + ((long)bb * T) * D + d;
    float acc = T > N ? 0.f : STCAT_NEG_INF;
    for (int t = s; t < e; ++t) {
      const float v = src[(long)t * D];
      acc = T > N ? acc + v : fmaxf(acc, v);
    }
    pooled[i] = T > N ? acc / (float)(e - s) : acc;
  }
}

// pooled [b][N][D] -> map NHWC [b][N][N][D]: cell c of the sparse list -> (ci[c], cj[c]); the rest of the map is zero
// (caller-zeroed).  One thread per (batch, cell, 4 channels).
__global__ void __launch_bounds__(256) map2d_cells_kernel(const float* pooled, const int* ci, const int* cj, int ncells,
                                                         float* map, int b, int N, int D) {
  const int d4n = D / 4;
  const long total = (long)b * ncells * d4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % d4n), c = (int)((i / d4n) % ncells), bb = (int)(i / ((long)d4n * ncells));
    const int i0 = ci[c], j0 = cj[c];
    float4 m = make_float4(STCAT_NEG_INF, STCAT_NEG_INF, STCAT_NEG_INF, STCAT_NEG_INF);
    for (int n = i0; n <= j0; ++n) {
      const float4 v = stcat_ld4(pooled + ((long)bb * N + n) * D + d4 * 4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    stcat_st4(map + (((long)bb * N + i0) * N + j0) * D + d4 * 4, m);
  }
}

// Backward of the two kernels above (TempPredictionHead in train mode returns raw scores for a loss, map2d_head.py:122-124).
// A cell's gradient goes to the FIRST maximum of its range, which is where the reference's cascade of max-pools routes it
// (every MaxPool1d backward picks the first maximum of its window; a cascade of them the leftmost of the range).
__global__ void __launch_bounds__(256) map2d_cells_bwd_kernel(const float* pooled, const int* ci, const int* cj, int ncells,
                                                             const float* dmap, float* dpooled, int b, int N, int D) {
  const long total = (long)b * ncells * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D), c = (int)((i / D) % ncells), bb = (int)(i / ((long)D * ncells));
    const int i0 = ci[c], j0 = cj[c];
    const float* src = pooled + ((long)bb * N) * D + d;
    float best = src[(long)i0 * D];
    int arg = i0;
    for (int n = i0 + 1; n <= j0; ++n) {
      const float v = src[(long)n * D];
      if (v > best) { best = v; arg = n; }
    }
    const float g = dmap[(((long)bb * N + i0) * N + j0) * D + d];
    atomicAdd(&dpooled[((long)bb * N + arg) * D + d], g);
  }
}
// dx (zeroed by the caller) += the gradient of pooled: T > N averages its window, T <= N takes the (first) maximum of it
__global__ void __launch_bounds__(256) map2d_pool_bwd_kernel(const float* x, const float* dpooled, float* dx, int b, int T,
                                                            int N, int D) {
  const long total = (long)b * N * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D), n = (int)((i / D) % N), bb = (int)(i / ((long)D * N));
    const int s = (n * T) / N, e = ((n + 1) * T + N - 1) / N;
    const float g = dpooled[i];
    float* dst = dx + ((long)bb * T) * D + d;
    if (T > N) {
      const float gi = g / (float)(e - s);
      for (int t = s; t < e; ++t) atomicAdd(&dst[(long)t * D], gi);
    } else {
      const float* src = x + ((long)bb * T) * D + d;
      float best = src[(long)s * D];
      int arg = s;
      for (int t = s + 1; t < e; ++t) {
        const float v = src[(long)t * D];
        if (v > best) { best = v; arg = t; }
      }
      atomicAdd(&dst[(long)arg * D], g);
    }
  }
}

// y[m][:] *= w[m % period]  — the mask-normalisation weight of TempConvInteraction (:245-249): one factor per map pixel
__global__ void __launch_bounds__(256) rowscale_kernel(float* y, const float* w, long rows, int C, int period) {
  const int c4n = C / 4;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / c4n;
    const float f = w[m % period];
    float4 v = stcat_ld4(y + i * 4);
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    stcat_st4(y + i * 4, v);
  }
}
