// Attention kernels (fp32 MFMA 32x32x2, head dim 32).
//
// (1) mha_self_*  — joint self-attention over S <= 256 tokens per (frame, head): the
//     encoder's spatial layers (S = H*W + L + 1, batch = T frames; modal_encoder.py:161-168,
//     228-242), its temporal layers (S = T+1, batch 1; :180-185) and both decoders'
//     temporal self-attention over T queries (query_decoder.py:341, 604-610).
//     One workgroup per (batch, head), one wave per 32-query tile.  K and V of the
//     (batch, head) live wholly in LDS (<= 67 KB), so softmax is single-pass.  The score
//     tile is computed TRANSPOSED (S^T = K Q^T): with the 32x32 MFMA C-layout every lane
//     then owns one query column and 16 keys per tile, so max / sum over keys are in-lane
//     reductions plus one lane^32 exchange, and the same registers are directly the A
//     operand of the P·V MFMA (no LDS round trip, no permutes).
//     Probabilities are kept for backward as Pt[b][h][key][query] (padded to Sp = 32*NT):
//     with 288 GB of HBM, storing P (<= 103 MB / layer) is cheaper than recomputing it on
//     the 64-cycle fp32 matrix pipe.
// (2) attn_q1_*   — the decoders' time-aligned cross-attention: exactly ONE query per
//     frame against that frame's S' memory tokens (query_decoder.py:386-417, 618-639;
//     custom MHA attention.py:184-393 with k-dim 64 = [content | position], v-dim 32).
//     This is GEMV-class and HBM-bound; one wave per (frame, head), lanes over keys.
#pragma once
#include "stcat_platform.h"
#include "stcat_rng.h"

struct AttnParams {
  const float* Q;  // [B][S][ldq], head h at column h*32
  const float* K;
  const float* V;
  float* O;          // [B][S][ldo]
  float* Pt;         // [B][H][Sp][Sp] probabilities, key-major; null = do not keep them (inference)
  const unsigned char* kpm;  // [B][S], 1 = padded key (-> -inf), or null
  int B, H, S;
  int ldq, ldk, ldv, ldo;
  float scale;
  DropParams drop;   // dropout on the probabilities (attention.py:381 / nn.MultiheadAttention); Pt keeps the
                     // UNdropped softmax, every consumer regenerates the mask from counter (bh*Sp + key)*Sp + query
  float* Lse;        // [B][H][Sp][2] (row maximum, 1 / row sum) for the recomputing backward (round 5), or null
};


template <int NT>
__global__ void __launch_bounds__(64 * NT) mha_self_fwd_kernel(AttnParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SP = NT * 32, KLD = 33;
  __shared__ __attribute__((aligned(16))) float Ks[SP * KLD];
  __shared__ __attribute__((aligned(16))) float Vs[SP * 32];
  __shared__ float kb[SP];
  const int t = threadIdx.x, lane = t & 63, qt = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const float* Kg = p.K + (long)b * p.S * p.ldk + h * 32;
  const float* Vg = p.V + (long)b * p.S * p.ldv + h * 32;
  for (int i = t; i < SP * 8; i += 64 * NT) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < p.S) {
      kv = stcat_ld4(Kg + (long)row * p.ldk + c4);
      vv = stcat_ld4(Vg + (long)row * p.ldv + c4);
    }
    float* kd = &Ks[row * KLD + c4];
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    stcat_st4(&Vs[row * 32 + c4], vv);
  }
  for (int i = t; i < SP; i += 64 * NT)
    kb[i] = (i < p.S && !(p.kpm && p.kpm[(long)b * p.S + i])) ? 0.f : STCAT_NEG_INF;
  // this lane's query row, dims hi*16 .. hi*16+15, pre-scaled (attention.py:283-285)
  const int q = qt * 32 + l31;
  float qr[16];
  {
    const float* Qg = p.Q + ((long)b * p.S + q) * p.ldq + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 v4 = q < p.S ? stcat_ld4(Qg + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qr[c * 4 + 0] = v4.x * p.scale; qr[c * 4 + 1] = v4.y * p.scale;
      qr[c * 4 + 2] = v4.z * p.scale; qr[c * 4 + 3] = v4.w * p.scale;
    }
  }
  __syncthreads();
  f32x16 sc[NT];
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s)
      sc[kt] = STCAT_MFMA_32x32x2(Ks[(kt * 32 + l31) * KLD + hi * 16 + s], qr[s], sc[kt]);
  }
  float mx = STCAT_NEG_INF;
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      sc[kt][r] += kb[key];
      mx = fmaxf(mx, sc[kt][r]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      sc[kt][r] = __expf(sc[kt][r] - mx);
      sum += sc[kt][r];
    }
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
  if (p.Lse && hi == 0) {    // (rows past S: finite values that make the recomputed probability zero)
    float* l2 = p.Lse + ((long)blockIdx.x * SP + q) * 2;
    l2[0] = q < p.S ? mx : 0.f;
    l2[1] = q < p.S ? inv : 0.f;
  }
  float* Ptg = p.Pt + (long)blockIdx.x * SP * SP;
  f32x16 o;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = sc[kt][r] * inv;
      if (p.Pt) Ptg[(long)key * SP + q] = pr;  // the stash exists for backward / head-mean weights only
      const float pd = pr * stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      o = STCAT_MFMA_32x32x2(pd, Vs[key * 32 + l31], o);
    }
  }
  float* Og = p.O + (long)b * p.S * p.ldo + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (qq < p.S) Og[(long)qq * p.ldo] = o[r];
  }
}

struct AttnBwdParams {
  const float* Q;
  const float* K;
  const float* V;
  const float* dO;   // [B][S][ldo]
  const float* Pt;   // [B][H][Sp][Sp]
  const float* dW;   // [B][S][S] gradient of the head-averaged weights, or null
  const float* O;    // [B][S][ldo] forward output (delta_q = dO . O)
  float* corr;       // [B][H][S] scratch: (1/H) sum_k P dW, only used with dW
  float* dSt;        // [B][H][Sp][Sp] scratch: scale * dS, key-major
  float* dQ;         // [B][S][ldg]
  float* dK;
  float* dV;
  int B, H, S;
  int ldq, ldk, ldv, ldo, ldg, ldgv;  // ldg: row stride of dQ and dK, ldgv: of dV
  float scale;
  DropParams drop;
  const float* Lse;          // recomputing form (round 5): [B][H][Sp][2] from the forward; Pt / dSt unused
  const unsigned char* kpm;  // recomputing form: the forward's key padding mask
};

// backward, phase 1: one wave per query tile -> dS (stored key-major, pre-multiplied by scale) and dQ
template <int NT>
__global__ void __launch_bounds__(64 * NT) mha_self_bwd_dq_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SP = NT * 32, KLD = 33;
  __shared__ __attribute__((aligned(16))) float Vs[SP * KLD];  // A operand of dP^T = V dO^T
  __shared__ __attribute__((aligned(16))) float Ks[SP * 32];   // B operand of dQ = dS K
  const int t = threadIdx.x, lane = t & 63, qt = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const float* Kg = p.K + (long)b * p.S * p.ldk + h * 32;
  const float* Vg = p.V + (long)b * p.S * p.ldv + h * 32;
  for (int i = t; i < SP * 8; i += 64 * NT) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < p.S) {
      kv = stcat_ld4(Kg + (long)row * p.ldk + c4);
      vv = stcat_ld4(Vg + (long)row * p.ldv + c4);
    }
    float* vd = &Vs[row * KLD + c4];
    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
    stcat_st4(&Ks[row * 32 + c4], kv);
  }
  const int q = qt * 32 + l31;
  float dor[16];
  float delta = 0.f;  // delta_q = sum_k P[q,k] dP[q,k] = dO[q,:] . O[q,:]  (+ head-mean-weights term)
  {
    const float* g = p.dO + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    const float* og = p.O + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f), o4 = v4;
      if (q < p.S) {
        v4 = stcat_ld4(g + c * 4);
        o4 = stcat_ld4(og + c * 4);
      }
      dor[c * 4 + 0] = v4.x; dor[c * 4 + 1] = v4.y; dor[c * 4 + 2] = v4.z; dor[c * 4 + 3] = v4.w;
      delta += v4.x * o4.x + v4.y * o4.y + v4.z * o4.z + v4.w * o4.w;
    }
  }
  delta += __shfl_xor(delta, 32);
  if (p.dW && q < p.S) delta += p.corr[(long)blockIdx.x * p.S + q];
  __syncthreads();
  const float* Ptg = p.Pt + (long)blockIdx.x * SP * SP + hi * 4 * SP + q;
  float* dStg = p.dSt + (long)blockIdx.x * SP * SP + hi * 4 * SP + q;
  const float invH = 1.f / (float)p.H;
  f32x16 dq;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 dp;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s)
      dp = STCAT_MFMA_32x32x2(Vs[(kt * 32 + l31) * KLD + hi * 16 + s], dor[s], dp);
    float pr_[16];  // the 16 stashed probabilities of this tile: issued together, ahead of the dependent MFMA chain
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) pr_[r] = Ptg[(long)(kt * 32 + (r & 3) + 8 * (r >> 2)) * SP];
    float dw_[16];  // likewise the head-mean-weights gradient (time decoder only)
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key_ = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      dw_[r] = (p.dW && q < p.S && key_ < p.S) ? p.dW[((long)b * p.S + q) * p.S + key_] * invH : 0.f;
    }
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int krel = kt * 32 + (r & 3) + 8 * (r >> 2);  // + 4*hi folded into the base pointers
      const int key = krel + 4 * hi;
      float dpv = dp[r];
      dpv += dw_[r];
      dpv *= stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);  // dP = M' * dP'
      const float ds = pr_[r] * (dpv - delta) * p.scale;
      dStg[(long)krel * SP] = ds;
      dq = STCAT_MFMA_32x32x2(ds, Ks[key * 32 + l31], dq);
    }
  }
  float* g = p.dQ + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (qq < p.S) g[(long)qq * p.ldg] = dq[r];
  }
}

// backward, phase 2: one wave per key tile -> dV = P^T dO, dK = (scale*dS)^T Q
template <int NT>
__global__ void __launch_bounds__(64 * NT) mha_self_bwd_dkv_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SP = NT * 32;
  __shared__ __attribute__((aligned(16))) float dOs[SP * 32];
  __shared__ __attribute__((aligned(16))) float Qs[SP * 32];
  const int t = threadIdx.x, lane = t & 63, kt = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const float* Qg = p.Q + (long)b * p.S * p.ldq + h * 32;
  const float* Gg = p.dO + (long)b * p.S * p.ldo + h * 32;
  for (int i = t; i < SP * 8; i += 64 * NT) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
    if (row < p.S) {
      qv = stcat_ld4(Qg + (long)row * p.ldq + c4);
      gv = stcat_ld4(Gg + (long)row * p.ldo + c4);
    }
    stcat_st4(&Qs[row * 32 + c4], qv);
    stcat_st4(&dOs[row * 32 + c4], gv);
  }
  __syncthreads();
  const int key = kt * 32 + l31;  // A-operand row of this lane
  const float* Prow = p.Pt + (long)blockIdx.x * SP * SP + (long)key * SP + hi * 4;
  const float* Srow = p.dSt + (long)blockIdx.x * SP * SP + (long)key * SP + hi * 4;
  f32x16 dv, dk;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  // reduction over queries in chunks of 8: half `hi` feeds queries qc*8 + hi*4 + e
  for (int qc = 0; qc < SP / 8; ++qc) {
    float4 pv = stcat_ld4(Prow + qc * 8);
    const float4 sv = stcat_ld4(Srow + qc * 8);
    const int qb = qc * 8 + hi * 4;
    if (p.drop.thresh) {  // dV = P'^T dO with P' = M' * P
      const unsigned long long c0 = ((unsigned long long)blockIdx.x * SP + key) * SP + qb;
      pv.x *= stcat_drop_mul(p.drop, c0); pv.y *= stcat_drop_mul(p.drop, c0 + 1);
      pv.z *= stcat_drop_mul(p.drop, c0 + 2); pv.w *= stcat_drop_mul(p.drop, c0 + 3);
    }
    dv = STCAT_MFMA_32x32x2(pv.x, dOs[(qb + 0) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.x, Qs[(qb + 0) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.y, dOs[(qb + 1) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.y, Qs[(qb + 1) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.z, dOs[(qb + 2) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.z, Qs[(qb + 2) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.w, dOs[(qb + 3) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.w, Qs[(qb + 3) * 32 + l31], dk);
  }
  float* gv = p.dV + (long)b * p.S * p.ldgv + h * 32 + l31;
  float* gk = p.dK + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (kk < p.S) {
      gv[(long)kk * p.ldgv] = dv[r];
      gk[(long)kk * p.ldg] = dk[r];
    }
  }
}

// ---------------------------------------------------------------------------------
// Recomputing backward (round 5): the forward keeps (row maximum, 1 / row sum) per query instead of the S x S probability
// matrix, and both backward kernels rebuild their probability tiles with the forward's own instruction sequence
// (K q^T on the fp32 pipe, exp(s + mask - max) / sum).  Per layer at C3 this removes the 103 MB probability stash the
// forward wrote and the 103 + 103 MB (Pt, dSt) each backward kernel moved — 515 MB of the attention's HBM traffic — for
// 112 (dQ) and 224 (dK / dV) more fp32 MFMAs per wave.  Used when nobody reads the head-mean weights and S <= 256.
// ---------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * NT) mha_self_bwd_dq_rc_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SP = NT * 32, KLD = 33;
  STCAT_DYN_SHARED(float, sm);
  float* Ks = sm;                 // [SP][33]: column reads (A operand of K q^T) and row reads (B operand of dS K)
  float* Vs = Ks + SP * KLD;      // [SP][33]: A operand of dP^T = V dO^T
  float* kb = Vs + SP * KLD;
  const int t = threadIdx.x, lane = t & 63, qt = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const float* Kg = p.K + (long)b * p.S * p.ldk + h * 32;
  const float* Vg = p.V + (long)b * p.S * p.ldv + h * 32;
  for (int i = t; i < SP * 8; i += 64 * NT) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < p.S) {
      kv = stcat_ld4(Kg + (long)row * p.ldk + c4);
      vv = stcat_ld4(Vg + (long)row * p.ldv + c4);
    }
    float* kd = &Ks[row * KLD + c4];
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    float* vd = &Vs[row * KLD + c4];
    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
  }
  for (int i = t; i < SP; i += 64 * NT)
    kb[i] = (i < p.S && !(p.kpm && p.kpm[(long)b * p.S + i])) ? 0.f : STCAT_NEG_INF;
  const int q = qt * 32 + l31;
  float qr[16], dor[16];
  float delta = 0.f;
  {
    const float* Qg = p.Q + ((long)b * p.S + q) * p.ldq + h * 32 + hi * 16;
    const float* g = p.dO + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    const float* og = p.O + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = q4, o4 = q4;
      if (q < p.S) {
        q4 = stcat_ld4(Qg + c * 4);
        v4 = stcat_ld4(g + c * 4);
        o4 = stcat_ld4(og + c * 4);
      }
      qr[c * 4 + 0] = q4.x * p.scale; qr[c * 4 + 1] = q4.y * p.scale; qr[c * 4 + 2] = q4.z * p.scale; qr[c * 4 + 3] = q4.w * p.scale;
      dor[c * 4 + 0] = v4.x; dor[c * 4 + 1] = v4.y; dor[c * 4 + 2] = v4.z; dor[c * 4 + 3] = v4.w;
      delta += v4.x * o4.x + v4.y * o4.y + v4.z * o4.z + v4.w * o4.w;
    }
  }
  delta += __shfl_xor(delta, 32);
  const float* l2 = p.Lse + ((long)blockIdx.x * SP + q) * 2;
  const float mx = l2[0], inv = l2[1];
  __syncthreads();
  f32x16 dq;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  STCAT_UNROLL
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 sc, dp;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s) {      // (two independent accumulators: the chains interleave)
      sc = STCAT_MFMA_32x32x2(Ks[(kt * 32 + l31) * KLD + hi * 16 + s], qr[s], sc);
      dp = STCAT_MFMA_32x32x2(Vs[(kt * 32 + l31) * KLD + hi * 16 + s], dor[s], dp);
    }
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = __expf(sc[r] + kb[key] - mx) * inv;
      const float dpv = dp[r] * stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      const float ds = pr * (dpv - delta) * p.scale;
      dq = STCAT_MFMA_32x32x2(ds, Ks[key * KLD + l31], dq);
    }
  }
  float* g = p.dQ + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (qq < p.S) g[(long)qq * p.ldg] = dq[r];
  }
}

// one wave per key tile: for every block of 32 queries rebuild P^T and dP^T tiles with the KEY as the lane (A = the query rows
// of Q * scale resp. dO from LDS, B = this lane's own K resp. V row): accumulator register r of such a tile holds query
// (r & 3) + 8 (r >> 2) + 4 hi of the block — exactly the pairing of A value and k index the next MFMA (dV += P'^T dO,
// dK += dS^T Q) wants, so the tiles never leave the registers
template <int NT>
__global__ void __launch_bounds__(64 * NT) mha_self_bwd_dkv_rc_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SP = NT * 32, KLD = 33;
  STCAT_DYN_SHARED(float, sm);
  float* Qs = sm;                  // [SP][33], PRE-SCALED (q * scale, as the forward's query registers)
  float* dOs = Qs + SP * KLD;      // [SP][33]
  float* mxs = dOs + SP * KLD;     // per query: row maximum, 1 / row sum, delta = dO . O
  float* ivs = mxs + SP;
  float* dls = ivs + SP;
  const int t = threadIdx.x, lane = t & 63, kt = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const float* Qg = p.Q + (long)b * p.S * p.ldq + h * 32;
  const float* Gg = p.dO + (long)b * p.S * p.ldo + h * 32;
  const float* Og = p.O + (long)b * p.S * p.ldo + h * 32;
  for (int i = t; i < SP * 8; i += 64 * NT) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv, ov = qv;
    if (row < p.S) {
      qv = stcat_ld4(Qg + (long)row * p.ldq + c4);
      gv = stcat_ld4(Gg + (long)row * p.ldo + c4);
      ov = stcat_ld4(Og + (long)row * p.ldo + c4);
    }
    float* qd = &Qs[row * KLD + c4];
    qd[0] = qv.x * p.scale; qd[1] = qv.y * p.scale; qd[2] = qv.z * p.scale; qd[3] = qv.w * p.scale;
    float* gd = &dOs[row * KLD + c4];
    gd[0] = gv.x; gd[1] = gv.y; gd[2] = gv.z; gd[3] = gv.w;
    float d = gv.x * ov.x + gv.y * ov.y + gv.z * ov.z + gv.w * ov.w;     // the 8 lanes of a row sit next to each other
    d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
    if ((i & 7) == 0) dls[row] = d;
  }
  for (int i = t; i < SP; i += 64 * NT) {
    const float* l2 = p.Lse + ((long)blockIdx.x * SP + i) * 2;
    mxs[i] = l2[0];
    ivs[i] = l2[1];
  }
  // this lane's key row: dims hi*16 .. hi*16+15 of K and V, and its padding bias
  const int key = kt * 32 + l31;
  float kr[16], vr[16];
  {
    const float* Kg = p.K + ((long)b * p.S + key) * p.ldk + h * 32 + hi * 16;
    const float* Vg = p.V + ((long)b * p.S + key) * p.ldv + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
      if (key < p.S) {
        k4 = stcat_ld4(Kg + c * 4);
        v4 = stcat_ld4(Vg + c * 4);
      }
      kr[c * 4 + 0] = k4.x; kr[c * 4 + 1] = k4.y; kr[c * 4 + 2] = k4.z; kr[c * 4 + 3] = k4.w;
      vr[c * 4 + 0] = v4.x; vr[c * 4 + 1] = v4.y; vr[c * 4 + 2] = v4.z; vr[c * 4 + 3] = v4.w;
    }
  }
  const float kbv = (key < p.S && !(p.kpm && p.kpm[(long)b * p.S + key])) ? 0.f : STCAT_NEG_INF;
  __syncthreads();
  f32x16 dv, dk;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  for (int qb = 0; qb < NT; ++qb) {
    f32x16 st, dpt;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s) {
      st = STCAT_MFMA_32x32x2(Qs[(qb * 32 + l31) * KLD + hi * 16 + s], kr[s], st);
      dpt = STCAT_MFMA_32x32x2(dOs[(qb * 32 + l31) * KLD + hi * 16 + s], vr[s], dpt);
    }
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int q = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = __expf(st[r] + kbv - mxs[q]) * ivs[q];
      const float m = stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      const float ds = pr * (dpt[r] * m - dls[q]);           // (the scale rides in Qs: dK = dS^T (q * scale))
      dv = STCAT_MFMA_32x32x2(pr * m, dOs[q * KLD + l31], dv);
      dk = STCAT_MFMA_32x32x2(ds, Qs[q * KLD + l31], dk);
    }
  }
  float* gv = p.dV + (long)b * p.S * p.ldgv + h * 32 + l31;
  float* gk = p.dK + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (kk < p.S) {
      gv[(long)kk * p.ldgv] = dv[r];
      gk[(long)kk * p.ldg] = dk[r];
    }
  }
}

// ---------------------------------------------------------------------------------
// The same three kernels for 256 < S <= 512 tokens per frame: non-square clips.  The reference's transforms resize the
// short side to 448 with max_size 720 (datasets/build.py:20-44): a 16:9 video becomes 405 x 720 -> 13 x 23 visual tokens
// + text + [CLS] = 310..340 rows (modal_encoder.py:161-168).  K / V (resp. Q / dO) of the (frame, head) still fit the
// CU's LDS (<= 135 KB, dynamic); a workgroup is 8 waves = 8 query (key) tiles and grid.y walks the tile groups.  The
// forward keeps softmax exact without 16 score tiles in registers by running the key tiles twice: pass 1 = running
// max / sum (online), pass 2 = recompute the scores, normalise, stash Pt, accumulate P V.
// ---------------------------------------------------------------------------------
static __device__ __forceinline__ void stcat_attn_stage2(const float* Ag, int lda, const float* Bg, int ldb, float* As,
                                                         int a_ld, float* Bs, int S, int SP, int t, int nthreads) {
  // As [SP][a_ld] (a_ld = 33: conflict-free column reads; 32: row-major B operand), Bs [SP][32]; rows >= S are zero
  for (int i = t; i < SP * 8; i += nthreads) {
    const int row = i >> 3, c4 = (i & 7) * 4;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
    if (row < S) {
      av = stcat_ld4(Ag + (long)row * lda + c4);
      bv = stcat_ld4(Bg + (long)row * ldb + c4);
    }
    float* ad = &As[row * a_ld + c4];
    ad[0] = av.x; ad[1] = av.y; ad[2] = av.z; ad[3] = av.w;
    stcat_st4(&Bs[row * 32 + c4], bv);
  }
}

__global__ void __launch_bounds__(512) mha_self_fwd_long_kernel(AttnParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int KLD = 33;
  const int NT = (p.S + 31) >> 5, SP = NT * 32;
  STCAT_DYN_SHARED(float, sm);
  float* Ks = sm;
  float* Vs = Ks + SP * KLD;
  float* kb = Vs + SP * 32;
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int qt = blockIdx.y * 8 + (t >> 6);
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  stcat_attn_stage2(p.K + (long)b * p.S * p.ldk + h * 32, p.ldk, p.V + (long)b * p.S * p.ldv + h * 32, p.ldv, Ks, KLD, Vs,
                    p.S, SP, t, 512);
  for (int i = t; i < SP; i += 512)
    kb[i] = (i < p.S && !(p.kpm && p.kpm[(long)b * p.S + i])) ? 0.f : STCAT_NEG_INF;
  const int q = qt * 32 + l31;
  float qr[16];
  {
    const float* Qg = p.Q + ((long)b * p.S + q) * p.ldq + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 v4 = q < p.S ? stcat_ld4(Qg + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qr[c * 4 + 0] = v4.x * p.scale; qr[c * 4 + 1] = v4.y * p.scale;
      qr[c * 4 + 2] = v4.z * p.scale; qr[c * 4 + 3] = v4.w * p.scale;
    }
  }
  __syncthreads();
  if (qt >= NT) return;
  // pass 1: this lane's 16 keys of every tile -> running (max, sum); the lane pair (l31, l31 + 32) is merged afterwards
  float mx = STCAT_NEG_INF, sum = 0.f;
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 sc;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s) sc = STCAT_MFMA_32x32x2(Ks[(kt * 32 + l31) * KLD + hi * 16 + s], qr[s], sc);
    float tm = STCAT_NEG_INF;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      sc[r] += kb[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
      tm = fmaxf(tm, sc[r]);
    }
    const float mn = fmaxf(mx, tm);
    if (mn > STCAT_NEG_INF) {
      float ts = 0.f;
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) ts += __expf(sc[r] - mn);
      sum = sum * __expf(mx - mn) + ts;
      mx = mn;
    }
  }
  {
    const float mo = __shfl_xor(mx, 32), so = __shfl_xor(sum, 32);
    const float mn = fmaxf(mx, mo);
    sum = (mx > STCAT_NEG_INF ? sum * __expf(mx - mn) : 0.f) + (mo > STCAT_NEG_INF ? so * __expf(mo - mn) : 0.f);
    mx = mn;
  }
  const float inv = 1.f / sum;
  float* Ptg = p.Pt + (long)blockIdx.x * SP * SP;
  f32x16 o;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 sc;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s) sc = STCAT_MFMA_32x32x2(Ks[(kt * 32 + l31) * KLD + hi * 16 + s], qr[s], sc);
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = __expf(sc[r] + kb[key] - mx) * inv;
      if (p.Pt) Ptg[(long)key * SP + q] = pr;
      const float pd = pr * stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      o = STCAT_MFMA_32x32x2(pd, Vs[key * 32 + l31], o);
    }
  }
  float* Og = p.O + (long)b * p.S * p.ldo + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (qq < p.S) Og[(long)qq * p.ldo] = o[r];
  }
}

__global__ void __launch_bounds__(512) mha_self_bwd_dq_long_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int KLD = 33;
  const int NT = (p.S + 31) >> 5, SP = NT * 32;
  STCAT_DYN_SHARED(float, sm);
  float* Vs = sm;              // [SP][33]: A operand of dP^T = V dO^T
  float* Ks = Vs + SP * KLD;   // [SP][32]: B operand of dQ = dS K
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int qt = blockIdx.y * 8 + (t >> 6);
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  stcat_attn_stage2(p.V + (long)b * p.S * p.ldv + h * 32, p.ldv, p.K + (long)b * p.S * p.ldk + h * 32, p.ldk, Vs, KLD, Ks,
                    p.S, SP, t, 512);
  const int q = qt * 32 + l31;
  float dor[16];
  float delta = 0.f;
  {
    const float* g = p.dO + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    const float* og = p.O + ((long)b * p.S + q) * p.ldo + h * 32 + hi * 16;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f), o4 = v4;
      if (q < p.S) {
        v4 = stcat_ld4(g + c * 4);
        o4 = stcat_ld4(og + c * 4);
      }
      dor[c * 4 + 0] = v4.x; dor[c * 4 + 1] = v4.y; dor[c * 4 + 2] = v4.z; dor[c * 4 + 3] = v4.w;
      delta += v4.x * o4.x + v4.y * o4.y + v4.z * o4.z + v4.w * o4.w;
    }
  }
  delta += __shfl_xor(delta, 32);
  if (p.dW && q < p.S) delta += p.corr[(long)blockIdx.x * p.S + q];
  __syncthreads();
  if (qt >= NT) return;
  const float* Ptg = p.Pt + (long)blockIdx.x * SP * SP + (long)hi * 4 * SP + q;
  float* dStg = p.dSt + (long)blockIdx.x * SP * SP + (long)hi * 4 * SP + q;
  const float invH = 1.f / (float)p.H;
  f32x16 dq;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 dp;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    STCAT_UNROLL
    for (int s = 0; s < 16; ++s) dp = STCAT_MFMA_32x32x2(Vs[(kt * 32 + l31) * KLD + hi * 16 + s], dor[s], dp);
    float pr_[16];
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) pr_[r] = Ptg[(long)(kt * 32 + (r & 3) + 8 * (r >> 2)) * SP];
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int krel = kt * 32 + (r & 3) + 8 * (r >> 2);
      const int key = krel + 4 * hi;
      float dpv = dp[r];
      if (p.dW && q < p.S && key < p.S) dpv += p.dW[((long)b * p.S + q) * p.S + key] * invH;
      dpv *= stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      const float ds = pr_[r] * (dpv - delta) * p.scale;
      dStg[(long)krel * SP] = ds;
      dq = STCAT_MFMA_32x32x2(ds, Ks[key * 32 + l31], dq);
    }
  }
  float* g = p.dQ + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (qq < p.S) g[(long)qq * p.ldg] = dq[r];
  }
}

__global__ void __launch_bounds__(512) mha_self_bwd_dkv_long_kernel(AttnBwdParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  const int NT = (p.S + 31) >> 5, SP = NT * 32;
  STCAT_DYN_SHARED(float, sm);
  float* dOs = sm;             // [SP][32]
  float* Qs = dOs + SP * 32;   // [SP][32]
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int kt = blockIdx.y * 8 + (t >> 6);
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  stcat_attn_stage2(p.dO + (long)b * p.S * p.ldo + h * 32, p.ldo, p.Q + (long)b * p.S * p.ldq + h * 32, p.ldq, dOs, 32, Qs,
                    p.S, SP, t, 512);
  __syncthreads();
  if (kt >= NT) return;
  const int key = kt * 32 + l31;
  const float* Prow = p.Pt + (long)blockIdx.x * SP * SP + (long)key * SP + hi * 4;
  const float* Srow = p.dSt + (long)blockIdx.x * SP * SP + (long)key * SP + hi * 4;
  f32x16 dv, dk;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  for (int qc = 0; qc < SP / 8; ++qc) {
    float4 pv = stcat_ld4(Prow + qc * 8);
    const float4 sv = stcat_ld4(Srow + qc * 8);
    const int qb = qc * 8 + hi * 4;
    if (p.drop.thresh) {
      const unsigned long long c0 = ((unsigned long long)blockIdx.x * SP + key) * SP + qb;
      pv.x *= stcat_drop_mul(p.drop, c0); pv.y *= stcat_drop_mul(p.drop, c0 + 1);
      pv.z *= stcat_drop_mul(p.drop, c0 + 2); pv.w *= stcat_drop_mul(p.drop, c0 + 3);
    }
    dv = STCAT_MFMA_32x32x2(pv.x, dOs[(qb + 0) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.x, Qs[(qb + 0) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.y, dOs[(qb + 1) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.y, Qs[(qb + 1) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.z, dOs[(qb + 2) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.z, Qs[(qb + 2) * 32 + l31], dk);
    dv = STCAT_MFMA_32x32x2(pv.w, dOs[(qb + 3) * 32 + l31], dv);
    dk = STCAT_MFMA_32x32x2(sv.w, Qs[(qb + 3) * 32 + l31], dk);
  }
  float* gv = p.dV + (long)b * p.S * p.ldgv + h * 32 + l31;
  float* gk = p.dK + (long)b * p.S * p.ldg + h * 32 + l31;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) {
    const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (kk < p.S) {
      gv[(long)kk * p.ldgv] = dv[r];
      gk[(long)kk * p.ldg] = dk[r];
    }
  }
}

// head-averaged attention weights W[b][q][k] = mean_h P[b][h][q][k]  (nn.MultiheadAttention
// need_weights=True; consumed only for the time decoder: pipeline.py:84-85)
__global__ void attn_weights_mean_kernel(const float* Pt, float* W, int B, int H, int S, int SP, DropParams drop) {
  drop = stcat_drop_resolve(drop);
  const long n = (long)B * S * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % S), q = (int)((i / S) % S), b = (int)(i / ((long)S * S));
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
      const long c = (((long)b * H + h) * SP + k) * SP + q;
      acc += Pt[c] * stcat_drop_mul(drop, (unsigned long long)c);  // train mode returns the DROPPED weights
    }
    W[i] = acc / (float)H;
  }
}

// corr[b][h][q] = (1/H) sum_k P[b][h][q][k] * dW[b][q][k]   (softmax-backward delta term of the
// head-averaged weights gradient; only the time decoder's self-attention has one)
__global__ void __launch_bounds__(256) attn_dw_corr_kernel(const float* Pt, const float* dW, float* corr, int B, int H,
                                                           int S, int SP, DropParams drop) {
  // one workgroup per (b, h): 64 queries x 4 key slices per pass, slices reduced through LDS (the one-thread-per-
  // output form walked the S keys serially: 34 us for 0.03 MFLOP)
  drop = stcat_drop_resolve(drop);
  __shared__ float part[4][64];
  const int t = threadIdx.x, ql = t & 63, ks = t >> 6;
  const int bh = blockIdx.x, b = bh / H;
  for (int q0 = 0; q0 < S; q0 += 64) {
    const int q = q0 + ql;
    float acc = 0.f;
    if (q < S) {
      for (int k = ks; k < S; k += 4) {
        const long c = ((long)bh * SP + k) * SP + q;
        acc += Pt[c] * stcat_drop_mul(drop, (unsigned long long)c) * dW[((long)b * S + q) * S + k];
      }
    }
    part[ks][ql] = acc;
    __syncthreads();
    if (ks == 0 && q < S) corr[(long)bh * S + q] = (part[0][ql] + part[1][ql] + part[2][ql] + part[3][ql]) / (float)H;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// one query per frame
// ---------------------------------------------------------------------------------
struct AttnQ1Params {
  const float* q1;  // [B][ldq] (head h at h*32)
  const float* q2;  // second 32-dim part per head (sine/position part) or null
  const float* k1;  // [B][S][ldk]
  const float* k2;  // or null
  const float* v;   // [B][S][ldv]
  const unsigned char* kpm;  // [B][S]
  float* out;       // [B][H*32]
  float* P;         // [B][H][S]
  // backward
  const float* dout;
  float* dq1;
  float* dq2;
  float* dk1;  // [B][S][H*32]
  float* dk2;
  float* dv;
  int B, H, S;
  int ldq, ldk, ldv;
  float scale;
  DropParams drop;  // counter = bh * S + s; P keeps the undropped softmax
};

#define STCAT_Q1_MAXC 4  // key chunks of 256: S <= 1024

// One workgroup per (frame, head); its four waves take 64 keys each per 256-key chunk, so 4x as many waves are in flight
// as with one wave per (frame, head) — the kernel is a latency-bound gather of 128-byte key / value rows
// (3 x 13.5 MB per layer at C3), and memory-level parallelism is what it needs.  NC = number of 256-key chunks: the
// square benchmark clips (S' = 206) are NC = 1; a 405 x 720 clip of the reference's transforms (datasets/build.py:20-44:
// 13 x 23 + text tokens = 310..340 keys) is NC = 2.  A lane keeps its NC scores in registers: softmax stays single-pass.
template <int NC>
__global__ void __launch_bounds__(256) attn_q1_fwd_kernel(AttnQ1Params p) {
  p.drop = stcat_drop_resolve(p.drop);
  __shared__ float ps[NC * 256];
  __shared__ float red[2][4];
  __shared__ float opart[4][32];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  float qa[32], qb[32];
  STCAT_UNROLL
  for (int c = 0; c < 8; ++c) {
    float4 v4 = stcat_ld4(p.q1 + (long)b * p.ldq + h * 32 + c * 4);
    qa[c * 4] = v4.x; qa[c * 4 + 1] = v4.y; qa[c * 4 + 2] = v4.z; qa[c * 4 + 3] = v4.w;
    float4 u4 = p.q2 ? stcat_ld4(p.q2 + (long)b * p.ldq + h * 32 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    qb[c * 4] = u4.x; qb[c * 4 + 1] = u4.y; qb[c * 4 + 2] = u4.z; qb[c * 4 + 3] = u4.w;
  }
  float val[NC];
  float mloc = STCAT_NEG_INF;
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s = ch * 256 + w * 64 + lane;
    val[ch] = STCAT_NEG_INF;
    if (s < p.S && !(p.kpm && p.kpm[(long)b * p.S + s])) {
      const float* kr = p.k1 + ((long)b * p.S + s) * p.ldk + h * 32;
      float dot = 0.f;
      STCAT_UNROLL
      for (int d4 = 0; d4 < 8; ++d4) {
        float4 kv = stcat_ld4(kr + d4 * 4);
        dot += kv.x * qa[d4 * 4] + kv.y * qa[d4 * 4 + 1] + kv.z * qa[d4 * 4 + 2] + kv.w * qa[d4 * 4 + 3];
      }
      if (p.k2) {
        const float* kr2 = p.k2 + ((long)b * p.S + s) * p.ldk + h * 32;
        STCAT_UNROLL
        for (int d4 = 0; d4 < 8; ++d4) {
          float4 kv = stcat_ld4(kr2 + d4 * 4);
          dot += kv.x * qb[d4 * 4] + kv.y * qb[d4 * 4 + 1] + kv.z * qb[d4 * 4 + 2] + kv.w * qb[d4 * 4 + 3];
        }
      }
      val[ch] = dot * p.scale;
    }
    mloc = fmaxf(mloc, val[ch]);
  }
  const float mw = stcat_wave_max(mloc);
  if (lane == 0) red[0][w] = mw;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  float esum = 0.f;
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    val[ch] = __expf(val[ch] - mx);
    esum += val[ch];
  }
  const float sw = stcat_wave_sum(esum);
  if (lane == 0) red[1][w] = sw;
  __syncthreads();
  const float inv = 1.f / (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s = ch * 256 + w * 64 + lane;
    const float pr = val[ch] * inv;
    ps[s] = pr * stcat_drop_mul(p.drop, (unsigned long long)bh * p.S + s);
    if (s < p.S) p.P[(long)bh * p.S + s] = pr;
  }
  __syncthreads();
  float o = 0.f;
  const float* vb = p.v + (long)b * p.S * p.ldv + h * 32 + l31;
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s0 = ch * 256 + w * 64, s_end = min(p.S, s0 + 64);
    for (int k = s0 + hi; k < s_end; k += 2) o += ps[k] * vb[(long)k * p.ldv];
  }
  o += __shfl_xor(o, 32);
  if (hi == 0) opart[w][l31] = o;
  __syncthreads();
  if (t < 32) p.out[(long)b * p.H * 32 + h * 32 + t] = opart[0][t] + opart[1][t] + opart[2][t] + opart[3][t];
}

template <int NC>
__global__ void __launch_bounds__(256) attn_q1_bwd_kernel(AttnQ1Params p) {
  p.drop = stcat_drop_resolve(p.drop);
  __shared__ float dss[NC * 256];
  __shared__ float red[4];
  __shared__ float qpart[2][4][32];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int HD = p.H * 32;
  float go[32], qa[32], qb[32];
  STCAT_UNROLL
  for (int c = 0; c < 8; ++c) {
    float4 g4 = stcat_ld4(p.dout + (long)b * HD + h * 32 + c * 4);
    go[c * 4] = g4.x; go[c * 4 + 1] = g4.y; go[c * 4 + 2] = g4.z; go[c * 4 + 3] = g4.w;
    float4 v4 = stcat_ld4(p.q1 + (long)b * p.ldq + h * 32 + c * 4);
    qa[c * 4] = v4.x; qa[c * 4 + 1] = v4.y; qa[c * 4 + 2] = v4.z; qa[c * 4 + 3] = v4.w;
    float4 u4 = p.q2 ? stcat_ld4(p.q2 + (long)b * p.ldq + h * 32 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    qb[c * 4] = u4.x; qb[c * 4 + 1] = u4.y; qb[c * 4 + 2] = u4.z; qb[c * 4 + 3] = u4.w;
  }
  float pr[NC], dp[NC], dm[NC];
  float dloc = 0.f;
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s = ch * 256 + w * 64 + lane;
    pr[ch] = 0.f; dp[ch] = 0.f; dm[ch] = 1.f;
    if (s < p.S) {
      pr[ch] = p.P[(long)bh * p.S + s];
      const float* vr = p.v + ((long)b * p.S + s) * p.ldv + h * 32;
      float dot = 0.f;
      STCAT_UNROLL
      for (int d4 = 0; d4 < 8; ++d4) {
        float4 vv = stcat_ld4(vr + d4 * 4);
        dot += vv.x * go[d4 * 4] + vv.y * go[d4 * 4 + 1] + vv.z * go[d4 * 4 + 2] + vv.w * go[d4 * 4 + 3];
      }
      dm[ch] = stcat_drop_mul(p.drop, (unsigned long long)bh * p.S + s);
      dp[ch] = dot * dm[ch];  // dP = M' * (dO . V)
    }
    dloc += pr[ch] * dp[ch];
  }
  const float dw = stcat_wave_sum(dloc);
  if (lane == 0) red[w] = dw;
  __syncthreads();
  const float delta = red[0] + red[1] + red[2] + red[3];
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s = ch * 256 + w * 64 + lane;
    const float ds = pr[ch] * (dp[ch] - delta) * p.scale;
    const float pd = pr[ch] * dm[ch];  // dV = P' dO
    dss[s] = ds;
    if (s < p.S) {
      float* gv = p.dv + ((long)b * p.S + s) * HD + h * 32;
      float* gk1 = p.dk1 + ((long)b * p.S + s) * HD + h * 32;
      STCAT_UNROLL
      for (int d4 = 0; d4 < 8; ++d4) {
        stcat_st4(gv + d4 * 4, make_float4(pd * go[d4 * 4], pd * go[d4 * 4 + 1], pd * go[d4 * 4 + 2], pd * go[d4 * 4 + 3]));
        stcat_st4(gk1 + d4 * 4, make_float4(ds * qa[d4 * 4], ds * qa[d4 * 4 + 1], ds * qa[d4 * 4 + 2], ds * qa[d4 * 4 + 3]));
      }
      if (p.dk2) {
        float* gk2 = p.dk2 + ((long)b * p.S + s) * HD + h * 32;
        STCAT_UNROLL
        for (int d4 = 0; d4 < 8; ++d4)
          stcat_st4(gk2 + d4 * 4, make_float4(ds * qb[d4 * 4], ds * qb[d4 * 4 + 1], ds * qb[d4 * 4 + 2], ds * qb[d4 * 4 + 3]));
      }
    }
  }
  __syncthreads();
  float a1 = 0.f, a2 = 0.f;
  const float* k1b = p.k1 + (long)b * p.S * p.ldk + h * 32 + l31;
  const float* k2b = p.k2 ? p.k2 + (long)b * p.S * p.ldk + h * 32 + l31 : nullptr;
  STCAT_UNROLL
  for (int ch = 0; ch < NC; ++ch) {
    const int s0 = ch * 256 + w * 64, s_end = min(p.S, s0 + 64);
    for (int k = s0 + hi; k < s_end; k += 2) {
      const float d = dss[k];
      a1 += d * k1b[(long)k * p.ldk];
      if (k2b) a2 += d * k2b[(long)k * p.ldk];
    }
  }
  a1 += __shfl_xor(a1, 32);
  a2 += __shfl_xor(a2, 32);
  if (hi == 0) { qpart[0][w][l31] = a1; qpart[1][w][l31] = a2; }
  __syncthreads();
  if (t < 32) {
    p.dq1[(long)b * HD + h * 32 + t] = qpart[0][0][t] + qpart[0][1][t] + qpart[0][2][t] + qpart[0][3][t];
    if (p.dq2) p.dq2[(long)b * HD + h * 32 + t] = qpart[1][0][t] + qpart[1][1][t] + qpart[1][2][t] + qpart[1][3][t];
  }
}
