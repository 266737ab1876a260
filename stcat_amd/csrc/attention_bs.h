// Self-attention on the bf16 matrix pipe (round 2): softmax(scale Q K^T + key padding) V per (batch, head), head dim 32,
// for the encoder's spatial / temporal layers (modal_encoder.py:161-168, 180-185, 228-242) and every other call of
// nn.MultiheadAttention on the path that does not return its weights.
//
// Products are the bf16x3 split contraction of the GEMM family (x = hi + lo in bf16; hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate): d_h = 32 is two k-steps of the 32x32x16 tile.  What changed against
// the fp32-MFMA kernels of attention.h:
//   * online softmax over chunks of 128 keys: at most four score tiles are live per wave, any S works (the S <= 256
//     limit of the fp32 kernels excluded non-square clips: 13 x 23 + text + cls = 310 tokens);
//   * the backward pass RECOMPUTES the probabilities from Q, K and the row log-sum-exp (8 B per query instead of the
//     103 MB probability stash + 103 MB dS scratch per spatial layer at C3), in one kernel: pass A (lane = query)
//     gives dQ, pass B (lane = key) gives dK and dV — no atomics, no second launch;
//   * no operand is transposed in memory: K / V / Q / dO are staged once as row-major bf16 hi/lo planes (64-byte rows,
//     16-byte chunks XOR-ed with (row >> 2) & 3) and read either as 16-byte row fragments (k = head dim) or through
//     ds_read_b64_tr_b16 (k = token).  The probabilities never leave registers: the scores are computed transposed
//     (lane = query column or key column), so the accumulator registers of one tile ARE the B operand of the next
//     MFMA once the k-slots of the A operand are read in the matching (permuted) token order.
#pragma once
#include "attention.h"

struct AttnBsParams {
  const float* Q; const float* K; const float* V;   // [B][S][ld*], head h at column h*32
  float* O;                                         // [B][S][ldo]
  float* lse;                                       // [B][H][S] row log-sum-exp of the scaled, masked scores
  const unsigned char* kpm;                         // [B][S] 1 = padded key, or null
  // backward only
  const float* dO; float* dQ; float* dK; float* dV;
  int B, H, S;
  int ldq, ldk, ldv, ldo, ldg, ldgv;
  float scale;
  DropParams drop;   // dropout on the probabilities; counter of (b, h, key, query) = ((bh*SP + key)*SP + query), SP = 32*ceil(S/32)
};

#define STCAT_ABS_ROWB 64  // bytes per staged row: 32 bf16

// byte offset of 16-byte chunk c (0..3) of staged row r
static __device__ __forceinline__ unsigned stcat_abs_off(int r, int c) {
  return (unsigned)(r * STCAT_ABS_ROWB + ((c ^ ((r >> 2) & 3)) << 4));
}

// stage rows [r0, r0 + ROWS) of a [S][ld] fp32 matrix (32 columns of head h) as swizzled hi / lo planes; rows >= S are zero
template <int ROWS>
static __device__ __forceinline__ void stcat_abs_stage(const float* g, int ld, int r0, int S, char* ph, char* pl, int t,
                                                       int nthr, float mul) {
  for (int i = t; i < ROWS * 8; i += nthr) {
    const int r = i >> 3, c4 = i & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < S) v = stcat_ld4(g + (long)(r0 + r) * ld + c4 * 4);
    const float x[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
    bf16x4 h4, l4;
    STCAT_UNROLL
    for (int e = 0; e < 4; ++e) {
      const __bf16 hh = (__bf16)x[e];
      h4[e] = hh;
      l4[e] = (__bf16)(x[e] - (float)hh);
    }
    const unsigned off = stcat_abs_off(r, c4 >> 1) + (c4 & 1) * 8;
    *reinterpret_cast<bf16x4*>(ph + off) = h4;
    *reinterpret_cast<bf16x4*>(pl + off) = l4;
  }
}

// 16-byte row fragment: row r, head-dim chunk c
#define STCAT_ABS_ROWFRAG(PLANE, r, c) (*reinterpret_cast<const bf16x8*>((PLANE) + stcat_abs_off((r), (c))))

// transposed fragment for a contraction over tokens: lane (i = lane & 31 = head dim, hi) gets the 8 tokens
// t0 + 4 hi + {0,1,2,3, 8,9,10,11} of column i — the token order in which a 32x32 accumulator tile holds its rows
static __device__ __forceinline__ bf16x8 stcat_abs_trfrag(const char* plane, int t0, int lane) {
  const int pl = lane & 15, gq = (lane >> 4) & 1, hi = lane >> 5;
  const int r = t0 + 4 * hi + (pl >> 2);            // (r >> 2) is the same for the 4 rows of one read: t0 % 16 == 0
  const int colb = gq * 32 + (pl & 3) * 8;          // byte column inside the 64-byte row
  const unsigned o0 = (unsigned)(r * STCAT_ABS_ROWB + ((((colb >> 4) ^ ((r >> 2) & 3)) << 4) | (colb & 15)));
  const int r1 = r + 8;
  const unsigned o1 = (unsigned)(r1 * STCAT_ABS_ROWB + ((((colb >> 4) ^ ((r1 >> 2) & 3)) << 4) | (colb & 15)));
  const bf16x4 a = stcat_lds_tr4(reinterpret_cast<const __bf16*>(plane + o0));
  const bf16x4 b = stcat_lds_tr4(reinterpret_cast<const __bf16*>(plane + o1));
  bf16x8 f;
  STCAT_UNROLL
  for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
  return f;
}

// registers 8 j .. 8 j + 7 of an accumulator tile -> bf16 hi / lo B-operand fragments
static __device__ __forceinline__ void stcat_abs_split_regs(const f32x16& x, int j, bf16x8& h, bf16x8& l) {
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) {
    const float v = x[8 * j + e];
    const __bf16 hh = (__bf16)v;
    h[e] = hh;
    l[e] = (__bf16)(v - (float)hh);
  }
}

#define STCAT_ABS_MMA3(ACC, AH, AL, BH, BL)                 \
  ACC = STCAT_MFMA_BF16_32x32x16(AL, BH, ACC);              \
  ACC = STCAT_MFMA_BF16_32x32x16(AH, BL, ACC);              \
  ACC = STCAT_MFMA_BF16_32x32x16(AH, BH, ACC);

// ---- NP-plane forms (round 6).  NP = 2: the three-product split above (modes bf16x3 / bf16x3p / bf16x6).  NP = 3: x = p0 + p1 + p2
// EXACTLY (every residual of the split is exact in fp32) and a product keeps the six cross terms down to 2^-16 relative —
// p0 q0 + p0 q1 + p1 q0 + p1 q1 + p0 q2 + p2 q0 — the arithmetic of the plane GEMMs of mode bf16x6p (igemm_pl.h), issued
// smallest terms first.  Mode bf16x6p's self-attention ran on the fp32 matrix pipe (attention.h: 64-cycle MFMAs, the S x S
// probabilities and their gradient stashed in HBM: 397 MB read per backward launch at C3); with NP = 3 it runs here.
template <int NP> struct AbsFrag { bf16x8 p[NP]; };

template <int NP>
static __device__ __forceinline__ void stcat_abs_split8(const float* x, AbsFrag<NP>& f) {
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) {
    float r = x[e];
    STCAT_UNROLL
    for (int q = 0; q < NP; ++q) {
      const __bf16 b = (__bf16)r;
      f.p[q][e] = b;
      r -= (float)b;
    }
  }
}

// registers 8 j .. 8 j + 7 of an accumulator tile -> NP B-operand fragments
template <int NP>
static __device__ __forceinline__ void stcat_abs_split_acc(const f32x16& x, int j, AbsFrag<NP>& f) {
  float v[8];
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) v[e] = x[8 * j + e];
  stcat_abs_split8<NP>(v, f);
}

template <int NP>
static __device__ __forceinline__ void stcat_abs_mma(f32x16& acc, const AbsFrag<NP>& a, const AbsFrag<NP>& b) {
  if constexpr (NP == 3) {
    acc = STCAT_MFMA_BF16_32x32x16(a.p[2], b.p[0], acc);
    acc = STCAT_MFMA_BF16_32x32x16(a.p[0], b.p[2], acc);
    acc = STCAT_MFMA_BF16_32x32x16(a.p[1], b.p[1], acc);
  }
  acc = STCAT_MFMA_BF16_32x32x16(a.p[1], b.p[0], acc);
  acc = STCAT_MFMA_BF16_32x32x16(a.p[0], b.p[1], acc);
  acc = STCAT_MFMA_BF16_32x32x16(a.p[0], b.p[0], acc);
}

// stage rows [r0, r0 + ROWS) of a [S][ld] fp32 matrix (32 columns of one head) as NP swizzled planes at base + q * plane_bytes
template <int ROWS, int NP>
static __device__ __forceinline__ void stcat_abs_stage_n(const float* g, int ld, int r0, int S, char* base, int plane_bytes,
                                                         int t, int nthr, float mul) {
  for (int i = t; i < ROWS * 8; i += nthr) {
    const int r = i >> 3, c4 = i & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < S) v = stcat_ld4(g + (long)(r0 + r) * ld + c4 * 4);
    float x[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
    const unsigned off = stcat_abs_off(r, c4 >> 1) + (c4 & 1) * 8;
    STCAT_UNROLL
    for (int q = 0; q < NP; ++q) {
      bf16x4 w;
      STCAT_UNROLL
      for (int e = 0; e < 4; ++e) {
        const __bf16 b = (__bf16)x[e];
        w[e] = b;
        x[e] -= (float)b;
      }
      *reinterpret_cast<bf16x4*>(base + q * plane_bytes + off) = w;
    }
  }
}

template <int NP>
static __device__ __forceinline__ void stcat_abs_rowfrag_n(const char* base, int plane_bytes, int r, int c, AbsFrag<NP>& f) {
  STCAT_UNROLL
  for (int q = 0; q < NP; ++q) f.p[q] = STCAT_ABS_ROWFRAG(base + q * plane_bytes, r, c);
}

template <int NP>
static __device__ __forceinline__ void stcat_abs_trfrag_n(const char* base, int plane_bytes, int t0, int lane, AbsFrag<NP>& f) {
  STCAT_UNROLL
  for (int q = 0; q < NP; ++q) f.p[q] = stcat_abs_trfrag(base + q * plane_bytes, t0, lane);
}

// this lane's 16 values of a [S][ld] fp32 row block (row `row`, head dims 16 s + 8 hi .. + 7 for k-step s), times mul, as planes
template <int NP>
static __device__ __forceinline__ void stcat_abs_rowregs(const float* g, int ld, int row, int S, int hi, float mul,
                                                         AbsFrag<NP> (&f)[2]) {
  STCAT_UNROLL
  for (int s = 0; s < 2; ++s) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (row < S) { a = stcat_ld4(g + (long)row * ld + 16 * s + 8 * hi); c = stcat_ld4(g + (long)row * ld + 16 * s + 8 * hi + 4); }
    const float x[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, c.x * mul, c.y * mul, c.z * mul, c.w * mul};
    stcat_abs_split8<NP>(x, f[s]);
  }
}

// ---------------------------------------------------------------------------------------------------
// forward: grid (B*H, ceil(query tiles / NW)), NW waves, one 32-query tile per wave; keys stream through LDS in
// super-chunks of 256 (one for S <= 256), consumed in chunks of 128 with the online-softmax update.  NP planes per operand
// (2 x NP x 16 KB of LDS + the key bias).
// ---------------------------------------------------------------------------------------------------
template <int NW, int NP>
__global__ void __launch_bounds__(64 * NW) mha_bs_fwd_kernel(AttnBsParams p) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SC = 256, PLANE = SC * STCAT_ABS_ROWB;  // 16 KB per plane
  STCAT_DYN_SHARED(char, smem);
  char* Kp = smem; char* Vp = smem + NP * PLANE;
  float* kb = reinterpret_cast<float*>(smem + 2 * NP * PLANE);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int SP = ((p.S + 31) / 32) * 32;
  const int qt = blockIdx.y * NW + wave;
  const int q = qt * 32 + l31;
  const float* Kg = p.K + (long)b * p.S * p.ldk + h * 32;
  const float* Vg = p.V + (long)b * p.S * p.ldv + h * 32;
  // this lane's query row, head dims 16 s + 8 hi .. + 7 for k-step s, pre-scaled (attention.py:283-285), split
  AbsFrag<NP> qf[2];
  stcat_abs_rowregs<NP>(p.Q + (long)b * p.S * p.ldq + h * 32, p.ldq, q, p.S, hi, p.scale, qf);
  f32x16 o;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = STCAT_NEG_INF, l = 0.f;
  for (int sc0 = 0; sc0 < p.S; sc0 += SC) {
    if (sc0) __syncthreads();  // everybody is done with the previous super-chunk
    stcat_abs_stage_n<SC, NP>(Kg, p.ldk, sc0, p.S, Kp, PLANE, t, 64 * NW, 1.f);
    stcat_abs_stage_n<SC, NP>(Vg, p.ldv, sc0, p.S, Vp, PLANE, t, 64 * NW, 1.f);
    for (int i = t; i < SC; i += 64 * NW)
      kb[i] = (sc0 + i < p.S && !(p.kpm && p.kpm[(long)b * p.S + sc0 + i])) ? 0.f : STCAT_NEG_INF;
    __syncthreads();
    STCAT_UNROLL
    for (int ch = 0; ch < 2; ++ch) {
      if (sc0 + ch * 128 >= p.S) break;
      f32x16 sc[4];
      float cmax = STCAT_NEG_INF;
      STCAT_UNROLL
      for (int kt = 0; kt < 4; ++kt) {
        const int k0 = ch * 128 + kt * 32;  // first key of the tile inside the super-chunk
        STCAT_UNROLL
        for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
        if (sc0 + k0 < p.S) {  // tile-uniform
          STCAT_UNROLL
          for (int s = 0; s < 2; ++s) {
            AbsFrag<NP> kf;
            stcat_abs_rowfrag_n<NP>(Kp, PLANE, k0 + l31, 2 * s + hi, kf);
            stcat_abs_mma<NP>(sc[kt], kf, qf[s]);
          }
          STCAT_UNROLL
          for (int r = 0; r < 16; ++r) {
            sc[kt][r] += kb[k0 + (r & 3) + 8 * (r >> 2) + 4 * hi];
            cmax = fmaxf(cmax, sc[kt][r]);
          }
        } else {
          STCAT_UNROLL
          for (int r = 0; r < 16; ++r) sc[kt][r] = STCAT_NEG_INF;
        }
      }
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
      const float mn = fmaxf(m, cmax);
      // a fully masked chunk (mn = -inf) contributes nothing and must not produce NaNs
      const float alpha = (m == STCAT_NEG_INF) ? 0.f : __expf(m - mn);
      float csum = 0.f;
      STCAT_UNROLL
      for (int kt = 0; kt < 4; ++kt) {
        STCAT_UNROLL
        for (int r = 0; r < 16; ++r) {
          const float e_ = (mn == STCAT_NEG_INF) ? 0.f : __expf(sc[kt][r] - mn);
          sc[kt][r] = e_;
          csum += e_;
        }
      }
      csum += __shfl_xor(csum, 32);
      l = l * alpha + csum;
      m = mn;
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      STCAT_UNROLL
      for (int kt = 0; kt < 4; ++kt) {
        const int k0 = ch * 128 + kt * 32;
        if (sc0 + k0 >= p.S) break;
        if (p.drop.thresh) {
          STCAT_UNROLL
          for (int r = 0; r < 16; ++r) {
            const int key = sc0 + k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            sc[kt][r] *= stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
          }
        }
        STCAT_UNROLL
        for (int j = 0; j < 2; ++j) {
          AbsFrag<NP> pf, vf;
          stcat_abs_split_acc<NP>(sc[kt], j, pf);
          stcat_abs_trfrag_n<NP>(Vp, PLANE, k0 + 16 * j, lane, vf);
          stcat_abs_mma<NP>(o, vf, pf);
        }
      }
    }
  }
  if (q < p.S) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    float* Og = p.O + ((long)b * p.S + q) * p.ldo + h * 32 + 4 * hi;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c)  // registers 4 c .. 4 c + 3 = head dims 8 c + 4 hi .. + 3
      stcat_st4(Og + 8 * c, make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv));
    if (hi == 0 && p.lse) p.lse[(long)blockIdx.x * p.S + q] = m + __logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward (S <= 256): one workgroup per (batch, head), NW = ceil(S / 32) waves.  Q (pre-scaled), K, V, dO staged as
// planes; delta_q = dO_q . O_q.  Pass A, wave = query tile: dQ.  Pass B, wave = key tile: dK, dV.
// ---------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(64 * NW) mha_bs_bwd_kernel(AttnBsParams p, const float* Og) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SPc = NW * 32, PLANE = SPc * STCAT_ABS_ROWB;
  STCAT_DYN_SHARED(char, smem);
  char* Qh = smem; char* Ql = smem + PLANE; char* Kh = smem + 2 * PLANE; char* Kl = smem + 3 * PLANE;
  char* Vh = smem + 4 * PLANE; char* Vl = smem + 5 * PLANE; char* Gh = smem + 6 * PLANE; char* Gl = smem + 7 * PLANE;
  float* kb = reinterpret_cast<float*>(smem + 8 * PLANE);
  float* lse = kb + SPc;
  float* dlt = lse + SPc;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int SP = ((p.S + 31) / 32) * 32;
  stcat_abs_stage<SPc>(p.Q + (long)b * p.S * p.ldq + h * 32, p.ldq, 0, p.S, Qh, Ql, t, 64 * NW, p.scale);
  stcat_abs_stage<SPc>(p.K + (long)b * p.S * p.ldk + h * 32, p.ldk, 0, p.S, Kh, Kl, t, 64 * NW, 1.f);
  stcat_abs_stage<SPc>(p.V + (long)b * p.S * p.ldv + h * 32, p.ldv, 0, p.S, Vh, Vl, t, 64 * NW, 1.f);
  stcat_abs_stage<SPc>(p.dO + (long)b * p.S * p.ldo + h * 32, p.ldo, 0, p.S, Gh, Gl, t, 64 * NW, 1.f);
  for (int i = t; i < SPc; i += 64 * NW) {
    kb[i] = (i < p.S && !(p.kpm && p.kpm[(long)b * p.S + i])) ? 0.f : STCAT_NEG_INF;
    float d = 0.f, ls = 0.f;
    if (i < p.S) {
      const float* g = p.dO + ((long)b * p.S + i) * p.ldo + h * 32;
      const float* og = Og + ((long)b * p.S + i) * p.ldo + h * 32;
      STCAT_UNROLL
      for (int c = 0; c < 8; ++c) {
        const float4 a = stcat_ld4(g + 4 * c), o4 = stcat_ld4(og + 4 * c);
        d += a.x * o4.x + a.y * o4.y + a.z * o4.z + a.w * o4.w;
      }
      ls = p.lse[(long)blockIdx.x * p.S + i];
    }
    dlt[i] = d;
    lse[i] = ls;
  }
  __syncthreads();
  // ---- pass A: lane = query column q of tile `wave`; rows of the score tile = keys
  {
    const int q = wave * 32 + l31;
    bf16x8 qh[2], ql[2], gh[2], gl[2];
    STCAT_UNROLL
    for (int s = 0; s < 2; ++s) {
      qh[s] = STCAT_ABS_ROWFRAG(Qh, q, 2 * s + hi); ql[s] = STCAT_ABS_ROWFRAG(Ql, q, 2 * s + hi);
      gh[s] = STCAT_ABS_ROWFRAG(Gh, q, 2 * s + hi); gl[s] = STCAT_ABS_ROWFRAG(Gl, q, 2 * s + hi);
    }
    const float lq = lse[q], dq_ = dlt[q];
    f32x16 dq;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    for (int kt = 0; kt < NW; ++kt) {
      const int k0 = kt * 32;
      f32x16 s_, dp;
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp[r] = 0.f; }
      STCAT_UNROLL
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kh = STCAT_ABS_ROWFRAG(Kh, k0 + l31, 2 * s + hi), kl = STCAT_ABS_ROWFRAG(Kl, k0 + l31, 2 * s + hi);
        const bf16x8 vh = STCAT_ABS_ROWFRAG(Vh, k0 + l31, 2 * s + hi), vl = STCAT_ABS_ROWFRAG(Vl, k0 + l31, 2 * s + hi);
        STCAT_ABS_MMA3(s_, kh, kl, qh[s], ql[s])
        STCAT_ABS_MMA3(dp, vh, vl, gh[s], gl[s])
      }
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float pr = (q < p.S) ? __expf(s_[r] + kb[key] - lq) : 0.f;   // masked / padded keys: exp(-inf) = 0
        float dpv = dp[r];
        if (p.drop.thresh) dpv *= stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
        s_[r] = pr * (dpv - dq_);                                          // dS (the scale rides in the staged Q / below)
      }
      STCAT_UNROLL
      for (int j = 0; j < 2; ++j) {
        bf16x8 dh, dl;
        stcat_abs_split_regs(s_, j, dh, dl);
        const bf16x8 kth = stcat_abs_trfrag(Kh, k0 + 16 * j, lane), ktl = stcat_abs_trfrag(Kl, k0 + 16 * j, lane);
        STCAT_ABS_MMA3(dq, kth, ktl, dh, dl)
      }
    }
    if (q < p.S) {
      float* g = p.dQ + ((long)b * p.S + q) * p.ldg + h * 32 + 4 * hi;
      STCAT_UNROLL
      for (int c = 0; c < 4; ++c)
        stcat_st4(g + 8 * c, make_float4(dq[4 * c] * p.scale, dq[4 * c + 1] * p.scale, dq[4 * c + 2] * p.scale,
                                         dq[4 * c + 3] * p.scale));
    }
  }
  // ---- pass B: lane = key column of tile `wave`; rows of the score tile = queries
  {
    const int key = wave * 32 + l31;
    bf16x8 kh[2], kl[2], vh[2], vl[2];
    STCAT_UNROLL
    for (int s = 0; s < 2; ++s) {
      kh[s] = STCAT_ABS_ROWFRAG(Kh, key, 2 * s + hi); kl[s] = STCAT_ABS_ROWFRAG(Kl, key, 2 * s + hi);
      vh[s] = STCAT_ABS_ROWFRAG(Vh, key, 2 * s + hi); vl[s] = STCAT_ABS_ROWFRAG(Vl, key, 2 * s + hi);
    }
    const float kbias = kb[key];
    f32x16 dv, dk;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
    for (int qt = 0; qt < NW; ++qt) {
      const int q0 = qt * 32;
      f32x16 s_, dp;
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp[r] = 0.f; }
      STCAT_UNROLL
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = STCAT_ABS_ROWFRAG(Qh, q0 + l31, 2 * s + hi), al = STCAT_ABS_ROWFRAG(Ql, q0 + l31, 2 * s + hi);
        const bf16x8 bh = STCAT_ABS_ROWFRAG(Gh, q0 + l31, 2 * s + hi), bl = STCAT_ABS_ROWFRAG(Gl, q0 + l31, 2 * s + hi);
        STCAT_ABS_MMA3(s_, ah, al, kh[s], kl[s])   // S[q][key]
        STCAT_ABS_MMA3(dp, bh, bl, vh[s], vl[s])   // dP[q][key] = dO_q . V_key
      }
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float pr = (qq < p.S) ? __expf(s_[r] + kbias - lse[qq]) : 0.f;
        const float dm = p.drop.thresh ? stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + qq) : 1.f;
        s_[r] = pr * (dp[r] * dm - dlt[qq]);   // dS[q][key]
        dp[r] = pr * dm;                        // P' (dropped probabilities) for dV
      }
      STCAT_UNROLL
      for (int j = 0; j < 2; ++j) {
        bf16x8 ph, pl_, dh, dl;
        stcat_abs_split_regs(dp, j, ph, pl_);
        stcat_abs_split_regs(s_, j, dh, dl);
        const bf16x8 gth = stcat_abs_trfrag(Gh, q0 + 16 * j, lane), gtl = stcat_abs_trfrag(Gl, q0 + 16 * j, lane);
        const bf16x8 qth = stcat_abs_trfrag(Qh, q0 + 16 * j, lane), qtl = stcat_abs_trfrag(Ql, q0 + 16 * j, lane);
        STCAT_ABS_MMA3(dv, gth, gtl, ph, pl_)   // dV^T[d][key] += dO^T[d][q] P'[q][key]
        STCAT_ABS_MMA3(dk, qth, qtl, dh, dl)    // dK^T[d][key] += (scale Q)^T[d][q] dS[q][key]
      }
    }
    if (key < p.S) {
      float* gv = p.dV + ((long)b * p.S + key) * p.ldgv + h * 32 + 4 * hi;
      float* gk = p.dK + ((long)b * p.S + key) * p.ldg + h * 32 + 4 * hi;
      STCAT_UNROLL
      for (int c = 0; c < 4; ++c) {
        stcat_st4(gv + 8 * c, make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]));
        stcat_st4(gk + 8 * c, make_float4(dk[4 * c], dk[4 * c + 1], dk[4 * c + 2], dk[4 * c + 3]));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward in TWO launches (round 6; any NP, used for NP = 3 where Q, K, V, dO as three planes each would be 172 KB of LDS
// at S = 224): the probabilities are recomputed on the bf16 pipe from the row log-sum-exp in both.
//   dQ kernel:    wave = query tile.  K, V planes in LDS (2 NP planes), this wave's Q (pre-scaled) and dO rows split in
//                 registers straight from HBM, delta_q = dO_q . O_q from the same rows.
//   dK/dV kernel: wave = key tile.  Q (pre-scaled), dO planes in LDS, this wave's K and V rows in registers.
// One workgroup per (batch, head), NW = ceil(S / 32) waves, S <= 256.
// ---------------------------------------------------------------------------------------------------
template <int NW, int NP>
__global__ void __launch_bounds__(64 * NW) mha_bs_bwd_dq_kernel(AttnBsParams p, const float* Og) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SPc = NW * 32, PLANE = SPc * STCAT_ABS_ROWB;
  STCAT_DYN_SHARED(char, smem);
  char* Kp = smem; char* Vp = smem + NP * PLANE;
  float* kb = reinterpret_cast<float*>(smem + 2 * NP * PLANE);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int SP = ((p.S + 31) / 32) * 32;
  stcat_abs_stage_n<SPc, NP>(p.K + (long)b * p.S * p.ldk + h * 32, p.ldk, 0, p.S, Kp, PLANE, t, 64 * NW, 1.f);
  stcat_abs_stage_n<SPc, NP>(p.V + (long)b * p.S * p.ldv + h * 32, p.ldv, 0, p.S, Vp, PLANE, t, 64 * NW, 1.f);
  for (int i = t; i < SPc; i += 64 * NW) kb[i] = (i < p.S && !(p.kpm && p.kpm[(long)b * p.S + i])) ? 0.f : STCAT_NEG_INF;
  const int q = wave * 32 + l31;
  AbsFrag<NP> qf[2], gf[2];
  stcat_abs_rowregs<NP>(p.Q + (long)b * p.S * p.ldq + h * 32, p.ldq, q, p.S, hi, p.scale, qf);
  stcat_abs_rowregs<NP>(p.dO + (long)b * p.S * p.ldo + h * 32, p.ldo, q, p.S, hi, 1.f, gf);
  float dq_ = 0.f, lq = 0.f;
  if (q < p.S) {
    const float* g = p.dO + ((long)b * p.S + q) * p.ldo + h * 32 + 16 * hi;     // (half a row per lane of the pair)
    const float* og = Og + ((long)b * p.S + q) * p.ldo + h * 32 + 16 * hi;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      const float4 a = stcat_ld4(g + 4 * c), o4 = stcat_ld4(og + 4 * c);
      dq_ += a.x * o4.x + a.y * o4.y + a.z * o4.z + a.w * o4.w;
    }
    lq = p.lse[(long)blockIdx.x * p.S + q];
  }
  dq_ += __shfl_xor(dq_, 32);
  __syncthreads();
  f32x16 dq;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  for (int kt = 0; kt < NW; ++kt) {
    const int k0 = kt * 32;
    f32x16 s_, dp;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp[r] = 0.f; }
    STCAT_UNROLL
    for (int s = 0; s < 2; ++s) {
      AbsFrag<NP> kf, vf;
      stcat_abs_rowfrag_n<NP>(Kp, PLANE, k0 + l31, 2 * s + hi, kf);
      stcat_abs_rowfrag_n<NP>(Vp, PLANE, k0 + l31, 2 * s + hi, vf);
      stcat_abs_mma<NP>(s_, kf, qf[s]);
      stcat_abs_mma<NP>(dp, vf, gf[s]);
    }
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = (q < p.S) ? __expf(s_[r] + kb[key] - lq) : 0.f;   // masked / padded keys: exp(-inf) = 0
      float dpv = dp[r];
      if (p.drop.thresh) dpv *= stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + q);
      s_[r] = pr * (dpv - dq_);                                          // dS (the scale rides in the staged Q / below)
    }
    STCAT_UNROLL
    for (int j = 0; j < 2; ++j) {
      AbsFrag<NP> df, ktf;
      stcat_abs_split_acc<NP>(s_, j, df);
      stcat_abs_trfrag_n<NP>(Kp, PLANE, k0 + 16 * j, lane, ktf);
      stcat_abs_mma<NP>(dq, ktf, df);
    }
  }
  if (q < p.S) {
    float* g = p.dQ + ((long)b * p.S + q) * p.ldg + h * 32 + 4 * hi;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c)
      stcat_st4(g + 8 * c, make_float4(dq[4 * c] * p.scale, dq[4 * c + 1] * p.scale, dq[4 * c + 2] * p.scale,
                                       dq[4 * c + 3] * p.scale));
  }
}

template <int NW, int NP>
__global__ void __launch_bounds__(64 * NW) mha_bs_bwd_dkv_kernel(AttnBsParams p, const float* Og) {
  p.drop = stcat_drop_resolve(p.drop);
  constexpr int SPc = NW * 32, PLANE = SPc * STCAT_ABS_ROWB;
  STCAT_DYN_SHARED(char, smem);
  char* Qp = smem; char* Gp = smem + NP * PLANE;
  float* lse = reinterpret_cast<float*>(smem + 2 * NP * PLANE);
  float* dlt = lse + SPc;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int SP = ((p.S + 31) / 32) * 32;
  stcat_abs_stage_n<SPc, NP>(p.Q + (long)b * p.S * p.ldq + h * 32, p.ldq, 0, p.S, Qp, PLANE, t, 64 * NW, p.scale);
  stcat_abs_stage_n<SPc, NP>(p.dO + (long)b * p.S * p.ldo + h * 32, p.ldo, 0, p.S, Gp, PLANE, t, 64 * NW, 1.f);
  for (int i = t; i < SPc; i += 64 * NW) {
    float d = 0.f, ls = 0.f;
    if (i < p.S) {
      const float* g = p.dO + ((long)b * p.S + i) * p.ldo + h * 32;
      const float* og = Og + ((long)b * p.S + i) * p.ldo + h * 32;
      STCAT_UNROLL
      for (int c = 0; c < 8; ++c) {
        const float4 a = stcat_ld4(g + 4 * c), o4 = stcat_ld4(og + 4 * c);
        d += a.x * o4.x + a.y * o4.y + a.z * o4.z + a.w * o4.w;
      }
      ls = p.lse[(long)blockIdx.x * p.S + i];
    }
    dlt[i] = d;
    lse[i] = ls;
  }
  const int key = wave * 32 + l31;
  AbsFrag<NP> kf[2], vf[2];
  stcat_abs_rowregs<NP>(p.K + (long)b * p.S * p.ldk + h * 32, p.ldk, key, p.S, hi, 1.f, kf);
  stcat_abs_rowregs<NP>(p.V + (long)b * p.S * p.ldv + h * 32, p.ldv, key, p.S, hi, 1.f, vf);
  const float kbias = (key < p.S && !(p.kpm && p.kpm[(long)b * p.S + key])) ? 0.f : STCAT_NEG_INF;
  __syncthreads();
  f32x16 dv, dk;
  STCAT_UNROLL
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  for (int qt = 0; qt < NW; ++qt) {
    const int q0 = qt * 32;
    f32x16 s_, dp;
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp[r] = 0.f; }
    STCAT_UNROLL
    for (int s = 0; s < 2; ++s) {
      AbsFrag<NP> af, bf;
      stcat_abs_rowfrag_n<NP>(Qp, PLANE, q0 + l31, 2 * s + hi, af);
      stcat_abs_rowfrag_n<NP>(Gp, PLANE, q0 + l31, 2 * s + hi, bf);
      stcat_abs_mma<NP>(s_, af, kf[s]);   // S[q][key]
      stcat_abs_mma<NP>(dp, bf, vf[s]);   // dP[q][key] = dO_q . V_key
    }
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float pr = (qq < p.S) ? __expf(s_[r] + kbias - lse[qq]) : 0.f;
      const float dm = p.drop.thresh ? stcat_drop_mul(p.drop, ((unsigned long long)blockIdx.x * SP + key) * SP + qq) : 1.f;
      s_[r] = pr * (dp[r] * dm - dlt[qq]);   // dS[q][key]
      dp[r] = pr * dm;                        // P' (dropped probabilities) for dV
    }
    STCAT_UNROLL
    for (int j = 0; j < 2; ++j) {
      AbsFrag<NP> pf, df, gtf, qtf;
      stcat_abs_split_acc<NP>(dp, j, pf);
      stcat_abs_split_acc<NP>(s_, j, df);
      stcat_abs_trfrag_n<NP>(Gp, PLANE, q0 + 16 * j, lane, gtf);
      stcat_abs_trfrag_n<NP>(Qp, PLANE, q0 + 16 * j, lane, qtf);
      stcat_abs_mma<NP>(dv, gtf, pf);   // dV^T[d][key] += dO^T[d][q] P'[q][key]
      stcat_abs_mma<NP>(dk, qtf, df);   // dK^T[d][key] += (scale Q)^T[d][q] dS[q][key]
    }
  }
  if (key < p.S) {
    float* gv = p.dV + ((long)b * p.S + key) * p.ldgv + h * 32 + 4 * hi;
    float* gk = p.dK + ((long)b * p.S + key) * p.ldg + h * 32 + 4 * hi;
    STCAT_UNROLL
    for (int c = 0; c < 4; ++c) {
      stcat_st4(gv + 8 * c, make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]));
      stcat_st4(gk + 8 * c, make_float4(dk[4 * c], dk[4 * c + 1], dk[4 * c + 2], dk[4 * c + 3]));
    }
  }
}
