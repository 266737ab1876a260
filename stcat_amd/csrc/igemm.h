// Implicit-GEMM kernels on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak on gfx950).
//
// One family covers every dense contraction of the hot path:
//   fwd   : conv (NHWC, OHWI weights) + FrozenBN scale/bias + residual + ReLU epilogue
//           (models/vision_model/backbone.py:56-66 folded in); a Linear is the 1x1 case
//           on an [M,1,1,K] tensor (torch.nn.Linear call sites in modal_encoder.py /
//           query_decoder.py / pipeline.py:41).
//   dgrad : data gradient of the same conv / dX of a Linear.
//   wgrad : weight gradient, split over the M (pixel) reduction with fp32 atomics.
//
// Tiling: 256 threads = 4 waves (2x2); each wave owns (BM/2)x(BN/2) of the BMxBN block
// tile as TMxTN accumulators of 32x32 (16 VGPRs each).  K step 16, LDS tiles stored
// reduction-major [16][BM+4] so an MFMA operand fetch is one conflict-free ds_read_b32
// (lanes 0-31 consecutive rows, lanes 32-63 the next k).  Global->LDS goes through
// registers (float4), double-buffered: tile t+1 is in flight while tile t is multiplied.
// fp32 MFMA issues every 64 cycles per SIMD, so the LDS/VALU side has ample slack; the
// kernel is matrix-pipe bound by construction.  Workgroup ids are remapped so that the
// N-tiles sharing one activation row-block run back-to-back on the same XCD (L2 reuse).
#pragma once
#include "stcat_platform.h"
#include "stcat_rng.h"

struct IgemmGeom {
  int H, W, C, ld;         // gathered NHWC tensor: spatial dims, channels, pixel stride (floats)
  int OH, OW;              // spatial extent of the row index space (m -> nb, oh, ow)
  int KH, KW;
  int mul, off, sgn, div;  // h = oh*mul + off + kh*sgn; if div>1: need h%div==0, then h/=div
  unsigned mg_ow, sh_ow, mg_ohw, sh_ohw;  // multiply-shift constants for m / OW and m / (OH*OW)  (stcat_fastdiv)
};

// floor(n / d) for 0 <= n < 2^31 as one 32x32->64 multiply and a shift: mg = floor(2^(31+sh) / d) + 1,
// sh = ceil(log2 d)  (Granlund & Montgomery).  The host fills (mg, sh) with stcat_fastdiv_magic.
static __device__ __forceinline__ int stcat_fastdiv(int n, unsigned mg, unsigned sh) {
  return (int)(((unsigned long long)(unsigned)n * mg) >> (31u + sh));
}
static inline void stcat_fastdiv_magic(int d, unsigned* mg, unsigned* sh) {
  unsigned s = 0;
  while ((1ll << s) < (long long)d) ++s;
  *sh = s;
  *mg = (unsigned)(((1ull << (31 + s)) / (unsigned long long)d) + 1ull);
}

struct IgemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* scale;  // per output column (FrozenBN scale) or null
  const float* bias;   // per output column or null
  const float* res;    // residual [M][ldr] or null
  const float* mask;   // dgrad only: activation y [M][ldc] of the producing conv — output is zeroed where y <= 0 ...
  const float* mscale; // ... and multiplied by that conv's FrozenBN scale[n] (fused ReLU+BN backward), or null
  float* C2;           // dgrad only: optional second output C2 = C * c2scale[n] (dz and dz*scale of a block boundary)
  const float* c2scale;
  float* rowsum;       // wgrad only: optional rowsum[m] += sum_k A[k][m] (the bias gradient of a Linear)
  float* sk_ws;        // stream-K forward only: partial-tile workspace, 2 slots of BM*BN floats per worker
  int M, N, K;
  int ldb, ldc, ldr;
  int c_group, c_group_stride;  // output row m -> (m / c_group) * c_group_stride + (m % c_group) * ldc
  int relu;
  int k_chunk;  // wgrad: rows of the M reduction per grid.z slice (multiple of 16)
  unsigned a_bytes, b_bytes;  // extents of A and B for the bounds-checked buffer loads (split-bf16 kernels)
  unsigned b_tap_stride;      // split-bf16 fwd-path B addressing: byte offset of K-tile = tap*b_tap_stride + c0*4
  // split-bf16 forward kernel only (round 5): dropout applied to the epilogue's result (after bias / residual / ReLU),
  // element index m * N + n of the site's counter range — the FFN's `dropout(relu(linear1 x))` (modal_encoder.py:239,
  // query_decoder.py:435, 657) without its own pass over the [M, 2048] tensor
  DropParams drop;
  float mask_gain;            // with `mask`: the kept elements are also multiplied by this (0 = 1): the 1 / (1 - p) of a dropout
                              // whose output IS the mask tensor (y > 0 <=> ReLU passed and the element was kept)
  IgemmGeom g;
};

static __device__ __forceinline__ int stcat_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// gather coordinate -> element offset of the pixel, or -1 when out of range / not on the stride lattice
static __device__ __forceinline__ long stcat_gather_pix(const IgemmGeom& g, int nb, int bh, int bw, int kh, int kw) {
  int h = bh + kh * g.sgn, w = bw + kw * g.sgn;
  if (g.div > 1) {
    if ((h % g.div) != 0 || (w % g.div) != 0) return -1;
    h /= g.div;
    w /= g.div;
  }
  if (h < 0 || h >= g.H || w < 0 || w >= g.W) return -1;
  return ((long)(nb * g.H + h) * g.W + w) * g.ld;
}

#define STCAT_IGEMM_COMPUTE(AS, BS)                                                        \
  STCAT_UNROLL                                                                             \
  for (int kk = 0; kk < BK; kk += 2) {                                                     \
    float a_[TM], b_[TN];                                                                  \
    STCAT_UNROLL                                                                           \
    for (int tm = 0; tm < TM; ++tm) a_[tm] = (AS)[(kk + hi) * LDA + wm * TM * 32 + tm * 32 + l31]; \
    STCAT_UNROLL                                                                           \
    for (int tn = 0; tn < TN; ++tn) b_[tn] = (BS)[(kk + hi) * LDBS + wn * TN * 32 + tn * 32 + l31]; \
    STCAT_UNROLL                                                                           \
    for (int tm = 0; tm < TM; ++tm) {                                                      \
      STCAT_UNROLL                                                                         \
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = STCAT_MFMA_32x32x2(a_[tm], b_[tn], acc[tm][tn]); \
    }                                                                                      \
  }

// Two-level accumulation: the MFMA accumulates one sequential fp32 chain per output; over K up to 4608
// that costs ~sqrt(K) ulps.  Every 8 K-tiles (128 terms) the running tile is folded into `tot`, which
// brings round-off to the level of a blocked CPU sum at the price of 16*TM*TN more VGPRs.
#define STCAT_ACC_FLUSH(KT)                                              \
  if ((((KT) & 7) == 7) || (KT) == nk - 1) {                             \
    STCAT_UNROLL                                                         \
    for (int i_ = 0; i_ < TM; ++i_) {                                    \
      STCAT_UNROLL                                                       \
      for (int j_ = 0; j_ < TN; ++j_) {                                  \
        STCAT_UNROLL                                                     \
        for (int r_ = 0; r_ < 16; ++r_) {                                \
          tot[i_][j_][r_] += acc[i_][j_][r_];                            \
          acc[i_][j_][r_] = 0.f;                                         \
        }                                                                \
      }                                                                  \
    }                                                                    \
  }

// ---------------------------------------------------------------------------------
// forward: C[m][n] = epi( sum_r Agather[m][r] * B[n][r] ),  r = (kh, kw, ci), ci fastest
// ---------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(256) igemm_fwd_kernel(IgemmParams p) {
  constexpr int BK = 16, LDA = BM + 4, LDBS = BN + 4, TM = BM / 64, TN = BN / 64;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDBS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int num_n = p.N / BN;
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;
  const IgemmGeom g = p.g;

  // per-thread gather rows (A) : row = t/4 + 64*j, 4 consecutive r at (t%4)*4
  const int r4 = (t & 3) * 4;
  int a_nb[TM], a_bh[TM], a_bw[TM];
  STCAT_UNROLL
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (t >> 2) + 64 * j;
    if (m < p.M) {
      const int ohw = g.OH * g.OW;
      const int nb = m / ohw, rem = m - nb * ohw, oh = rem / g.OW, ow = rem - oh * g.OW;
      a_nb[j] = nb;
      a_bh[j] = oh * g.mul + g.off;
      a_bw[j] = ow * g.mul + g.off;
    } else {
      a_nb[j] = -1;
      a_bh[j] = 0;
      a_bw[j] = 0;
    }
  }
  const float* brow[TN];
  STCAT_UNROLL
  for (int j = 0; j < TN; ++j) brow[j] = p.B + (long)(n0 + (t >> 2) + 64 * j) * p.ldb + r4;

  f32x16 acc[TM][TN], tot[TM][TN];
  STCAT_UNROLL
  for (int i = 0; i < TM; ++i) {
    STCAT_UNROLL
    for (int j = 0; j < TN; ++j) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    }
  }

  float4 ra[TM], rb[TN];
  const int nk = p.K / BK;
#define STCAT_FWD_LOAD(KT)                                                               \
  {                                                                                      \
    const int r0 = (KT) * BK, tap = r0 / g.C, ci0 = r0 - tap * g.C;                      \
    const int kh = tap / g.KW, kw = tap - kh * g.KW;                                     \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TM; ++j) {                                                       \
      long pix = a_nb[j] < 0 ? -1 : stcat_gather_pix(g, a_nb[j], a_bh[j], a_bw[j], kh, kw); \
      ra[j] = pix < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : stcat_ld4(p.A + pix + ci0 + r4); \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TN; ++j) rb[j] = stcat_ld4(brow[j] + r0);                        \
  }
#define STCAT_FWD_STORE(BUF)                                                             \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TM; ++j) {                                                       \
      float* d = &As[BUF][r4 * LDA + (t >> 2) + 64 * j];                                 \
      d[0] = ra[j].x; d[LDA] = ra[j].y; d[2 * LDA] = ra[j].z; d[3 * LDA] = ra[j].w;      \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TN; ++j) {                                                       \
      float* d = &Bs[BUF][r4 * LDBS + (t >> 2) + 64 * j];                                \
      d[0] = rb[j].x; d[LDBS] = rb[j].y; d[2 * LDBS] = rb[j].z; d[3 * LDBS] = rb[j].w;   \
    }                                                                                    \
  }
  STCAT_FWD_LOAD(0)
  STCAT_FWD_STORE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) STCAT_FWD_LOAD(kt + 1)
    STCAT_IGEMM_COMPUTE(As[buf], Bs[buf])
    STCAT_ACC_FLUSH(kt)
    if (kt + 1 < nk) STCAT_FWD_STORE(buf ^ 1)
    __syncthreads();
  }
#undef STCAT_FWD_LOAD
#undef STCAT_FWD_STORE

  // epilogue: lane -> column, regs -> rows
  STCAT_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * TN * 32 + tn * 32 + l31;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float bi = p.bias ? p.bias[n] : 0.f;
    STCAT_UNROLL
    for (int tm = 0; tm < TM; ++tm) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m < p.M) {
          float val = tot[tm][tn][r] * sc + bi;
          if (p.res) val += p.res[(long)m * p.ldr + n];
          if (p.relu) val = fmaxf(val, 0.f);
          const long row = p.c_group >= p.M ? (long)m * p.ldc
                                            : (long)(m / p.c_group) * p.c_group_stride + (long)(m % p.c_group) * p.ldc;
          p.C[row + n] = val;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// dgrad: C[m][n] = epi( sum_r Agather[m][r] * W[co(r)][tap(r)][n] ), r = (kh, kw, co), co fastest
// A gathers dY (g.C = Cout, dgrad coordinate map); W is OHWI with row stride ldb (= KH*KW*Cin);
// N = Cin.  A Linear's dX is the 1x1 case: dX[m][k] = sum_n dY[m][n] W[n][k].
// ---------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(256) igemm_dgrad_kernel(IgemmParams p) {
  constexpr int BK = 16, LDA = BM + 4, LDBS = BN + 4, TM = BM / 64, TN = BN / 64;
  constexpr int BF4 = BN / 4;           // float4 per B tile row
  constexpr int BJ = (BK * BF4) / 256;  // B float4 per thread
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDBS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int num_n = p.N / BN;
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;
  const IgemmGeom g = p.g;

  const int r4 = (t & 3) * 4;
  int a_nb[TM], a_bh[TM], a_bw[TM];
  STCAT_UNROLL
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (t >> 2) + 64 * j;
    if (m < p.M) {
      const int ohw = g.OH * g.OW;
      const int nb = m / ohw, rem = m - nb * ohw, oh = rem / g.OW, ow = rem - oh * g.OW;
      a_nb[j] = nb;
      a_bh[j] = oh * g.mul + g.off;
      a_bw[j] = ow * g.mul + g.off;
    } else {
      a_nb[j] = -1;
      a_bh[j] = 0;
      a_bw[j] = 0;
    }
  }
  f32x16 acc[TM][TN], tot[TM][TN];
  STCAT_UNROLL
  for (int i = 0; i < TM; ++i) {
    STCAT_UNROLL
    for (int j = 0; j < TN; ++j) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    }
  }
  float4 ra[TM], rb[BJ];
  const int nk = p.K / BK;
#define STCAT_DG_LOAD(KT)                                                                \
  {                                                                                      \
    const int r0 = (KT) * BK, tap = r0 / g.C, co0 = r0 - tap * g.C;                      \
    const int kh = tap / g.KW, kw = tap - kh * g.KW;                                     \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TM; ++j) {                                                       \
      long pix = a_nb[j] < 0 ? -1 : stcat_gather_pix(g, a_nb[j], a_bh[j], a_bw[j], kh, kw); \
      ra[j] = pix < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : stcat_ld4(p.A + pix + co0 + r4); \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < BJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / BF4, c4 = i - rr * BF4;                        \
      rb[j] = stcat_ld4(p.B + (long)(co0 + rr) * p.ldb + (long)tap * p.N + n0 + c4 * 4); \
    }                                                                                    \
  }
#define STCAT_DG_STORE(BUF)                                                              \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < TM; ++j) {                                                       \
      float* d = &As[BUF][r4 * LDA + (t >> 2) + 64 * j];                                 \
      d[0] = ra[j].x; d[LDA] = ra[j].y; d[2 * LDA] = ra[j].z; d[3 * LDA] = ra[j].w;      \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < BJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / BF4, c4 = i - rr * BF4;                        \
      stcat_st4(&Bs[BUF][rr * LDBS + c4 * 4], rb[j]);                                    \
    }                                                                                    \
  }
  STCAT_DG_LOAD(0)
  STCAT_DG_STORE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) STCAT_DG_LOAD(kt + 1)
    STCAT_IGEMM_COMPUTE(As[buf], Bs[buf])
    STCAT_ACC_FLUSH(kt)
    if (kt + 1 < nk) STCAT_DG_STORE(buf ^ 1)
    __syncthreads();
  }
#undef STCAT_DG_LOAD
#undef STCAT_DG_STORE

  STCAT_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * TN * 32 + tn * 32 + l31;
    const float msc = p.mscale ? p.mscale[n] : 1.f;
    STCAT_UNROLL
    for (int tm = 0; tm < TM; ++tm) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m < p.M) {
          float val = tot[tm][tn][r];
          if (p.res) val += p.res[(long)m * p.ldr + n];  // fused gradient accumulation
          if (p.mask) val = p.mask[(long)m * p.ldc + n] > 0.f ? val * msc : 0.f;
          p.C[(long)m * p.ldc + n] = val;
          if (p.C2) p.C2[(long)m * p.ldc + n] = val * p.c2scale[n];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// wgrad: C[row][col] += sum_m A[m][row] * Xgather[m][col], row = co, col = (kh, kw, ci)
// A = dY [M][lda = g-independent p.ldb], gathered X described by g (forward coordinate map).
// The M reduction is split over grid.z in chunks of p.k_chunk rows; partial tiles are
// accumulated with fp32 atomics into a zero-initialised C (OHWI, row stride ldc).
// Requires N-tile inside one tap: g.C % BN == 0.
// ---------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(256) igemm_wgrad_kernel(IgemmParams p) {
  constexpr int BK = 16, LDA = BM + 4, LDBS = BN + 4, TM = BM / 64, TN = BN / 64;
  constexpr int AF4 = BM / 4, AJ = (BK * AF4) / 256;
  constexpr int BF4 = BN / 4, BJ = (BK * BF4) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDBS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int num_n = p.N / BN;
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;  // m0: first co row, n0: first (tap,ci) column
  const IgemmGeom g = p.g;
  const int tap = n0 / g.C, ci0 = n0 - tap * g.C;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int red0 = blockIdx.z * p.k_chunk;
  const int red1 = min(p.K, red0 + p.k_chunk);
  const int ohw = g.OH * g.OW;

  f32x16 acc[TM][TN];
  STCAT_UNROLL
  for (int i = 0; i < TM; ++i) {
    STCAT_UNROLL
    for (int j = 0; j < TN; ++j) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  }
  float4 ra[AJ], rb[BJ];
  const int nk = (red1 - red0 + BK - 1) / BK;
#define STCAT_WG_LOAD(KT)                                                                \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < AJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / AF4, c4 = i - rr * AF4;                        \
      const int m = red0 + (KT) * BK + rr;                                               \
      ra[j] = m < red1 ? stcat_ld4(p.A + (long)m * p.ldb + m0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < BJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / BF4, c4 = i - rr * BF4;                        \
      const int m = red0 + (KT) * BK + rr;                                               \
      long pix = -1;                                                                     \
      if (m < red1) {                                                                    \
        const int nb = m / ohw, rem = m - nb * ohw, oh = rem / g.OW, ow = rem - oh * g.OW; \
        pix = stcat_gather_pix(g, nb, oh * g.mul + g.off, ow * g.mul + g.off, kh, kw);   \
      }                                                                                  \
      rb[j] = pix < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : stcat_ld4(p.B + pix + ci0 + c4 * 4); \
    }                                                                                    \
  }
#define STCAT_WG_STORE(BUF)                                                              \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < AJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / AF4, c4 = i - rr * AF4;                        \
      stcat_st4(&As[BUF][rr * LDA + c4 * 4], ra[j]);                                     \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int j = 0; j < BJ; ++j) {                                                       \
      const int i = t + 256 * j, rr = i / BF4, c4 = i - rr * BF4;                        \
      stcat_st4(&Bs[BUF][rr * LDBS + c4 * 4], rb[j]);                                    \
    }                                                                                    \
  }
  if (nk > 0) {
    STCAT_WG_LOAD(0)
    STCAT_WG_STORE(0)
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) STCAT_WG_LOAD(kt + 1)
    STCAT_IGEMM_COMPUTE(As[buf], Bs[buf])
    if (kt + 1 < nk) STCAT_WG_STORE(buf ^ 1)
    __syncthreads();
  }
#undef STCAT_WG_LOAD
#undef STCAT_WG_STORE
  if (nk == 0) return;
  STCAT_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * TN * 32 + tn * 32 + l31;
    STCAT_UNROLL
    for (int tm = 0; tm < TM; ++tm) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        atomicAdd(&p.C[(long)m * p.ldc + n], acc[tm][tn][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// stem: 7x7/2 conv on the NCHW frame tensor -> NHWC, K = 3*49 = 147 (r = ci*49 + kh*7 + kw,
// the OIHW weight order), FrozenBN + ReLU epilogue.  Scalar gather (no 16-B alignment on
// either operand); the frame tensor is read straight from its [T,3,H,W] layout.
// ---------------------------------------------------------------------------------
#define STCAT_STEM_KMAX 160  // 3 * 7 * 7 = 147 reduction rows, padded to whole K-tiles of 16
// U8 = true: the frames arrive as the video decoder leaves them — uint8 [n][H][W][3] (HWC) — and the input pipeline's
// ToTensor + Normalize (datasets/vidstg.py:140, datasets/transforms.py:155-168: (x / 255 - mean[c]) / std[c]) is applied
// inside the gather: a quarter of the input bytes of the fp32 [n,3,H,W] tensor (38.5 MB instead of 154 MB at C3), no
// separate normalisation pass.  p.scale2u = 1 / (255 std[c]), p.shift2u = -mean[c] / std[c] ride in p.mscale / p.c2scale.
template <bool U8>
__global__ void __launch_bounds__(256) igemm_stem_kernel(IgemmParams p) {
  constexpr int BM = 128, BN = 64, BK = 16, LDA = BM + 4, LDBS = BN + 4, TM = 2, TN = 1;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDBS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = 0;
  const IgemmGeom g = p.g;  // H, W input dims; OH, OW output dims; C = 3; KH = KW = 7
  // A: thread -> row t%128, k-rows (t/128)*8 .. +7 ;  B: thread -> col t%64, k-rows (t/64)*4 .. +3
  const int am = t & 127, ak = (t >> 7) * 8;
  const int bn = t & 63, bk = (t >> 6) * 4;
  int nb = -1, bh = 0, bw = 0;
  {
    const int m = m0 + am;
    if (m < p.M) {
      const int ohw = g.OH * g.OW;
      nb = m / ohw;
      const int rem = m - nb * ohw, oh = rem / g.OW, ow = rem - oh * g.OW;
      bh = oh * g.mul + g.off;
      bw = ow * g.mul + g.off;
    }
  }
  f32x16 acc[TM][TN];
  STCAT_UNROLL
  for (int i = 0; i < TM; ++i) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  }
  float ra[8], rb[4];
  const int khw = g.KH * g.KW;
  const int nk = (p.K + BK - 1) / BK;
  // (ci, kh, kw) of a reduction index depend on the index alone: decode all <= 160 of them once per workgroup
  // (the per-element divisions were ~60 VALU ops each and made the kernel index-math-bound)
  __shared__ int tdh[STCAT_STEM_KMAX], tdw[STCAT_STEM_KMAX], toff[STCAT_STEM_KMAX];
  __shared__ float tsc[STCAT_STEM_KMAX], tsh[STCAT_STEM_KMAX];
  if (t < STCAT_STEM_KMAX) {
    int dh = 1 << 20, dw_ = 0, off = 0;  // rows past K: forced out of range -> zero fill
    float sc_ = 0.f, sh_ = 0.f;
    if (t < p.K) {
      const int ci = t / khw, rem = t - ci * khw;
      dh = rem / g.KW;
      dw_ = rem - dh * g.KW;
      off = U8 ? (dh * g.W + dw_) * g.C + ci : (ci * g.H + dh) * g.W + dw_;
      if (U8) { sc_ = p.mscale[ci]; sh_ = p.c2scale[ci]; }
    }
    tdh[t] = dh; tdw[t] = dw_; toff[t] = off; tsc[t] = sc_; tsh[t] = sh_;
  }
  __syncthreads();
  // fp32 NCHW: + toff[r] = ((nb*C + ci)*H + bh + kh)*W + bw + kw;  uint8 HWC: ((nb*H + bh + kh)*W + bw + kw)*C + ci
  const long abase = U8 ? (((long)nb * g.H + bh) * g.W + bw) * g.C : ((long)nb * g.C * g.H + bh) * g.W + bw;
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.A);
#define STCAT_STEM_LOAD(KT)                                                              \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int e = 0; e < 8; ++e) {                                                        \
      const int r = (KT) * BK + ak + e;                                                  \
      const int h = bh + tdh[r], w = bw + tdw[r];                                        \
      float val = 0.f;                                                                   \
      if (nb >= 0 && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W)                         \
        val = U8 ? (float)A8[abase + toff[r]] * tsc[r] + tsh[r] : p.A[abase + toff[r]];                  \
      ra[e] = val;                                                                       \
    }                                                                                    \
    STCAT_UNROLL                                                                         \
    for (int e = 0; e < 4; ++e) {                                                        \
      const int r = (KT) * BK + bk + e;                                                  \
      rb[e] = r < p.K ? p.B[(long)(n0 + bn) * p.ldb + r] : 0.f;                          \
    }                                                                                    \
  }
#define STCAT_STEM_STORE(BUF)                                                            \
  {                                                                                      \
    STCAT_UNROLL                                                                         \
    for (int e = 0; e < 8; ++e) As[BUF][(ak + e) * LDA + am] = ra[e];                    \
    STCAT_UNROLL                                                                         \
    for (int e = 0; e < 4; ++e) Bs[BUF][(bk + e) * LDBS + bn] = rb[e];                   \
  }
  STCAT_STEM_LOAD(0)
  STCAT_STEM_STORE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) STCAT_STEM_LOAD(kt + 1)
    STCAT_IGEMM_COMPUTE(As[buf], Bs[buf])
    if (kt + 1 < nk) STCAT_STEM_STORE(buf ^ 1)
    __syncthreads();
  }
#undef STCAT_STEM_LOAD
#undef STCAT_STEM_STORE
  const int n = n0 + wn * 32 + l31;
  const float sc = p.scale ? p.scale[n] : 1.f;
  const float bi = p.bias ? p.bias[n] : 0.f;
  STCAT_UNROLL
  for (int tm = 0; tm < TM; ++tm) {
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (m < p.M) {
        float val = acc[tm][0][r] * sc + bi;
        if (p.relu) val = fmaxf(val, 0.f);
        p.C[(long)m * p.ldc + n] = val;
      }
    }
  }
}
