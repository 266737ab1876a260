// Plane-format split-bf16 implicit GEMM for the ResNet-101 conv stack (torchvision resnet101 call site
// models/vision_model/backbone.py:115-119, FrozenBN :56-66) — round-2 rebuild of the dominant kernel.
//
// Arithmetic is the bf16x3 contraction of igemm_bs.h (x = hi + lo in bf16; hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate), but the operands ARRIVE split: every activation / gradient / weight
// of the backbone lives in HBM as two bf16 planes (hi = bf16(x), lo = bf16(x - hi); 4 bytes per element like fp32),
// written that way by the producing epilogue.  What that buys in the K loop (measured, tools/proto/gemm_pl_probe.hip,
// profiles/r02_gemm_pl_probe.log):
//   * operands go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane): no staging VGPRs, no fp32->bf16 split
//     VALU, no ds_write pass; padding / stride-lattice / tail rows are the descriptor's out-of-range zero fill;
//   * 256-wide tiles, 8 waves, ONE workgroup per CU: per K-step of 32 a wave issues 48 MFMAs against 24 ds_read_b128
//     and 8 DMA instructions (the 128x128 4-wave kernel: 24 MFMAs against 16 reads + 16 loads + ~100 split VALU);
//   * fragments are double-buffered across k-steps AND K-tiles (the reads of step s+1 are issued among the MFMAs of
//     step s), one barrier per K-tile, the DMA of tile t+2 is issued among the MFMAs that follow the barrier of
//     tile t+1 and is waited for one whole tile later (two tiles in flight: two 64 KB stages);
//   * LDS rows are 64 B (32 bf16) with the 16-byte chunk index XOR-ed by (row >> 2) & 3 — applied to the per-lane
//     SOURCE address of the DMA (its destination is lane-linear) and to the fragment read: conflict-free for the
//     16-lane groups of ds_read_b128.
// Plain-GEMM form of this loop at 8192 x 8192 x 2304: 433-440 TF algorithmic = 1.3 PF issued (52 % of the dense
// bf16 peak; the MFMA-only ceiling of the same wave tile on random data is 588 TF) against 320-330 TF for igemm_bs.h.
//
// Epilogue: accumulators -> wave-private LDS block -> whole 16-byte row segments per lane: FrozenBN scale/bias,
// residual (planes), ReLU, fused ReLU+BN backward mask of the layer below (dgrad), second scaled output, then the
// hi/lo split and 16-byte plane stores (and/or an fp32 store for the consumer outside the backbone).
//
// The weight gradient (igemm_pl_wgrad_kernel) contracts over pixels: both operands are k-major in HBM.  They are
// staged as they lie ([32 pixels][BM or BN channels]) and transposed on the way to the MFMA by ds_read_b64_tr_b16.
#pragma once
#include "igemm.h"

struct PlParams {
  const __bf16* Ah; const __bf16* Al;   // gathered operand planes: NHWC pixels, pixel stride g.ld elements
  const __bf16* Bh; const __bf16* Bl;   // weight planes: row = output column n (stride ldb elements); K-tile (tap, c0) at
                                        // element offset tap * b_tap_stride + c0 inside a row
  __bf16* Ch; __bf16* Cl;               // output planes [M][ldc] (both or neither)
  float* Cf;                            // fp32 output [M][ldc] or null
  const float* scale; const float* bias;
  const __bf16* Rh; const __bf16* Rl;   // residual planes [M][ldr] or null
  const __bf16* Yh; const __bf16* Yl;   // dgrad: planes of y [M][ldc] (output of the layer below): result zeroed where y <= 0 ...
  const unsigned char* Mi;              // the same mask as ONE BIT per element [M][ldc / 8] (bit e of byte j = y[m][8 j + e] > 0),
                                        // written by the forward epilogue of the layer below (Mo): 1/32 of the plane bytes
  unsigned char* Mo;                    // forward: optional bit mask of the output (y > 0) for the backward pass
  const float* mscale;                  // ... and multiplied by that layer's FrozenBN scale (or null)
  __bf16* C2h; __bf16* C2l; const float* c2scale;  // optional second output C * c2scale[n]
  int M, N, K, ldb, ldc, ldr, relu;
  unsigned a_bytes, b_bytes;            // extent of ONE plane of A / B (bytes)
  unsigned b_tap_stride;                // elements
  int par;                              // fwd kernel, stride-2 data gradient on even H, W: rows are enumerated parity class by
                                        // parity class ((y & 1, x & 1), M / 4 pixels each, whole tiles per class) so that a
                                        // tile needs only the filter taps of its class — the lattice zeros (3/4 of the
                                        // gathered rows otherwise) are never multiplied
  int k_chunk;                          // wgrad: pixels of the reduction per grid.z slice (multiple of 32)
  // exact-fp32 form (template parameter F32): the A / B "planes" are the fp32 tensors themselves, addressed in units of
  // half a float (the caller doubles C, ld, ldb, b_tap_stride, K), products on v_mfma_f32_32x32x2_f32; epilogue operands fp32
  const float* Rf;                      // residual [M][ldr]
  const float* Yf;                      // dgrad mask source y [M][ldc]
  float* C2f;                           // second scaled output
  float* Wf;                            // wgrad: fp32 output dW [rows][ldc] (atomics into a zeroed buffer)
  float* Ws;                            // wgrad: optional workspace [slices][rows][ldc]: every reduction slice STORES its partial
                                        // tile there (no atomics) and pl_wgrad_reduce_kernel sums the slices in order into Wf
  const float* wscale;                  // wgrad: optional per-row factor dW[m][:] *= wscale[m] (a FrozenBN scale folded out of dY)
  float acc_mul;                        // the accumulator is multiplied by this before scale / bias (fwd, dgrad) or before
                                        // the atomics (wgrad): 1, or 2^-k undoing the operand scales of mode f16x3p
  int stagger;                          // fwd kernel: phase stagger of the first-round workgroups, in 10 ns ticks per quarter
                                        // (0 = off); set by the launcher for many-round, epilogue-heavy launches
  int debug;                            // timing experiments only (stcat_debug_pl_flags): 1 = no wgrad atomics,
                                        // 2 = epilogue without global loads / stores
  DropParams drop;                      // fwd kernels, round 5: dropout on the epilogue's result (after ReLU), element index
                                        // m * N + n of the site's counter range — a Linear on planes with the FFN's
                                        // `dropout(relu(.))` in its epilogue (dense [M, N] output)
  int radd_div, radd_h, radd_w;         // fwd kernel, round 5: the residual / `add` planes live on the COARSE grid of a stride-
                                        // radd_div conv ([n][radd_h][radd_w][ldr]) and are added at the output pixels on its
                                        // lattice only ((oh, ow) multiples of radd_div; nothing elsewhere): the data gradient
                                        // of a bottleneck's stride-2 downsample conv never has to be scattered into a
                                        // full-resolution tensor that is 3/4 zeros (0 / 1: plain row-for-row residual)
  IgemmGeom g;
};

// Element type of a plane.  The planes are 16-bit containers (`__bf16` in every signature); what the 16 bits MEAN is a
// compile-time property of the kernel: bf16 (8 significand bits, fp32's range) or — mma mode f16x3p, round 4 — IEEE fp16
// (11 significand bits: hi + lo = 22 bits in TWO planes, three products, on v_mfma_f32_32x32x16_f16; fp16's range is what
// the per-class power-of-two scales of that mode are for: weights x 2^wlog, gradients x 2^glog, PlParams::acc_mul undoes them).
template <bool F16> struct PlElt {
  static __device__ __forceinline__ float up(__bf16 v) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v); else return (float)v;
  }
  static __device__ __forceinline__ __bf16 down(float x) {
    if constexpr (F16) {
#ifndef STCAT_EMU
      // The value to convert must be ONE fp32 number in a register: left to itself the compiler folds a preceding
      // multiply into the conversion (v_fma_mix*: fp16 of the UNROUNDED product) for the stored plane while the
      // subtraction that forms the next plane uses the rounded product — near a rounding midpoint the two disagree and
      // the pair is off by one fp16 ulp (measured: the scaled second output of the data gradient, 2^-12 instead of 2^-23).
      asm volatile("" : "+v"(x));
#endif
      return __builtin_bit_cast(__bf16, (_Float16)x);
    } else {
      return (__bf16)x;
    }
  }
};
template <bool F16>
static __device__ __forceinline__ f32x16 stcat_pl_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return STCAT_MFMA_F16_32x32x16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
  else return STCAT_MFMA_BF16_32x32x16(a, b, c);
}

static __device__ __forceinline__ void stcat_split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)v[e];
    h[e] = hh;
    l[e] = (__bf16)(v[e] - (float)hh);
  }
}
// Plane sets.  NP = 2 (mma mode bf16x3p): x ~ p0 + p1, 16 significand bits.  NP = 3 (mma mode bf16x6p): x = p0 + p1 + p2
// EXACTLY (p0 = bf16(x), p1 = bf16(x - p0), p2 = x - p0 - p1: 8 + 8 + 8 significand bits, every residual is exact in
// fp32), so the tensors in HBM carry all of fp32; the contraction keeps the six cross terms down to 2^-16 relative
// (p0 q0, p0 q1, p1 q0, p1 q1, p0 q2, p2 q0) and drops the three of relative size <= 2^-24 — fp32-class products.
// The planes of one tensor lie at EQUAL spacing in one allocation: callers pass the first two pointers, the third
// is l + (l - h).
static __device__ __forceinline__ const __bf16* stcat_plane(const __bf16* h, const __bf16* l, int i) {
  return i == 0 ? h : (i == 1 ? l : l + (l - h));
}
static __device__ __forceinline__ __bf16* stcat_plane(__bf16* h, __bf16* l, int i) {
  return i == 0 ? h : (i == 1 ? l : l + (l - h));
}
template <int NP, bool F16 = false>
static __device__ __forceinline__ void stcat_split8n(const float (&v)[8], bf16x8 (&o)[NP]) {
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) {
    float r = v[e];
    STCAT_UNROLL
    for (int i = 0; i < NP; ++i) {
      const __bf16 q = PlElt<F16>::down(r);
      o[i][e] = q;
      r -= PlElt<F16>::up(q);
    }
  }
}
// value of element e of a plane set held in registers (small pieces first: the sum is exact for NP = 3)
template <int NP, bool F16 = false>
static __device__ __forceinline__ float stcat_join1(const bf16x8 (&o)[NP], int e) {
  float r = PlElt<F16>::up(o[NP - 1][e]);
  STCAT_UNROLL
  for (int i = NP - 2; i >= 0; --i) r += PlElt<F16>::up(o[i][e]);
  return r;
}
// cross terms of the split contraction, smallest first: (index of the A piece, index of the B piece)
template <int NP> struct PlProd;
template <> struct PlProd<2> {
  static constexpr int N = 3;
  static __device__ __forceinline__ constexpr int a(int i) { return i == 0 ? 1 : 0; }
  static __device__ __forceinline__ constexpr int b(int i) { return i == 1 ? 1 : 0; }
};
template <> struct PlProd<3> {
  static constexpr int N = 6;   // (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
  static __device__ __forceinline__ constexpr int a(int i) { return i == 0 ? 2 : ((i == 2 || i == 3) ? 1 : 0); }
  static __device__ __forceinline__ constexpr int b(int i) { return i == 1 ? 2 : ((i == 2 || i == 4) ? 1 : 0); }
};
// float e (0..3) of a 16-byte fragment register
static __device__ __forceinline__ float stcat_f4(const bf16x8& v, int e) {
  const float4 f = __builtin_bit_cast(float4, v);
  return e == 0 ? f.x : (e == 1 ? f.y : (e == 2 ? f.z : f.w));
}
static __device__ __forceinline__ void stcat_join8(const __bf16* hp, const __bf16* lp, float (&v)[8]) {
  const bf16x8 h = *reinterpret_cast<const bf16x8*>(hp), l = *reinterpret_cast<const bf16x8*>(lp);
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) v[e] = (float)h[e] + (float)l[e];
}

#define STCAT_PL_ACC_INIT                                     \
  f32x16 acc[TM][TN];                                         \
  STCAT_UNROLL                                                \
  for (int i = 0; i < TM; ++i) {                              \
    STCAT_UNROLL                                              \
    for (int j = 0; j < TN; ++j) {                            \
      STCAT_UNROLL                                            \
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;        \
    }                                                         \
  }

// the MFMA groups of one k-step (16 reduction terms): 3 (NP = 2) or 6 (NP = 3) cross terms, term outermost so that
// consecutive MFMAs hit different accumulators
#define STCAT_PL_MMA(F)                                                                                 \
  STCAT_UNROLL                                                                                          \
  for (int pr_ = 0; pr_ < PlProd<NP>::N; ++pr_) {                                                       \
    STCAT_UNROLL                                                                                        \
    for (int tm = 0; tm < TM; ++tm) {                                                                   \
      STCAT_UNROLL                                                                                      \
      for (int tn = 0; tn < TN; ++tn)                                                                   \
        acc[tm][tn] = stcat_pl_mfma<F16>(F.a[PlProd<NP>::a(pr_)][tm], F.b[PlProd<NP>::b(pr_)][tn], acc[tm][tn]); \
    }                                                                                                   \
  }

// exact-fp32 form: a 16-byte fragment is FOUR fp32 reduction terms of the lane's row (the lane pair (l31, hi) covers 8
// consecutive k: hi = 0 the first four, hi = 1 the last four — the MFMA's own k index (= hi) just has to pair the same
// term of A and B); four v_mfma_f32_32x32x2_f32 per fragment pair, k-term outermost
#define STCAT_PL_MMA_F32(F)                                                                             \
  STCAT_UNROLL                                                                                          \
  for (int e = 0; e < 4; ++e) {                                                                         \
    STCAT_UNROLL                                                                                        \
    for (int tm = 0; tm < TM; ++tm) {                                                                   \
      STCAT_UNROLL                                                                                      \
      for (int tn = 0; tn < TN; ++tn)                                                                   \
        acc[tm][tn] = STCAT_MFMA_32x32x2(stcat_f4(F.a[0][tm], e), stcat_f4(F.b[0][tn], e), acc[tm][tn]);    \
    }                                                                                                   \
  }

// interleave request for the block [DMA | fragment reads | MFMAs] that follows: NV DMA instructions and ND fragment
// reads spread over NM MFMAs.  Measured (probe V8/V9): with the 8 DMA instructions clustered behind the barrier both
// waves of a SIMD sit in address/M0 set-up while the matrix pipe idles (-12 %).
#define STCAT_PL_INTERLEAVE(NM, ND, NV)                                                                 \
  STCAT_UNROLL                                                                                          \
  for (int i_ = 0; i_ < (NV); ++i_) { STCAT_SCHED_GROUP(0x008, 1); STCAT_SCHED_GROUP(0x020, 1); }       \
  STCAT_UNROLL                                                                                          \
  for (int i_ = 0; i_ < (ND); ++i_) {                                                                   \
    STCAT_SCHED_GROUP(0x008, ((NM) - (NV)) / (ND) > 0 ? ((NM) - (NV)) / (ND) : 1);                      \
    STCAT_SCHED_GROUP(0x100, 1);                                                                        \
  }

// ---------------------------------------------------------------------------------------------------
// forward / data gradient:  C[m][n] = epi( sum_r Agather[m][r] * B[n][r] ),  r = (tap, c), c fastest
// 8 waves as WM x WN, wave tile (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32 x 32.
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool F32 = false, int NP = 2, int NW = 8, bool F16 = false>
__global__ void __launch_bounds__(NW * 64, 2) igemm_pl_fwd_kernel(PlParams p) {
  // NW = 8: one workgroup per CU (two waves per SIMD).  NW = 4 (128 x 64 tile, three planes: 74 KB of LDS): TWO workgroups
  // per CU with the same two waves per SIMD — their phases drift apart, so one's epilogue / first loads (HBM) run under
  // the other's MFMAs: the form for the K <= 512 1x1 convolutions, whose epilogue traffic is as long as their K loop
  static_assert(WM * WN == NW, "NW waves");
  static_assert(!(F32 && NP != 2), "the exact-fp32 form keeps the two-plane LDS layout");
  static_assert(!(F16 && (F32 || NP != 2)), "fp16 planes come in pairs");
  constexpr int BK = 32, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NPL = F32 ? 1 : NP;                              // planes actually staged
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;            // bytes: rows x 64 B
  constexpr int STAGE = NP * (PLANE_A + PLANE_B);                // A planes, then B planes
  static_assert(2 * STAGE <= 160 * 1024, "two stages fit the CU's LDS");
  constexpr int QA = BM / 16, QB = BN / 16;                      // 1-KiB DMA pieces (16 rows) per plane
  constexpr int RQA = (QA + NW - 1) / NW, RQB = (QB + NW - 1) / NW;  // pieces per wave
  constexpr int LDE = TN * 32 + 4;                               // epilogue block: 32 rows x (TN*32) fp32, padded
  constexpr int EPI_WAVE = 32 * LDE * 4;
  static_assert(NW * EPI_WAVE <= 2 * STAGE, "epilogue blocks fit the operand stages");
  STCAT_DYN_SHARED(char, smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wq = STCAT_READFIRSTLANE(wave);
  const int wm = wave / WN, wn = wave % WN;
  const int num_n = p.N / BN;
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);
  const IgemmGeom g = p.g;
#ifndef STCAT_EMU
  if (p.stagger > 0 && blockIdx.x < 256u) {
    // Phase stagger (round 4).  Every tile of a launch costs the same, so the 256 workgroups of the first round start
    // together, run their K loops together (HBM nearly idle) and their epilogues together (HBM saturated) — and so does
    // every later round: matrix time and HBM time ADD although they use different units of the chip.  Delaying the
    // first-round workgroups by 0 / 1 / 2 / 3 quarters of a tile period (ids 8 apart share an XCD, so every XCD holds
    // all four phases) makes some CUs compute while others stream; the order is then kept by the dispatcher, which
    // hands out the next tile when a CU frees up.  One-time cost: 3/8 of a tile period per CU on average.
    const int phase = (blockIdx.x >> 3) & 3;
    if (phase) {
      const long t0 = (long)wall_clock64(), ticks = (long)phase * p.stagger;
      long tt = t0;
      while (tt - t0 < ticks) { __builtin_amdgcn_s_sleep(8); tt = (long)wall_clock64(); }
    }
  }
#endif
  // row space: plain = pixel index m; parity form = (class, index inside the class), whole tiles per class
  const int Mc = p.par ? p.M >> 2 : p.M;                    // rows per class
  const int tpc = (Mc + BM - 1) / BM;                       // tiles per class
  const int mt = v / num_n;
  const int cls = p.par ? mt / tpc : 0, py = cls >> 1, px = cls & 1;
  const int m0 = (mt - cls * tpc) * BM, n0 = (v % num_n) * BN;
  const int OHc = p.par ? g.OH >> 1 : g.OH, OWc = p.par ? g.OW >> 1 : g.OW;
  // filter taps that reach this class: (y + off - kh) even <=> kh = (py + off) mod 2 (mod 2); all taps in the plain form
  const int kstep = p.par ? 2 : 1;
  const int kh0 = p.par ? ((py + g.off) & 1) : 0, kw0 = p.par ? ((px + g.off) & 1) : 0;
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + kstep - 1) / kstep : 0, nkw = kw0 < g.KW ? (g.KW - kw0 + kstep - 1) / kstep : 0;
  const int nk = (p.debug & 16) ? 0 : (p.par ? nkh * nkw * (g.C / BK) : p.K / BK);   // (debug 16: epilogue only)

  // ---- DMA bookkeeping: piece q = wave + NW i covers rows 16 q .. 16 q + 15 of a plane; lane -> row 16 q + (lane >> 2),
  // LDS chunk (lane & 3) which holds SOURCE chunk (lane & 3) ^ ((row >> 2) & 3)
  int a_nb[RQA], a_bh[RQA], a_bw[RQA];
  unsigned a_c16[RQA], b_voff[RQB];
  STCAT_UNROLL
  for (int i = 0; i < RQA; ++i) {
    const int q = wave + NW * i, r = q * 16 + (lane >> 2), m = ((p.debug & 256) ? 0 : m0) + r;   // (debug 256: every tile loads rows 0..BM-1 — L2 hits)
    a_c16[i] = (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 16);
    a_nb[i] = -1; a_bh[i] = 0; a_bw[i] = 0;
    if (q < QA && m < Mc) {
      int nb, oh, ow;
      if (p.par) {
        nb = m / (OHc * OWc);
        const int rem = m - nb * OHc * OWc;
        oh = rem / OWc; ow = rem - oh * OWc;
        oh = 2 * oh + py; ow = 2 * ow + px;
      } else {
        nb = stcat_fastdiv(m, g.mg_ohw, g.sh_ohw);
        const int rem = m - nb * g.OH * g.OW;
        oh = stcat_fastdiv(rem, g.mg_ow, g.sh_ow); ow = rem - oh * g.OW;
      }
      a_nb[i] = nb; a_bh[i] = oh * g.mul + g.off; a_bw[i] = ow * g.mul + g.off;
    }
  }
  STCAT_UNROLL
  for (int i = 0; i < RQB; ++i) {
    const int q = wave + NW * i, r = q * 16 + (lane >> 2);
    b_voff[i] = q < QB ? (unsigned)(((n0 + r) * p.ldb) * 2 + ((lane & 3) ^ ((r >> 2) & 3)) * 16) : STCAT_BUF_OOB;
  }
  const __bf16* Ap[3] = {p.Ah, p.Al, stcat_plane(p.Ah, p.Al, 2)};
  const __bf16* Bp[3] = {p.Bh, p.Bl, stcat_plane(p.Bh, p.Bl, 2)};
  // load cursor: K-tile kl = (tap (kh, kw), channel offset c0); advances one tile per stage_load.  Tiles past the end
  // go through zero-length descriptors (zero fill into a stage nobody reads): the loop body stays branch-free, which
  // keeps the compiler's wait counts exact.
  int kl = 0, l_c0 = 0, l_kh = kh0, l_kw = kw0, l_tap = kh0 * g.KW + kw0;
  const int dmask = g.div - 1, dshift = g.div > 1 ? 31 - __builtin_clz(g.div) : 0;  // div is 1 or a power of two
#define STCAT_PL_STAGE_LOAD(ST)                                                                         \
  {                                                                                                     \
    const bool live_ = (kl < nk) & !(p.debug & 512);   /* (debug 512: zero-fill DMA, no operand traffic) */ \
    stcat_buf_t dA_[NPL], dB_[NPL];                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NPL; ++pi_) {                                                               \
      dA_[pi_] = stcat_make_buf(Ap[pi_], live_ ? p.a_bytes : 0u);                                       \
      dB_[pi_] = stcat_make_buf(Bp[pi_], live_ ? p.b_bytes : 0u);                                       \
    }                                                                                                   \
    const unsigned soA_ = (unsigned)l_c0 * 2u, soB_ = ((unsigned)l_tap * p.b_tap_stride + (unsigned)l_c0) * 2u; \
    char* base_ = smem + (ST) * STAGE + wq * 1024;                                                      \
    STCAT_UNROLL                                                                                        \
    for (int i = 0; i < RQA; ++i) {                                                                     \
      int h_ = a_bh[i] + l_kh * g.sgn, w_ = a_bw[i] + l_kw * g.sgn;                                     \
      bool ok_ = (a_nb[i] >= 0) & (((h_ | w_) & dmask) == 0);                                           \
      h_ >>= dshift; w_ >>= dshift;                                                                     \
      ok_ = ok_ & ((unsigned)h_ < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W);                      \
      const unsigned vo_ = ok_ ? (unsigned)(((a_nb[i] * g.H + h_) * g.W + w_) * g.ld) * 2u + a_c16[i] : STCAT_BUF_OOB; \
      if ((QA % NW == 0) || wq + NW * i < QA) {                                                          \
        STCAT_UNROLL                                                                                    \
        for (int pi_ = 0; pi_ < NPL; ++pi_) stcat_glds16(dA_[pi_], base_ + pi_ * PLANE_A + i * (NW * 1024), vo_, soA_); \
      }                                                                                                 \
    }                                                                                                   \
    STCAT_UNROLL                                                                                        \
    for (int i = 0; i < RQB; ++i) {                                                                     \
      if ((QB % NW == 0) || wq + NW * i < QB) {                                                          \
        STCAT_UNROLL                                                                                    \
        for (int pi_ = 0; pi_ < NPL; ++pi_)                                                             \
          stcat_glds16(dB_[pi_], base_ + NP * PLANE_A + pi_ * PLANE_B + i * (NW * 1024), b_voff[i], soB_);     \
      }                                                                                                 \
    }                                                                                                   \
    /* K order: channel chunk OUTER, filter tap INNER — the 9 taps of a 3x3 filter re-read the same pixels shifted  \
       by one row / column; walked back to back on one 32-channel chunk they find them in L2 (a tile's chunk is       \
       ~36 KB; tap-outer order re-fetched the tile's whole 230 KB input per tap: 9x the distinct bytes, PMC) */        \
    ++kl; l_kw += kstep;                                                                                \
    if (l_kw >= g.KW) {                                                                                 \
      l_kw = kw0; l_kh += kstep;                                                                        \
      if (l_kh >= g.KH) { l_kh = kh0; l_c0 += BK; }                                                     \
    }                                                                                                   \
    l_tap = l_kh * g.KW + l_kw;                                                                         \
  }

  // ---- fragment addressing: lane -> row l31 of its tile, k-step ks -> chunk (2 ks + hi) ^ ((row >> 2) & 3)
  unsigned fa_off[2], fb_off[2];
  STCAT_UNROLL
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + hi, sw = (l31 >> 2) & 3;
    const int lr = (p.debug & 1) ? 0 : l31;     // (debug 1: every lane reads row 0 — LDS broadcast, timing experiment)
    fa_off[ks] = (unsigned)((wm * TM * 32 + lr) * 64 + ((c ^ sw) * 16));
    fb_off[ks] = (unsigned)(NP * PLANE_A + (wn * TN * 32 + lr) * 64 + ((c ^ sw) * 16));
  }
  struct Frag { bf16x8 a[NPL][TM], b[NPL][TN]; };
#define STCAT_PL_READ_FRAG(F, SB, KS)                                                                   \
  STCAT_UNROLL                                                                                          \
  for (int tn = 0; tn < TN; ++tn) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NPL; ++pi_)                                                                 \
      F.b[pi_][tn] = *reinterpret_cast<const bf16x8*>((SB) + fb_off[KS] + pi_ * PLANE_B + tn * 2048);   \
  }                                                                                                     \
  STCAT_UNROLL                                                                                          \
  for (int tm = 0; tm < TM; ++tm) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NPL; ++pi_)                                                                 \
      F.a[pi_][tm] = *reinterpret_cast<const bf16x8*>((SB) + fa_off[KS] + pi_ * PLANE_A + tm * 2048);   \
  }

  // ---- epilogue addressing (declared ahead of the K loop: its first operand block is requested from inside the loop)
  // per wave, one 32-row block of its tile at a time through a private LDS block
  float* ew = reinterpret_cast<float*>(smem + wave * EPI_WAVE);
  constexpr int LPR = TN * 4;          // lanes per output row (8 columns each)
  constexpr int RPP = 64 / LPR;        // rows per pass
  const int erow = lane / LPR, ecol = (lane % LPR) * 8;
  const int n = n0 + wn * TN * 32 + ecol;
  // The epilogue's global operands — residual planes and the ReLU bit mask — of a 32-row block are requested BEFORE
  // the block goes through LDS, and those of block tm + 1 before block tm is processed: the K <= 512 layers are bound
  // by exactly these loads (one HBM round trip per pass otherwise: 4-16 serial round trips per tile).
  constexpr int NPS = 32 / RPP;
  struct Pre { bf16x8 r[NPS][NP]; unsigned bits[NPS]; int m[NPS]; };
  const __bf16* Rp[3] = {p.Rh, p.Rl, stcat_plane(p.Rh, p.Rl, 2)};
  __bf16* Cp[3] = {p.Ch, p.Cl, stcat_plane(p.Ch, p.Cl, 2)};
  __bf16* C2p[3] = {p.C2h, p.C2l, stcat_plane(p.C2h, p.C2l, 2)};
  auto prefetch = [&](Pre& q, int tm) {
    STCAT_UNROLL
    for (int ps = 0; ps < NPS; ++ps) {
      const int mrow = m0 + wm * TM * 32 + tm * 32 + ps * RPP + erow;
      int m = mrow;                      // output row = pixel index
      if (p.par && mrow < Mc) {
        const int nb = mrow / (OHc * OWc), rem = mrow - nb * OHc * OWc, oh = rem / OWc, ow = rem - oh * OWc;
        m = (nb * g.OH + 2 * oh + py) * g.OW + 2 * ow + px;
      }
      q.m[ps] = m;
      if (mrow < Mc && !(p.debug & (2 | 32))) {     // (debug 32: no epilogue loads)
        if (p.Rh) {
          long mr = m;
          if (p.radd_div > 1) {                     // coarse-grid residual: only the lattice pixels have one
            const int nb = stcat_fastdiv(m, g.mg_ohw, g.sh_ohw), rem = m - nb * g.OH * g.OW;
            const int oh = stcat_fastdiv(rem, g.mg_ow, g.sh_ow), ow = rem - oh * g.OW;
            const int dm = p.radd_div - 1, ds = p.radd_div == 2 ? 1 : 2;
            mr = ((oh | ow) & dm) ? -1 : ((long)nb * p.radd_h + (oh >> ds)) * p.radd_w + (ow >> ds);
          }
          STCAT_UNROLL
          for (int pi = 0; pi < NP; ++pi) {
            if (mr >= 0) q.r[ps][pi] = STCAT_LOAD_STREAM(reinterpret_cast<const bf16x8*>(Rp[pi] + mr * p.ldr + n));
            else { STCAT_UNROLL for (int e = 0; e < 8; ++e) q.r[ps][pi][e] = (__bf16)0.f; }
          }
        }
        if (p.Mi) q.bits[ps] = p.Mi[((long)m * p.ldc + n) >> 3];
      }
    }
  };
  STCAT_PL_ACC_INIT
  constexpr int NMMA = (F32 ? 4 : PlProd<NP>::N) * TM * TN, NRD = NPL * (TM + TN);
  constexpr int NDMA = NPL * ((QA >= NW ? RQA : 1) + (QB >= NW ? RQB : 1));
  Frag fa, fb;
  STCAT_PL_STAGE_LOAD(0)
  STCAT_PL_STAGE_LOAD(1)
  STCAT_WAIT_VM0();      // tiles 0 and 1 landed (once per workgroup: a counted wait would differ per wave when a plane
  STCAT_S_BARRIER();     // has fewer than 8 DMA pieces)
  STCAT_SCHED_FENCE();
  STCAT_PL_READ_FRAG(fa, smem, 0)
  // The epilogue's first block of global operands (residual planes / bit mask) is requested TWO K-tiles before the K loop
  // ends: one full HBM round trip per tile (2-3 us of a 25-40 us tile on the K <= 512 layers) leaves the critical path.
  // (three-plane tiles only: the 256 x 256 two-plane tile has no registers to spare; EARLY_PRE is a compile-time switch)
  constexpr bool DB = !(BM == 256 && BN == 256);
  // MEASURED NEUTRAL (round 4, same-box A/B on every 1x1 shape, profiles/r04_plane_gemm_experiments.log: +-1 %): the
  // tile is not waiting for its first epilogue operands.  Kept as an opt-in compile-time switch.
#ifdef STCAT_PL_EARLY_PRE
  constexpr bool EARLY_PRE = DB && NP == 3 && !F32;
#else
  constexpr bool EARLY_PRE = false;
#endif
  Pre pre[2];
  const int kt_pre = nk >= 2 ? nk - 2 : -1;
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + (kt & 1) * STAGE;
    const char* sn = smem + ((kt + 1) & 1) * STAGE;
    if (EARLY_PRE && kt == kt_pre) prefetch(pre[0], 0);
    // P0: k-step 1 fragments are read among the MFMAs of k-step 0
    STCAT_PL_READ_FRAG(fb, sb, 1)
    STCAT_PL_INTERLEAVE(NMMA, NRD, 0)
    if constexpr (F32) { STCAT_PL_MMA_F32(fa) } else { STCAT_PL_MMA(fa) }
    STCAT_SCHED_FENCE();
    // P1: own reads of this stage are back (the MFMAs below need fb anyway), tile kt+1 has landed, and after the
    // barrier every wave is past its reads of this stage: its buffer takes tile kt+2
    STCAT_WAIT_VM0_LGKM0();
    STCAT_S_BARRIER();
    STCAT_SCHED_FENCE();
    STCAT_PL_STAGE_LOAD(kt & 1)
    STCAT_PL_READ_FRAG(fa, sn, 0)
    STCAT_PL_INTERLEAVE(NMMA, NRD, NDMA)
    if constexpr (F32) { STCAT_PL_MMA_F32(fb) } else { STCAT_PL_MMA(fb) }
    STCAT_SCHED_FENCE();
  }
#undef STCAT_PL_STAGE_LOAD
#undef STCAT_PL_READ_FRAG
  STCAT_WAIT_VM0_LGKM0();  // past-the-end DMA (zero fill) has landed too: the stages are reused below
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();

  float sc[8], bi[8], ms[8], s2[8];
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) {
    sc[e] = (p.scale ? p.scale[n + e] : 1.f) * p.acc_mul;      // (acc_mul: 1, or the power of two that undoes mode f16x3p's operand scale)
    bi[e] = p.bias ? p.bias[n + e] : 0.f;
    ms[e] = p.mscale ? p.mscale[n + e] : 1.f;
    s2[e] = p.c2scale ? p.c2scale[n + e] : 1.f;
  }
  // (the 256 x 256 two-plane tile has no registers left for the second set; the three-plane 256 x 128 tile does: 212 VGPRs)
  if (DB && !(EARLY_PRE && kt_pre >= 0)) prefetch(pre[0], 0);
  STCAT_UNROLL
  for (int tm = 0; tm < TM; ++tm) {
    if (!DB) prefetch(pre[0], tm);
    STCAT_UNROLL
    for (int tn = 0; tn < TN; ++tn) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) ew[((r & 3) + 8 * (r >> 2) + 4 * hi) * LDE + tn * 32 + l31] = acc[tm][tn][r];
    }
    if (DB && tm + 1 < TM) prefetch(pre[(tm + 1) & 1], tm + 1);
    const Pre& cur = pre[DB ? (tm & 1) : 0];
    STCAT_WAVE_LDS_FENCE();
    STCAT_UNROLL
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * RPP + erow;
      const int mrow = m0 + wm * TM * 32 + tm * 32 + row;
      const int m = cur.m[ps];
      const float4 v0 = stcat_ld4(&ew[row * LDE + ecol]), v1 = stcat_ld4(&ew[row * LDE + ecol + 4]);
      float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (mrow < Mc && !((p.debug & 2) && x[0] != 12345.f)) {
        STCAT_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = x[e] * sc[e] + bi[e];
        if (p.Rh) {
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] += stcat_join1<NP, F16>(cur.r[ps], e);
        } else if (p.Rf) {          // fp32 residual: the exact-fp32 form, and the Linear layers on the plane kernels
          const float4 r0 = stcat_ld4(p.Rf + (long)m * p.ldr + n), r1 = stcat_ld4(p.Rf + (long)m * p.ldr + n + 4);
          x[0] += r0.x; x[1] += r0.y; x[2] += r0.z; x[3] += r0.w; x[4] += r1.x; x[5] += r1.y; x[6] += r1.z; x[7] += r1.w;
        }
        if (p.relu) {
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (p.drop.thresh) {
          const DropParams dp_ = stcat_drop_resolve(p.drop);
          const unsigned long long i0_ = (unsigned long long)m * (unsigned long long)p.N + (unsigned long long)n;
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] *= stcat_drop_mul(dp_, i0_ + e);
        }
        if (p.Mi) {
          const unsigned bits = cur.bits[ps];
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = ((bits >> e) & 1u) ? x[e] * ms[e] : 0.f;
        } else if (p.Yh) {
          // the sign of y is the sign of its leading plane (the residual planes never flip it; y == 0 <=> p0 == 0)
          const bf16x8 y0 = *reinterpret_cast<const bf16x8*>(p.Yh + (long)m * p.ldc + n);
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = PlElt<F16>::up(y0[e]) > 0.f ? x[e] * ms[e] : 0.f;
        } else if (F32 && p.Yf) {
          const float4 y0 = stcat_ld4(p.Yf + (long)m * p.ldc + n), y1 = stcat_ld4(p.Yf + (long)m * p.ldc + n + 4);
          const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = yy[e] > 0.f ? x[e] * ms[e] : 0.f;
        }
        if (p.Mo) {
          unsigned bits = 0u;
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) bits |= (x[e] > 0.f ? 1u : 0u) << e;
          p.Mo[((long)m * p.ldc + n) >> 3] = (unsigned char)bits;
        }
        if (p.Ch && !((p.debug & 64) && x[0] != 12345.f)) {     // (debug 64: no plane stores)
          bf16x8 o8[NP];
          stcat_split8n<NP, F16>(x, o8);
          STCAT_UNROLL
          for (int pi = 0; pi < NP; ++pi) STCAT_STORE_STREAM(reinterpret_cast<bf16x8*>(Cp[pi] + (long)m * p.ldc + n), o8[pi]);
        }
        if (p.Cf) {
          stcat_st4(p.Cf + (long)m * p.ldc + n, make_float4(x[0], x[1], x[2], x[3]));
          stcat_st4(p.Cf + (long)m * p.ldc + n + 4, make_float4(x[4], x[5], x[6], x[7]));
        }
        if (F32 && p.C2f) {
          stcat_st4(p.C2f + (long)m * p.ldc + n, make_float4(x[0] * s2[0], x[1] * s2[1], x[2] * s2[2], x[3] * s2[3]));
          stcat_st4(p.C2f + (long)m * p.ldc + n + 4, make_float4(x[4] * s2[4], x[5] * s2[5], x[6] * s2[6], x[7] * s2[7]));
        }
        if (p.C2h) {
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] *= s2[e];
          bf16x8 o8[NP];
          stcat_split8n<NP, F16>(x, o8);
          STCAT_UNROLL
          for (int pi = 0; pi < NP; ++pi) *reinterpret_cast<bf16x8*>(C2p[pi] + (long)m * p.ldc + n) = o8[pi];
        }
      }
    }
    STCAT_WAVE_LDS_FENCE();
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient:  dW[co][(tap, ci)] += sum over pixels m of dY[m][co] * X[pix(m, tap)][ci]
//   rows = co (BM), columns = (tap, ci) (BN, inside one tap: C % BN == 0), reduction = pixels, split over grid.z.
// Both operands are k-major in HBM (a pixel is a row of channels): they are staged as [32 pixels][BM | BN channels]
// images and the MFMA fragments (8 consecutive k per lane) come out of ds_read_b64_tr_b16.  A k-row is BM*2 (BN*2)
// bytes; its 64-byte segments are XOR-ed with (k & 3) inside each 256-byte group, so the four k-rows that one
// transposing read touches sit in four different bank quarters.
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NP = 2, bool F16 = false>
__global__ void __launch_bounds__(512) igemm_pl_wgrad_kernel(PlParams p) {
  static_assert(WM * WN == 8, "8 waves");
  static_assert(BM % 128 == 0 && BN % 128 == 0, "a k-row is a whole number of 256-byte groups");
  constexpr int BK = 32, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int ROWA = BM * 2, ROWB = BN * 2;                    // bytes per k-row
  constexpr int PLANE_A = BK * ROWA, PLANE_B = BK * ROWB;
  constexpr int STAGE = NP * (PLANE_A + PLANE_B);
  static_assert(2 * STAGE <= 160 * 1024, "two stages fit the CU's LDS");
  constexpr int QA = PLANE_A / 1024, QB = PLANE_B / 1024;        // 1-KiB DMA pieces per plane
  constexpr int RQA = (QA + 7) / 8, RQB = (QB + 7) / 8;
  constexpr int LA = ROWA / 16, LB = ROWB / 16;                  // lanes (16-byte chunks) per k-row
  STCAT_DYN_SHARED(char, smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, hi = lane >> 5;
  const int wq = STCAT_READFIRSTLANE(wave);
  const int wm = wave / WN, wn = wave % WN;
  const int num_n = p.N / BN;
  // XCD-aware order over (reduction slice, tile): workgroups are dealt to the 8 XCDs round-robin in launch order
  // (x fastest, then z); remapped so that each XCD owns a contiguous run of slices and, inside it, all tiles of a slice
  // run together — the tiles of one slice (the 9 taps of a 3x3 filter x the Cout / Cin blocks) re-read the SAME dY
  // pixels and overlapping X pixels, which then come from that XCD's L2 instead of HBM (PMC before: 538 MB read per
  // launch against ~100 MB of distinct operands)
  const int lin = stcat_xcd_remap(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x * gridDim.z);
  const int zsl = lin / gridDim.x, v = lin - zsl * gridDim.x;
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;        // m0: first co, n0: first (tap, ci) column
  const IgemmGeom g = p.g;
  const int tap = n0 / g.C, ci0 = n0 - tap * g.C;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int red0 = zsl * p.k_chunk;
  const int red1 = min(p.K, red0 + p.k_chunk);
  const int nk = (red1 - red0 + BK - 1) / BK;
  if (nk <= 0 && !p.Ws) return;          // (with a workspace every slice stores its tile: an empty slice stores zeros)
  const int ohw = g.OH * g.OW;

  // DMA: piece q = wave + 8 i of a plane covers bytes [1024 q, 1024 q + 1024) of the [32][ROW] image: k-row
  // kr = (1024 q + 16 lane) / ROW, physical chunk cp inside the row; it holds logical chunk cp ^ ((kr & 3) << 2)
  int a_kr[RQA], b_kr[RQB];
  unsigned a_col[RQA], b_col[RQB];
  STCAT_UNROLL
  for (int i = 0; i < RQA; ++i) {
    const int q = wave + 8 * i, byte = q * 1024 + lane * 16, kr = byte / ROWA, cp = (byte % ROWA) / 16;
    a_kr[i] = q < QA ? kr : -1;
    a_col[i] = (unsigned)((m0 * 2) + ((cp ^ ((kr & 3) << 2)) * 16));
  }
  STCAT_UNROLL
  for (int i = 0; i < RQB; ++i) {
    const int q = wave + 8 * i, byte = q * 1024 + lane * 16, kr = byte / ROWB, cp = (byte % ROWB) / 16;
    b_kr[i] = q < QB ? kr : -1;
    b_col[i] = (unsigned)((ci0 * 2) + ((cp ^ ((kr & 3) << 2)) * 16));
  }
  stcat_buf_t dA[NP], dB[NP];
  STCAT_UNROLL
  for (int pi = 0; pi < NP; ++pi) {
    dA[pi] = stcat_make_buf(stcat_plane(p.Ah, p.Al, pi), p.a_bytes);
    dB[pi] = stcat_make_buf(stcat_plane(p.Bh, p.Bl, pi), p.b_bytes);
  }
  int kl = 0;
#define STCAT_PLW_STAGE_LOAD(ST)                                                                        \
  {                                                                                                     \
    const int mb_ = red0 + kl * BK;                                                                     \
    char* base_ = smem + (ST) * STAGE + wq * 1024;                                                      \
    STCAT_UNROLL                                                                                        \
    for (int i = 0; i < RQA; ++i) {                                                                     \
      const int m_ = mb_ + a_kr[i];                                                                     \
      const unsigned vo_ = (a_kr[i] >= 0 && m_ < red1) ? (unsigned)(m_ * p.ldb) * 2u + a_col[i] : STCAT_BUF_OOB; \
      if ((QA % 8 == 0) || wq + 8 * i < QA) {                                                          \
        STCAT_UNROLL                                                                                    \
        for (int pi_ = 0; pi_ < NP; ++pi_) stcat_glds16(dA[pi_], base_ + pi_ * PLANE_A + i * 8192, vo_, 0u); \
      }                                                                                                 \
    }                                                                                                   \
    STCAT_UNROLL                                                                                        \
    for (int i = 0; i < RQB; ++i) {                                                                     \
      const int m_ = mb_ + b_kr[i];                                                                     \
      const int nb_ = stcat_fastdiv(m_, g.mg_ohw, g.sh_ohw), rem_ = m_ - nb_ * ohw;                     \
      const int oh_ = stcat_fastdiv(rem_, g.mg_ow, g.sh_ow), ow_ = rem_ - oh_ * g.OW;                   \
      const int h_ = oh_ * g.mul + g.off + kh, w_ = ow_ * g.mul + g.off + kw;                           \
      const bool ok_ = (b_kr[i] >= 0) & (m_ < red1) & ((unsigned)h_ < (unsigned)g.H) & ((unsigned)w_ < (unsigned)g.W); \
      const unsigned vo_ = ok_ ? (unsigned)(((nb_ * g.H + h_) * g.W + w_) * g.ld) * 2u + b_col[i] : STCAT_BUF_OOB; \
      if ((QB % 8 == 0) || wq + 8 * i < QB) {                                                          \
        STCAT_UNROLL                                                                                    \
        for (int pi_ = 0; pi_ < NP; ++pi_)                                                              \
          stcat_glds16(dB[pi_], base_ + NP * PLANE_A + pi_ * PLANE_B + i * 8192, vo_, 0u);              \
      }                                                                                                 \
    }                                                                                                   \
    ++kl;                                                                                               \
  }

  // fragment addressing (transposing reads): lane -> 16-lane group gq = (lane >> 4) & 1: channels 16 gq .. + 15 of its
  // 32-wide tile, k half hi; inside the group lane pl addresses k-row (pl >> 2), channels 4 (pl & 3) .. + 3 (8 bytes).
  // k = 16 ks + 8 hi + 4 j + (pl >> 2), j = 0, 1: two reads give the 8 consecutive k of the fragment.  k & 3 = pl >> 2
  // for every (ks, j), so the swizzle term is a lane constant per tile: tile T of the wave = 64-byte segment
  // (wave's first segment + T) ^ (pl >> 2).
  const int pl = lane & 15, gq = (lane >> 4) & 1, kx = pl >> 2;
  const unsigned ta_lane = (unsigned)((hi * 8 + kx) * ROWA + gq * 32 + (pl & 3) * 8);
  const unsigned tb_lane = (unsigned)(NP * PLANE_A + (hi * 8 + kx) * ROWB + gq * 32 + (pl & 3) * 8);
  unsigned ta_seg[TM], tb_seg[TN];
  STCAT_UNROLL
  for (int tm = 0; tm < TM; ++tm) ta_seg[tm] = (unsigned)(((wm * TM + tm) ^ kx) << 6);
  STCAT_UNROLL
  for (int tn = 0; tn < TN; ++tn) tb_seg[tn] = (unsigned)(((wn * TN + tn) ^ kx) << 6);
  struct Frag { bf16x8 a[NP][TM], b[NP][TN]; };
#define STCAT_PLW_TR8(DST, ADDR, ROW, KS)                                                               \
  {                                                                                                     \
    const bf16x4 lo4_ = stcat_lds_tr4(reinterpret_cast<const __bf16*>((ADDR) + ((KS) * 16) * (ROW)));   \
    const bf16x4 hi4_ = stcat_lds_tr4(reinterpret_cast<const __bf16*>((ADDR) + ((KS) * 16 + 4) * (ROW))); \
    STCAT_UNROLL                                                                                        \
    for (int e_ = 0; e_ < 4; ++e_) { DST[e_] = lo4_[e_]; DST[4 + e_] = hi4_[e_]; }                      \
  }
#define STCAT_PLW_READ_FRAG(F, SB, KS)                                                                  \
  STCAT_UNROLL                                                                                          \
  for (int tn = 0; tn < TN; ++tn) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_)                                                                  \
      STCAT_PLW_TR8(F.b[pi_][tn], (SB) + tb_lane + tb_seg[tn] + pi_ * PLANE_B, ROWB, KS)                \
  }                                                                                                     \
  STCAT_UNROLL                                                                                          \
  for (int tm = 0; tm < TM; ++tm) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_)                                                                  \
      STCAT_PLW_TR8(F.a[pi_][tm], (SB) + ta_lane + ta_seg[tm] + pi_ * PLANE_A, ROWA, KS)                \
  }

  STCAT_PL_ACC_INIT
  constexpr int NMMA = PlProd<NP>::N * TM * TN, NRD = 2 * NP * (TM + TN);
  constexpr int NDMA = NP * ((QA >= 8 ? RQA : 1) + (QB >= 8 ? RQB : 1));
  Frag fa, fb;
  STCAT_PLW_STAGE_LOAD(0)
  STCAT_PLW_STAGE_LOAD(1)
  STCAT_WAIT_VM0();
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();
  STCAT_PLW_READ_FRAG(fa, smem, 0)
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + (kt & 1) * STAGE;
    const char* sn = smem + ((kt + 1) & 1) * STAGE;
    STCAT_PLW_READ_FRAG(fb, sb, 1)
    STCAT_PL_INTERLEAVE(NMMA, NRD, 0)
    STCAT_PL_MMA(fa)
    STCAT_SCHED_FENCE();
    STCAT_WAIT_VM0_LGKM0();
    STCAT_S_BARRIER();
    STCAT_SCHED_FENCE();
    STCAT_PLW_STAGE_LOAD(kt & 1)
    STCAT_PLW_READ_FRAG(fa, sn, 0)
    STCAT_PL_INTERLEAVE(NMMA, NRD, NDMA)
    STCAT_PL_MMA(fb)
    STCAT_SCHED_FENCE();
  }
#undef STCAT_PLW_STAGE_LOAD
#undef STCAT_PLW_READ_FRAG
#undef STCAT_PLW_TR8
  STCAT_WAIT_VM0_LGKM0();
  const int l31 = lane & 31;
  STCAT_UNROLL
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * TN * 32 + tn * 32 + l31;
    STCAT_UNROLL
    for (int tm = 0; tm < TM; ++tm) {
      STCAT_UNROLL
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v_ = (p.wscale ? acc[tm][tn][r] * p.wscale[m] : acc[tm][tn][r]) * p.acc_mul;
        if (p.Ws) p.Ws[((long)zsl * p.M + m) * p.ldc + n] = v_;
        else if (!((p.debug & 1) && v_ != 12345.f)) atomicAdd(&p.Wf[(long)m * p.ldc + n], v_);
      }
    }
  }
}

// dW[i] += sum over the reduction slices, IN SLICE ORDER, of the partial tiles the weight-gradient workgroups stored (round 5):
// no atomics — the sum's order is fixed, so the gradient is bit-identical run to run — and the 8.4 M contended atomic adds of
// a layer3 1x1 launch (8-14 % of its time) become 34 MB of streaming stores + this pass
__global__ void __launch_bounds__(256) pl_wgrad_reduce_kernel(const float* ws, float* out, long n4, int slices, long stride) {
  // 64 float4 outputs per block x 4 lanes of slices (slice s goes to lane s & 3, each lane sums ITS slices in rising order,
  // the four lane sums are added in a fixed order): 4 x 4 loads in flight per output — the one-thread-per-output form walked
  // the slices one load at a time (20 us for 34 MB)
  __shared__ float4 red[4][64];
  const int o = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float* src = ws + i * 4;
    int s_ = gq;
    for (; s_ + 12 < slices; s_ += 16) {
      const float4 v0 = stcat_ld4(src + (long)s_ * stride), v1 = stcat_ld4(src + (long)(s_ + 4) * stride);
      const float4 v2 = stcat_ld4(src + (long)(s_ + 8) * stride), v3 = stcat_ld4(src + (long)(s_ + 12) * stride);
      a.x = (((a.x + v0.x) + v1.x) + v2.x) + v3.x; a.y = (((a.y + v0.y) + v1.y) + v2.y) + v3.y;
      a.z = (((a.z + v0.z) + v1.z) + v2.z) + v3.z; a.w = (((a.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; s_ < slices; s_ += 4) {
      const float4 v = stcat_ld4(src + (long)s_ * stride);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[gq][o] = a;
  __syncthreads();
  if (gq == 0 && i < n4) {
    const float4 r0 = red[0][o], r1 = red[1][o], r2 = red[2][o], r3 = red[3][o];
    float4 d = stcat_ld4(out + i * 4);
    d.x += (r0.x + r1.x) + (r2.x + r3.x); d.y += (r0.y + r1.y) + (r2.y + r3.y);
    d.z += (r0.z + r1.z) + (r2.z + r3.z); d.w += (r0.w + r1.w) + (r2.w + r3.w);
    stcat_st4(out + i * 4, d);
  }
}

// ---------------------------------------------------------------------------------------------------
// plane producers / consumers outside the GEMMs
// ---------------------------------------------------------------------------------------------------
// split 8 values into np (2 or 3) planes and store them at element offset `at` of each plane
static __device__ __forceinline__ void stcat_store_planes(const float (&v)[8], __bf16* h, __bf16* l, long at, int np) {
  if (np >> 8) {                           // fp16 hi + lo (mode f16x3p): np = 2 | 0x100
    bf16x8 o[2];
    stcat_split8n<2, true>(v, o);
    *reinterpret_cast<bf16x8*>(h + at) = o[0];
    *reinterpret_cast<bf16x8*>(l + at) = o[1];
    return;
  }
  if (np == 3) {
    bf16x8 o[3];
    stcat_split8n<3>(v, o);
    *reinterpret_cast<bf16x8*>(h + at) = o[0];
    *reinterpret_cast<bf16x8*>(l + at) = o[1];
    *reinterpret_cast<bf16x8*>(stcat_plane(h, l, 2) + at) = o[2];
  } else {
    bf16x8 h8, l8;
    stcat_split8(v, h8, l8);
    *reinterpret_cast<bf16x8*>(h + at) = h8;
    *reinterpret_cast<bf16x8*>(l + at) = l8;
  }
}
static __device__ __forceinline__ void stcat_load_planes(const __bf16* h, const __bf16* l, long at, int np, float (&v)[8]) {
  if (np >> 8) {
    const bf16x8 o[2] = {*reinterpret_cast<const bf16x8*>(h + at), *reinterpret_cast<const bf16x8*>(l + at)};
    STCAT_UNROLL
    for (int e = 0; e < 8; ++e) v[e] = stcat_join1<2, true>(o, e);
    return;
  }
  stcat_join8(h + at, l + at, v);
  if (np == 3) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(stcat_plane(h, l, 2) + at);
    STCAT_UNROLL
    for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
  }
}
// 3x3 stride-2 pad-1 max-pool, NHWC fp32 in (stem output) -> planes out (input of layer1)
// out[n] += sum_m (hi + mid + lo)[m][n]: the bias gradient of a Linear whose upstream gradient only exists as planes
// (encoder FFN linear1, round 5).  Block = 32 column groups of 8 x 8 row lanes; grid (row chunks, N / 256).
__global__ void __launch_bounds__(256) pl_colsum_kernel(const __bf16* h, const __bf16* l, float* out, int M, int N, int np,
                                                        int rows_per_block) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = blockIdx.y * 256 + cg * 8;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int m = m0 + rl; m < m1; m += 8) {
      STCAT_UNROLL
      for (int pi = 0; pi < 3; ++pi) {
        if (pi < np) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(stcat_plane(h, l, pi) + (long)m * N + n);
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
      }
    }
  }
  STCAT_UNROLL
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.y * 256 + c < N) {
    float v = 0.f;
    STCAT_UNROLL
    for (int r = 0; r < 8; ++r) v += red[r][c];
    atomicAdd(out + blockIdx.y * 256 + c, v);
  }
}

__global__ void __launch_bounds__(256) maxpool3x3s2_pl_kernel(const float* x, __bf16* yh, __bf16* yl, int n, int H, int W,
                                                             int C, int OH, int OW, int np) {
  const int c8n = C / 8;
  const long total = (long)n * OH * OW * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8n);
    const long pix = i / c8n;
    const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH), f = (int)(pix / ((long)OW * OH));
    float m[8];
    STCAT_UNROLL
    for (int e = 0; e < 8; ++e) m[e] = STCAT_NEG_INF;
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = oh * 2 - 1 + kh;
      if (hh < 0 || hh >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = ow * 2 - 1 + kw;
        if (ww < 0 || ww >= W) continue;
        const float* src = x + (((long)f * H + hh) * W + ww) * C + c8 * 8;
        const float4 a = stcat_ld4(src), b = stcat_ld4(src + 4);
        m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
        m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
      }
    }
    stcat_store_planes(m, yh, yl, i * 8, np);
  }
}

// element-wise plane kernels over n8 groups of 8 elements; channel of group i = (i * 8) % C
//   mode 0: split      out = x (fp32 -> planes)
//   mode 1: act_bwd    dz = dy * [y > 0] (dy, y fp32; relu = 0: no mask) -> dz planes (R), g = dz * scale[c] planes (G)
//   mode 2: scale      G = X(planes) * scale[c]
//   mode 3: join       f = X(planes) -> fp32
struct PlEwParams {
  const float* xf; const float* yf; const float* scale;
  const __bf16* Xh; const __bf16* Xl;
  __bf16* Gh; __bf16* Gl; __bf16* Rh; __bf16* Rl;
  float* of;
  long n8; int C; int mode; int relu;
  int np;   // planes per set: 2 or 3 (| 0x100: fp16 elements)
  float gmul;   // mode 1 (gradient entering the plane domain): factor on dz — 1, or mode f16x3p's loss scale 2^glog
};
__global__ void __launch_bounds__(256) planes_ew_kernel(PlEwParams p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    if (p.mode == 0 || p.mode == 1 || p.mode == 4) {
      const float4 a = stcat_ld4(p.xf + i * 8), b = stcat_ld4(p.xf + i * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      if (p.mode == 4) {        // split of a SUM (q = k = src + pos, modal_encoder.py:234): the fp32 sum is kept for the weight gradient
        const float4 c = stcat_ld4(p.yf + i * 8), d = stcat_ld4(p.yf + i * 8 + 4);
        v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; v[4] += d.x; v[5] += d.y; v[6] += d.z; v[7] += d.w;
        if (p.of) {
          stcat_st4(p.of + i * 8, make_float4(v[0], v[1], v[2], v[3]));
          stcat_st4(p.of + i * 8 + 4, make_float4(v[4], v[5], v[6], v[7]));
        }
      }
    } else {
      stcat_load_planes(p.Xh, p.Xl, i * 8, p.np, v);
    }
    if (p.mode == 3) {
      stcat_st4(p.of + i * 8, make_float4(v[0], v[1], v[2], v[3]));
      stcat_st4(p.of + i * 8 + 4, make_float4(v[4], v[5], v[6], v[7]));
      continue;
    }
    if (p.mode == 1 && p.relu) {
      const float4 a = stcat_ld4(p.yf + i * 8), b = stcat_ld4(p.yf + i * 8 + 4);
      const float yy[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      STCAT_UNROLL
      for (int e = 0; e < 8; ++e) v[e] = yy[e] > 0.f ? v[e] : 0.f;
    }
    if (p.mode == 1 && p.gmul != 1.f) {
      STCAT_UNROLL
      for (int e = 0; e < 8; ++e) v[e] *= p.gmul;
    }
    if (p.Rh) stcat_store_planes(v, p.Rh, p.Rl, i * 8, p.np);
    if (p.Gh) {
      if (p.scale) {
        const int c = (int)((i * 8) % p.C);
        STCAT_UNROLL
        for (int e = 0; e < 8; ++e) v[e] *= p.scale[c + e];
      }
      stcat_store_planes(v, p.Gh, p.Gl, i * 8, p.np);
    }
  }
}

// Weight planes for MANY conv weights in one launch: entry e describes W fp32 OHWI [Cout][taps][Cin] and writes
//   fwd planes  Wh/Wl [Cout][taps][Cin]   (B operand of the forward GEMM: row = co),
//   tr  planes  Th/Tl [taps][Cin][Cout]   (B operand of the data gradient: row = ci inside a tap)  — optional.
// A block handles a 32 (co) x 32 (ci) patch of one tap.
struct WplEntry {
  const float* w;
  __bf16* wh; __bf16* wl; __bf16* th; __bf16* tl;
  const float* tscale;   // optional [Cout]: the transposed planes hold w * tscale[co] (a FrozenBN scale folded into the
                         // data-gradient operand, so the upstream gradient dz * scale never has to be materialised)
  int Cout, taps, Cin;
  int blk0, nbx, nby;
  int pad_;
};
static __device__ __forceinline__ void stcat_store_planes1(float x, __bf16* h, __bf16* l, long idx, int np) {
  if (np >> 8) {
    const __bf16 q0 = PlElt<true>::down(x);
    h[idx] = q0;
    l[idx] = PlElt<true>::down(x - PlElt<true>::up(q0));
    return;
  }
  const __bf16 q0 = (__bf16)x;
  h[idx] = q0;
  const float r1 = x - (float)q0;
  const __bf16 q1 = (__bf16)r1;
  l[idx] = q1;
  if (np == 3) stcat_plane(h, l, 2)[idx] = (__bf16)(r1 - (float)q1);
}
__global__ void __launch_bounds__(256) weight_planes_multi_kernel(const WplEntry* tab, int n, int np, float wmul) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WplEntry e = tab[lo];
  const int rel = blockIdx.x - e.blk0;
  const int bx = rel % e.nbx, by = (rel / e.nbx) % e.nby, tap = rel / (e.nbx * e.nby);
  const int ci0 = bx * 32, co0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float x = 0.f;
    if (co < e.Cout && ci < e.Cin) {
      const long idx = ((long)co * e.taps + tap) * e.Cin + ci;
      x = e.w[idx] * wmul;                 // (wmul: 1, or mode f16x3p's weight scale 2^wlog)
      stcat_store_planes1(x, e.wh, e.wl, idx, np);
    }
    tile[r][tx] = (e.tscale && co < e.Cout) ? x * e.tscale[co] : x;
  }
  if (!e.th) return;
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < e.Cin && co < e.Cout) {
      const float x = tile[tx][r];
      const long idx = ((long)tap * e.Cin + ci) * e.Cout + co;
      stcat_store_planes1(x, e.th, e.tl, idx, np);
    }
  }
}
