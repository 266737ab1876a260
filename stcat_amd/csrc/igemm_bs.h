// Split-bf16 implicit GEMM on v_mfma_f32_32x32x16_bf16 (2.5 PF dense pipe on gfx950).
//
// Same contractions, geometry and epilogues as igemm.h, but every fp32 operand value x is split while it is
// staged into LDS:  x = p0 + p1 (+ p2),  p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1),
// and the product a*b is accumulated in fp32 from the significant cross terms:
//     NS = 2 ("bf16x3"):  a0*b0 + a0*b1 + a1*b0                      (~2^-16 relative per product)
//     NS = 3 ("bf16x6"):  + a0*b2 + a1*b1 + a2*b0                    (~2^-24: fp32-class)
// 3 (6) MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles per 16 reduction terms: 5.3x (2.7x) less matrix
// pipe time at fp32-input / fp32-output semantics.  Activations and weights stay fp32 in HBM.
//
// Tiling: 256 threads = 2x2 waves (two workgroups per CU) or 512 threads = 4x2 waves on a 256x128 tile (ONE workgroup
// per CU; forward / weight gradient), BMxBN block tile, K step 32.  LDS holds one plane per split piece,
// rows of 32 bf16 padded to 40 (80 B): an MFMA fragment is one 16-byte ds_read_b128 per lane and the 80-byte
// row stride is conflict-free for its 16-lane groups.  Two staging paths:
//   R : operand rows are contiguous along the reduction (NHWC pixels x channels, weights [N][K]) —
//       float4 -> split -> one ds_write_b64 per plane;
//   O : operand is reduction-major (dY^T / X^T in wgrad, W in dgrad) — each thread loads a 4(k) x 4(rows)
//       block with four coalesced float4 loads, transposes it in registers, then ds_write_b64 per row.
// Double-buffered LDS, next tile prefetched into registers during the MFMA phase, one barrier per K step.
#pragma once
#include "igemm.h"

#define STCAT_BS_LDK 40  // bf16 elements per LDS row (32 + 8 pad)

template <int NS>
static __device__ __forceinline__ void stcat_bs_split_store(__bf16* dst, int plane_elems, float x0, float x1, float x2,
                                                           float x3) {
  float r[4] = {x0, x1, x2, x3};
#ifdef STCAT_EXPERIMENT_NOSPLIT
  {  // timing experiment only: truncate, same value in every plane (WRONG results, ~6x fewer VALU ops)
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    u16x4 pk;
    STCAT_UNROLL
    for (int e = 0; e < 4; ++e) pk[e] = (unsigned short)(__float_as_uint(r[e]) >> 16);
    STCAT_UNROLL
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u16x4*>(dst + s * plane_elems) = pk;
    return;
  }
#endif
  STCAT_UNROLL
  for (int s = 0; s < NS; ++s) {
    bf16x4 pk;
    STCAT_UNROLL
    for (int e = 0; e < 4; ++e) {
      const __bf16 h = (__bf16)r[e];
      pk[e] = h;
      r[e] -= (float)h;
    }
    *reinterpret_cast<bf16x4*>(dst + s * plane_elems) = pk;
  }
}

// all MFMAs of one 16-wide k-step for a TMxTN wave tile
#define STCAT_BS_COMPUTE(AS, BS)                                                                        \
  STCAT_UNROLL                                                                                          \
  for (int ks = 0; ks < 2; ++ks) {                                                                      \
    bf16x8 a_[TM][NS], b_[TN][NS];                                                                      \
    STCAT_UNROLL                                                                                        \
    for (int s = 0; s < NS; ++s) {                                                                      \
      STCAT_UNROLL                                                                                      \
      for (int tm = 0; tm < TM; ++tm)                                                                   \
        a_[tm][s] = *reinterpret_cast<const bf16x8*>(&(AS)[s * (BM * LDK) + (wm * TM * 32 + tm * 32 + l31) * LDK + ks * 16 + hi * 8]); \
      STCAT_UNROLL                                                                                      \
      for (int tn = 0; tn < TN; ++tn)                                                                   \
        b_[tn][s] = *reinterpret_cast<const bf16x8*>(&(BS)[s * (BN * LDK) + (wn * TN * 32 + tn * 32 + l31) * LDK + ks * 16 + hi * 8]); \
    }                                                                                                   \
    /* split terms outermost: consecutive MFMAs write different accumulators (no dependent-issue stalls) */ \
    STCAT_UNROLL                                                                                        \
    for (int sa = NS - 1; sa >= 0; --sa) {                                                              \
      STCAT_UNROLL                                                                                      \
      for (int sb = NS - 1 - sa; sb >= 0; --sb) {                                                       \
        STCAT_UNROLL                                                                                    \
        for (int tm = 0; tm < TM; ++tm) {                                                               \
          STCAT_UNROLL                                                                                  \
          for (int tn = 0; tn < TN; ++tn)                                                               \
            acc[tm][tn] = STCAT_MFMA_BF16_32x32x16(a_[tm][sa], b_[tn][sb], acc[tm][tn]);                \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
  }

#define STCAT_BS_ACC_INIT                                     \
  f32x16 acc[TM][TN];                                         \
  STCAT_UNROLL                                                \
  for (int i = 0; i < TM; ++i) {                              \
    STCAT_UNROLL                                              \
    for (int j = 0; j < TN; ++j) {                            \
      STCAT_UNROLL                                            \
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;        \
    }                                                         \
  }

// ---- R-type staging of a [ROWS x 32] tile: thread -> k4 = t&7, rows trow + 32*j.  trow permutes t>>3 inside
// each group of 8 so that the two rows written by one 16-lane ds_write group are 4 rows (320 B) apart:
// with 80-byte rows that puts them on disjoint halves of the 32 write banks (adjacent rows overlap by 4).
#define STCAT_BS_STORE_R(DST, ROWS, REGS)                                                                \
  STCAT_UNROLL                                                                                           \
  for (int j = 0; j < (ROWS) / RP; ++j)                                                                  \
    stcat_bs_split_store<NS>(&(DST)[(trow + RP * j) * LDK + (t & 7) * 4]    , (ROWS) * LDK, (REGS)[j].x, \
                             (REGS)[j].y, (REGS)[j].z, (REGS)[j].w);

// ---- O-type staging of a [32(k) x ROWS] tile: block i = t + NTHR*j -> kgrp = i & 7 (4 k's), rowgrp = i >> 3 (4 rows).
// A 16-lane group then writes 8 x 8 B = one contiguous 64-B row segment for each of two row groups 320 B apart
// (conflict-free), and its global reads are 8 rows x 128 contiguous bytes (full lines).
#define STCAT_BS_STORE_O(DST, ROWS, REGS)                                                                \
  STCAT_UNROLL                                                                                           \
  for (int j = 0; j < ((ROWS) * 2 + NTHR - 1) / NTHR; ++j) {                                             \
    const int i = t + NTHR * j;                                                                          \
    if (i < (ROWS) * 2) {                                                                                \
      const int kgrp = i & 7, rowgrp = i >> 3;                                      \
      __bf16* d = &(DST)[(rowgrp * 4) * LDK + kgrp * 4];                                                 \
      stcat_bs_split_store<NS>(d, (ROWS) * LDK, (REGS)[j][0].x, (REGS)[j][1].x, (REGS)[j][2].x, (REGS)[j][3].x);           \
      stcat_bs_split_store<NS>(d + LDK, (ROWS) * LDK, (REGS)[j][0].y, (REGS)[j][1].y, (REGS)[j][2].y, (REGS)[j][3].y);     \
      stcat_bs_split_store<NS>(d + 2 * LDK, (ROWS) * LDK, (REGS)[j][0].z, (REGS)[j][1].z, (REGS)[j][2].z, (REGS)[j][3].z); \
      stcat_bs_split_store<NS>(d + 3 * LDK, (ROWS) * LDK, (REGS)[j][0].w, (REGS)[j][1].w, (REGS)[j][2].w, (REGS)[j][3].w); \
    }                                                                                                    \
  }

// ---- gathered A operand (fwd / dgrad): byte offset of each staged row for the CURRENT filter tap, or
// STCAT_BUF_OOB (hardware zero-fill).  Recomputed only when the tap changes; the channel offset inside the
// tap rides in the scalar soffset of the buffer load.
#define STCAT_BS_GATHER_DECL(ROWS)                                                                       \
  int a_nb[(ROWS) / RP], a_bh[(ROWS) / RP], a_bw[(ROWS) / RP];                                           \
  unsigned a_off[(ROWS) / RP];                                                                           \
  int cur_tap = -1;                                                                                      \
  STCAT_UNROLL                                                                                           \
  for (int j = 0; j < (ROWS) / RP; ++j) {                                                                \
    const int m = m0 + trow + RP * j;                                                                    \
    a_off[j] = STCAT_BUF_OOB;                                                                            \
    if (m < p.M) {                                                                                       \
      const int ohw = g.OH * g.OW;                                                                       \
      const int nb = m / ohw, rem = m - nb * ohw, oh = rem / g.OW, ow = rem - oh * g.OW;                 \
      a_nb[j] = nb; a_bh[j] = oh * g.mul + g.off; a_bw[j] = ow * g.mul + g.off;                          \
    } else {                                                                                             \
      a_nb[j] = -1; a_bh[j] = 0; a_bw[j] = 0;                                                            \
    }                                                                                                    \
  }

#define STCAT_BS_LOAD_A_GATHER(KT, ROWS, SET)                                                            \
  {                                                                                                      \
    const int r0 = (KT) * BK, tap = r0 / g.C, c0 = r0 - tap * g.C;                                       \
    if (tap != cur_tap) {                                                                                \
      cur_tap = tap;                                                                                     \
      const int kh = tap / g.KW, kw = tap - kh * g.KW;                                                   \
      STCAT_UNROLL                                                                                       \
      for (int j = 0; j < (ROWS) / RP; ++j) {                                                            \
        const long pix = a_nb[j] < 0 ? -1 : stcat_gather_pix(g, a_nb[j], a_bh[j], a_bw[j], kh, kw);      \
        a_off[j] = pix < 0 ? STCAT_BUF_OOB : (unsigned)(pix * 4) + (t & 7) * 16;                         \
      }                                                                                                  \
    }                                                                                                    \
    const stcat_buf_t bA_ = stcat_make_buf(p.A, (KT) < kend ? p.a_bytes : 0u);                             \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < (ROWS) / RP; ++j) ra[SET][j] = stcat_buf_ld4(bA_, a_off[j], (unsigned)c0 * 4);   \
  }

// Software pipeline shared by the three kernels: LDS double buffer + two register sets, so the global loads of
// K-tile kt+2 are in flight during the MFMA phases of tiles kt and kt+1.  The prefetch is issued
// UNCONDITIONALLY (a K-tile past the end is loaded through a zero-length buffer descriptor: hardware zero
// fill, no memory traffic): a branch around the loads makes the compiler's s_waitcnt conservative
// (vmcnt counted as if the loads were skipped), which drains the new loads every iteration.
#if defined(STCAT_EXP_NOLOAD)
#define STCAT_EXP_LOAD(L) if (kt < 0) { L }
#else
#define STCAT_EXP_LOAD(L) L
#endif
#if defined(STCAT_EXP_NOSTORE)
#define STCAT_EXP_STORE(S) if (kt < 0) { S }
#else
#define STCAT_EXP_STORE(S) S
#endif
#if defined(STCAT_EXP_NOMFMA)
#define STCAT_EXP_COMPUTE(C) if (kt < 0) { C }
#else
#define STCAT_EXP_COMPUTE(C) C
#endif
// Scheduling request for the block [prefetch loads | fragment reads | MFMAs]: spread the global loads between
// MFMA groups.  A wave issues in order and a buffer load only issues when the CU's address/L1 path accepts it
// (~33 B/clk/CU measured); with all loads clustered at the top of the iteration every wave of the CU queues
// there before its first MFMA, and load time ADDS to matrix time instead of hiding under it.
#ifdef STCAT_EXP_NOINTERLEAVE
#define STCAT_BS_INTERLEAVE
#else
#define STCAT_BS_INTERLEAVE                                                                              \
  STCAT_UNROLL                                                                                           \
  for (int i_ = 0; i_ < 8; ++i_) {                                                                       \
    STCAT_SCHED_GROUP(0x008, 2);                                                                         \
    STCAT_SCHED_GROUP(0x020, 1);                                                                         \
  }
#endif
#define STCAT_BS_PIPELINE(LOAD, STORE) STCAT_BS_PIPELINE_RANGE(LOAD, STORE, 0, nk)
// K-tiles [KB, KE) of the reduction (stream-K segments start and stop inside a tile's reduction); LOAD must treat
// tiles >= `kend` as past the end
#define STCAT_BS_PIPELINE_RANGE(LOAD, STORE, KB, KE)                                                     \
  LOAD((KB), 0)                                                                                          \
  LOAD((KB) + 1, 1)                                                                                      \
  STORE(0, 0)                                                                                            \
  __syncthreads();                                                                                       \
  for (int kt = (KB); kt < (KE); kt += 2) {                                                              \
    STCAT_EXP_LOAD(LOAD(kt + 2, 0))                                                                      \
    STCAT_EXP_COMPUTE(STCAT_BS_COMPUTE(As[0], Bs[0]))                                                    \
    STCAT_BS_INTERLEAVE                                                                                  \
    if (kt + 1 < (KE)) { STCAT_EXP_STORE(STORE(1, 1)) }                                                  \
    __syncthreads();                                                                                     \
    if (kt + 1 >= (KE)) break;                                                                           \
    STCAT_EXP_LOAD(LOAD(kt + 3, 1))                                                                      \
    STCAT_EXP_COMPUTE(STCAT_BS_COMPUTE(As[1], Bs[1]))                                                    \
    STCAT_BS_INTERLEAVE                                                                                  \
    if (kt + 2 < (KE)) { STCAT_EXP_STORE(STORE(0, 0)) }                                                  \
    __syncthreads();                                                                                     \
  }

// Epilogue through LDS: the wave-private MFMA accumulators (lane = column, register = row) are written to a
// padded fp32 tile, then every thread handles whole float4 row segments, so scale/bias/residual/mask reads
// and the output stores are 16-byte, 512-byte-coalesced accesses instead of 64 dword accesses per lane.
#define STCAT_BS_ACC_TO_LDS                                                                              \
  STCAT_UNROLL                                                                                           \
  for (int tn = 0; tn < TN; ++tn) {                                                                      \
    STCAT_UNROLL                                                                                         \
    for (int tm = 0; tm < TM; ++tm) {                                                                    \
      STCAT_UNROLL                                                                                       \
      for (int r = 0; r < 16; ++r)                                                                       \
        Cs[(wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * LDC + wn * TN * 32 + tn * 32 + l31] = acc[tm][tn][r]; \
    }                                                                                                    \
  }                                                                                                      \
  __syncthreads();

#define STCAT_BS_PROLOGUE                                                                                \
  constexpr int NTHR = NWV * 64, RP = NTHR / 8;   /* threads; rows staged per pass of the R-type path */  \
  constexpr int BK = 32, LDK = STCAT_BS_LDK, TM = BM / (NWV * 16), TN = BN / 64;  /* waves: (NWV/2) x 2 */ \
  constexpr int A_ELEMS = NS * BM * LDK, B_ELEMS = NS * BN * LDK, LDC = BN + 4;                          \
  constexpr int SMEM_BYTES = (2 * (A_ELEMS + B_ELEMS) * 2 > BM * LDC * 4) ? 2 * (A_ELEMS + B_ELEMS) * 2 : BM * LDC * 4; \
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];                                     \
  __bf16 (*As)[A_ELEMS] = reinterpret_cast<__bf16 (*)[A_ELEMS]>(smem_raw);                               \
  __bf16 (*Bs)[B_ELEMS] = reinterpret_cast<__bf16 (*)[B_ELEMS]>(smem_raw + 2 * A_ELEMS * 2);             \
  float* Cs = reinterpret_cast<float*>(smem_raw); /* epilogue: the fp32 output tile, rows padded to LDC */ \
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;                                               \
  const int l31 = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;                              \
  const int trow = ((t >> 3) & ~7) | (((t >> 3) & 1) << 2) | ((t >> 4) & 3);                             \
  const int num_n = p.N / BN;                                                                            \
  const IgemmGeom g = p.g;                                                                               \
  const stcat_buf_t bufA = stcat_make_buf(p.A, p.a_bytes);                                               \
  const stcat_buf_t bufB = stcat_make_buf(p.B, p.b_bytes);                                               \
  STCAT_BS_TILE_OF_BLOCK
// which output tile a workgroup owns: one per block (XCD-aware remap) — the stream-K kernel overrides this
#define STCAT_BS_TILE_OF_BLOCK_DEFAULT                                                                   \
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);                                                  \
  const int m0 = (v / num_n) * BM, n0 = (v % num_n) * BN;
#define STCAT_BS_TILE_OF_BLOCK STCAT_BS_TILE_OF_BLOCK_DEFAULT

// forward epilogue of one float4 of output row m, columns n..n+3 (also the data gradient through this kernel on
// pre-transposed weights: mask / mscale = fused ReLU+BN backward of the layer below, C2 = second scaled output)
#define STCAT_BS_FWD_EPILOGUE_F4(m, n, v4)                                                               \
  {                                                                                                      \
    if (p.scale) {                                                                                       \
      const float4 sc = stcat_ld4(p.scale + (n));                                                        \
      v4.x *= sc.x; v4.y *= sc.y; v4.z *= sc.z; v4.w *= sc.w;                                            \
    }                                                                                                    \
    if (p.bias) {                                                                                        \
      const float4 bi = stcat_ld4(p.bias + (n));                                                         \
      v4.x += bi.x; v4.y += bi.y; v4.z += bi.z; v4.w += bi.w;                                            \
    }                                                                                                    \
    if (p.res) {                                                                                         \
      const float4 rr = stcat_ld4(p.res + (long)(m) * p.ldr + (n));                                      \
      v4.x += rr.x; v4.y += rr.y; v4.z += rr.z; v4.w += rr.w;                                            \
    }                                                                                                    \
    if (p.relu) { v4.x = fmaxf(v4.x, 0.f); v4.y = fmaxf(v4.y, 0.f); v4.z = fmaxf(v4.z, 0.f); v4.w = fmaxf(v4.w, 0.f); } \
    if (p.drop.thresh) {                                                                                 \
      const DropParams dp_ = stcat_drop_resolve(p.drop);                                                 \
      const unsigned long long i0_ = (unsigned long long)(m) * (unsigned long long)p.N + (unsigned long long)(n); \
      v4.x *= stcat_drop_mul(dp_, i0_); v4.y *= stcat_drop_mul(dp_, i0_ + 1);                            \
      v4.z *= stcat_drop_mul(dp_, i0_ + 2); v4.w *= stcat_drop_mul(dp_, i0_ + 3);                        \
    }                                                                                                    \
    if (p.mask) {                                                                                        \
      const float4 mk = stcat_ld4(p.mask + (long)(m) * p.ldc + (n));                                     \
      const float mg_ = p.mask_gain != 0.f ? p.mask_gain : 1.f;                                          \
      float4 ms = make_float4(mg_, mg_, mg_, mg_);                                                       \
      if (p.mscale) { ms = stcat_ld4(p.mscale + (n)); ms.x *= mg_; ms.y *= mg_; ms.z *= mg_; ms.w *= mg_; } \
      v4.x = mk.x > 0.f ? v4.x * ms.x : 0.f; v4.y = mk.y > 0.f ? v4.y * ms.y : 0.f;                      \
      v4.z = mk.z > 0.f ? v4.z * ms.z : 0.f; v4.w = mk.w > 0.f ? v4.w * ms.w : 0.f;                      \
    }                                                                                                    \
    stcat_st4(p.C + (long)(m) * p.ldc + (n), v4);                                                        \
    if (p.C2) {                                                                                          \
      const float4 s2 = stcat_ld4(p.c2scale + (n));                                                      \
      stcat_st4(p.C2 + (long)(m) * p.ldc + (n), make_float4(v4.x * s2.x, v4.y * s2.y, v4.z * s2.z, v4.w * s2.w)); \
    }                                                                                                    \
  }

// split-K epilogue: slice 0 also contributes bias and residual; the host zeroes the output and guarantees that no
// scale / ReLU / mask / second output is requested
#define STCAT_BS_SPLITK_EPILOGUE_F4(m, n, v4)                                                            \
  {                                                                                                      \
    if (blockIdx.z == 0) {                                                                               \
      if (p.bias) {                                                                                      \
        const float4 bi = stcat_ld4(p.bias + (n));                                                       \
        v4.x += bi.x; v4.y += bi.y; v4.z += bi.z; v4.w += bi.w;                                          \
      }                                                                                                  \
      if (p.res) {                                                                                       \
        const float4 rr = stcat_ld4(p.res + (long)(m) * p.ldr + (n));                                    \
        v4.x += rr.x; v4.y += rr.y; v4.z += rr.z; v4.w += rr.w;                                          \
      }                                                                                                  \
    }                                                                                                    \
    float* d_ = p.C + (long)(m) * p.ldc + (n);                                                           \
    atomicAdd(d_, v4.x); atomicAdd(d_ + 1, v4.y); atomicAdd(d_ + 2, v4.z); atomicAdd(d_ + 3, v4.w);      \
  }

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// NWV = 4: 2x2 waves, two workgroups per CU.  NWV = 8 (256x128 tile): 4x2 waves, ONE workgroup per CU — a K step
// then carries twice the MFMA work per barrier while only 1.5x the operand bytes (tools/bench_quant.py: a second
// co-resident 4-wave workgroup adds no throughput, so the CU is better spent on one bigger tile).
template <int BM, int BN, int NS, int NWV = 4>
static __device__ __forceinline__ void igemm_bs_fwd_body(IgemmParams p) {
  STCAT_BS_PROLOGUE
  STCAT_BS_GATHER_DECL(BM)
  STCAT_BS_ACC_INIT
  unsigned b_off[BN / RP];
  STCAT_UNROLL
  for (int j = 0; j < BN / RP; ++j) b_off[j] = (unsigned)((n0 + trow + RP * j) * p.ldb + (t & 7) * 4) * 4;
  float4 ra[2][BM / RP], rb[2][BN / RP];
  // optional split of the reduction over grid.z (skinny launches: M <= 64 rows, K >= 1024 — the decoders' FFN):
  // K-tiles [kbeg, kend) per slice, partial tiles added atomically into the zeroed output
  const int nk = p.K / BK;
  const int kbeg = p.k_chunk ? (int)blockIdx.z * p.k_chunk : 0;
  const int kend = p.k_chunk ? min(nk, kbeg + p.k_chunk) : nk;
#define STCAT_BSF_LOAD(KT, SET)                                                                          \
  STCAT_BS_LOAD_A_GATHER(KT, BM, SET)                                                                    \
  {                                                                                                      \
    const int r0b = (KT) * BK, tapb = r0b / g.C, c0b = r0b - tapb * g.C;                                 \
    const unsigned soffb = (unsigned)tapb * p.b_tap_stride + (unsigned)c0b * 4;                          \
    const stcat_buf_t bB_ = stcat_make_buf(p.B, (KT) < kend ? p.b_bytes : 0u);                           \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < BN / RP; ++j) rb[SET][j] = stcat_buf_ld4(bB_, b_off[j], soffb);                  \
  }
#define STCAT_BSF_STORE(SET, BUF) \
  STCAT_BS_STORE_R(As[BUF], BM, ra[SET]) STCAT_BS_STORE_R(Bs[BUF], BN, rb[SET])
  STCAT_BS_PIPELINE_RANGE(STCAT_BSF_LOAD, STCAT_BSF_STORE, kbeg, kend)
#undef STCAT_BSF_LOAD
#undef STCAT_BSF_STORE
  STCAT_BS_ACC_TO_LDS
  constexpr int F4 = BN / 4;
  STCAT_UNROLL
  for (int j = 0; j < BM * F4 / NTHR; ++j) {
    const int i = t + NTHR * j, row = i / F4, c4 = i - row * F4;
    const int m = m0 + row, n = n0 + c4 * 4;
    if (m < p.M) {
      float4 v4 = stcat_ld4(&Cs[row * LDC + c4 * 4]);
      if (p.k_chunk) {
        STCAT_BS_SPLITK_EPILOGUE_F4(m, n, v4)
      } else {
        STCAT_BS_FWD_EPILOGUE_F4(m, n, v4)
      }
    }
  }
}

template <int BM, int BN, int NS, int NWV = 4>
__global__ void __launch_bounds__(NWV * 64, 2) igemm_bs_fwd_kernel(IgemmParams p) {
  igemm_bs_fwd_body<BM, BN, NS, NWV>(p);
}

// Several INDEPENDENT skinny problems of one shape in one launch (round 5): blockIdx.y picks the problem — the seven
// input projections of a box-decoder layer's self-attention (query_decoder.py:329-338), its three in-projections, ...:
// each was a ~8 us launch on a dependent chain although none depends on another.  Problems that write the SAME output
// (q = Wqc tgt + Wqt time + Wqp pos) are separate problems of the accumulating split-K form (atomic epilogue onto a
// zeroed output; slice 0 of each adds its bias).
struct IgemmMulti {
  IgemmParams base;
  const float* A[8];
  const float* B[8];
  const float* bias[8];      // fwd: bias; dgrad: the `add` operand; wgrad: unused
  float* C[8];
  float* rowsum[8];          // wgrad: bias-gradient accumulators (may be null)
};
template <int NS>
__global__ void __launch_bounds__(256, 2) igemm_bs_fwd_multi_kernel(IgemmMulti mp) {
  IgemmParams p = mp.base;
  const int j = blockIdx.y;
  p.A = mp.A[j]; p.B = mp.B[j]; p.bias = mp.bias[j]; p.C = mp.C[j];
  igemm_bs_fwd_body<64, 64, NS, 4>(p);
}

// ---------------------------------------------------------------------------------------------------
// stream-K forward (bf16x3, 8-wave 256x128 tiles): ONE workgroup per CU, each owning an equal share of the
// tiles x K-steps space.  A layer3 launch has 392 such tiles for 256 CUs: tile-per-workgroup scheduling runs two
// rounds (the second 53 % full); here every CU gets 110 K-steps: at most one tile tail, whole tiles, one tile head.
// Whole tiles take the normal epilogue.  The two partial pieces of a worker go to workspace slots (2w: tail piece
// [kb, nk) of its first tile, 2w+1: head piece [0, ke) of its last tile) as raw fp32 tiles, and a second small
// kernel adds the two halves of every split tile (tail of worker w + head of worker w-1) and applies the epilogue.
// The kernel boundary is the only synchronisation: no flags, no spinning, no cross-XCD visibility games.
// Requires tiles >= workers (so a tile is split between at most two workers).
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int NS, int NWV>
__global__ void __launch_bounds__(NWV * 64, 2) igemm_bs_fwd_sk_kernel(IgemmParams p) {
#undef STCAT_BS_TILE_OF_BLOCK
#define STCAT_BS_TILE_OF_BLOCK
  STCAT_BS_PROLOGUE
#undef STCAT_BS_TILE_OF_BLOCK
#define STCAT_BS_TILE_OF_BLOCK STCAT_BS_TILE_OF_BLOCK_DEFAULT
  const int nk = p.K / BK;
  const int n_tiles = ((p.M + BM - 1) / BM) * num_n;
  const long S = (long)n_tiles * nk;
  const int w = blockIdx.x, G = gridDim.x;
  long s = (long)w * S / G;
  const long s1 = (long)(w + 1) * S / G;
  constexpr int F4 = BN / 4;
  while (s < s1) {
    const int tile = (int)(s / nk), kb = (int)(s - (long)tile * nk);
    const int ke = (s1 - s) < (long)(nk - kb) ? kb + (int)(s1 - s) : nk;
    const int kend = ke;
    const int m0 = (tile / num_n) * BM, n0 = (tile % num_n) * BN;
    STCAT_BS_GATHER_DECL(BM)
    STCAT_BS_ACC_INIT
    unsigned b_off[BN / RP];
    STCAT_UNROLL
    for (int j = 0; j < BN / RP; ++j) b_off[j] = (unsigned)((n0 + trow + RP * j) * p.ldb + (t & 7) * 4) * 4;
    float4 ra[2][BM / RP], rb[2][BN / RP];
#define STCAT_BSF_LOAD(KT, SET)                                                                          \
  STCAT_BS_LOAD_A_GATHER(KT, BM, SET)                                                                    \
  {                                                                                                      \
    const int r0b = (KT) * BK, tapb = r0b / g.C, c0b = r0b - tapb * g.C;                                 \
    const unsigned soffb = (unsigned)tapb * p.b_tap_stride + (unsigned)c0b * 4;                          \
    const stcat_buf_t bB_ = stcat_make_buf(p.B, (KT) < kend ? p.b_bytes : 0u);                           \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < BN / RP; ++j) rb[SET][j] = stcat_buf_ld4(bB_, b_off[j], soffb);                  \
  }
#define STCAT_BSF_STORE(SET, BUF) \
  STCAT_BS_STORE_R(As[BUF], BM, ra[SET]) STCAT_BS_STORE_R(Bs[BUF], BN, rb[SET])
    STCAT_BS_PIPELINE_RANGE(STCAT_BSF_LOAD, STCAT_BSF_STORE, kb, ke)
#undef STCAT_BSF_LOAD
#undef STCAT_BSF_STORE
    STCAT_BS_ACC_TO_LDS
    if (kb == 0 && ke == nk) {
      STCAT_UNROLL
      for (int j = 0; j < BM * F4 / NTHR; ++j) {
        const int i = t + NTHR * j, row = i / F4, c4 = i - row * F4;
        const int m = m0 + row, n = n0 + c4 * 4;
        if (m < p.M) {
          float4 v4 = stcat_ld4(&Cs[row * LDC + c4 * 4]);
          STCAT_BS_FWD_EPILOGUE_F4(m, n, v4)
        }
      }
    } else {  // partial sum of a split tile -> this worker's workspace slot (raw fp32 tile, row-major BM x BN)
      float* slot = p.sk_ws + ((long)w * 2 + (kb == 0 ? 1 : 0)) * (BM * BN);
      STCAT_UNROLL
      for (int j = 0; j < BM * F4 / NTHR; ++j) {
        const int i = t + NTHR * j, row = i / F4, c4 = i - row * F4;
        stcat_st4(slot + row * BN + c4 * 4, stcat_ld4(&Cs[row * LDC + c4 * 4]));
      }
    }
    __syncthreads();  // the fp32 tile aliases the operand stages of the next segment
    s += ke - kb;
  }
}

// second kernel of the stream-K forward: block w finishes the tile that worker w entered in the middle
template <int BM, int BN>
__global__ void __launch_bounds__(256) igemm_bs_fwd_sk_fixup_kernel(IgemmParams p, int n_workers) {
  const int t = threadIdx.x, w = blockIdx.x;
  const int num_n = p.N / BN, nk = p.K / 32;
  const int n_tiles = ((p.M + BM - 1) / BM) * num_n;
  const long S = (long)n_tiles * nk;
  const long s0 = (long)w * S / n_workers;
  const int tile = (int)(s0 / nk), kb = (int)(s0 - (long)tile * nk);
  if (kb == 0) return;  // this worker started on a tile boundary: nothing was split here
  const int m0 = (tile / num_n) * BM, n0 = (tile % num_n) * BN;
  const float* tail = p.sk_ws + ((long)w * 2) * (BM * BN);           // [kb, nk) by worker w
  const float* head = p.sk_ws + ((long)(w - 1) * 2 + 1) * (BM * BN); // [0, kb) by worker w-1
  constexpr int F4 = BN / 4;
  for (int i = t; i < BM * F4; i += 256) {
    const int row = i / F4, c4 = i - row * F4;
    const int m = m0 + row, n = n0 + c4 * 4;
    if (m < p.M) {
      const float4 a = stcat_ld4(tail + row * BN + c4 * 4), b = stcat_ld4(head + row * BN + c4 * 4);
      float4 v4 = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
      STCAT_BS_FWD_EPILOGUE_F4(m, n, v4)
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// dgrad: A = gathered dY (R), B[k=(tap,co)][n=ci] = W[co][tap][ci] (O)
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int NS>
static __device__ __forceinline__ void igemm_bs_dgrad_body(IgemmParams p) {
  constexpr int NWV = 4;
  STCAT_BS_PROLOGUE
  constexpr int JB = (BN * 2 + 255) / 256;
  STCAT_BS_GATHER_DECL(BM)
  STCAT_BS_ACC_INIT
  unsigned b_off[JB][4];
  STCAT_UNROLL
  for (int j = 0; j < JB; ++j) {
    const int i = t + 256 * j, kgrp = i & 7, rowgrp = i >> 3;
    STCAT_UNROLL
    for (int e = 0; e < 4; ++e)
      b_off[j][e] = i < BN * 2 ? (unsigned)((kgrp * 4 + e) * p.ldb + n0 + rowgrp * 4) * 4 : STCAT_BUF_OOB;
  }
  float4 ra[2][BM / 32], rb[2][JB][4];
  const int nk = p.K / BK;
  const int kbeg = p.k_chunk ? (int)blockIdx.z * p.k_chunk : 0;   // optional split-K, as in the forward kernel
  const int kend = p.k_chunk ? min(nk, kbeg + p.k_chunk) : nk;
#define STCAT_BSD_LOAD(KT, SET)                                                                          \
  STCAT_BS_LOAD_A_GATHER(KT, BM, SET)                                                                    \
  {                                                                                                      \
    const int r0 = (KT) * BK, tap = r0 / g.C, co0 = r0 - tap * g.C;                                      \
    const unsigned soff = (unsigned)(co0 * p.ldb + tap * p.N) * 4;                                       \
    const stcat_buf_t bB_ = stcat_make_buf(p.B, (KT) < kend ? p.b_bytes : 0u);                           \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < JB; ++j) {                                                                       \
      STCAT_UNROLL                                                                                       \
      for (int e = 0; e < 4; ++e) rb[SET][j][e] = stcat_buf_ld4(bB_, b_off[j][e], soff);                 \
    }                                                                                                    \
  }
#define STCAT_BSD_STORE(SET, BUF) \
  STCAT_BS_STORE_R(As[BUF], BM, ra[SET]) STCAT_BS_STORE_O(Bs[BUF], BN, rb[SET])
  STCAT_BS_PIPELINE_RANGE(STCAT_BSD_LOAD, STCAT_BSD_STORE, kbeg, kend)
#undef STCAT_BSD_LOAD
#undef STCAT_BSD_STORE
  STCAT_BS_ACC_TO_LDS
  constexpr int F4 = BN / 4;
  STCAT_UNROLL
  for (int j = 0; j < BM * F4 / NTHR; ++j) {
    const int i = t + NTHR * j, row = i / F4, c4 = i - row * F4;
    const int m = m0 + row, n = n0 + c4 * 4;
    if (m < p.M) {
      float4 v4 = stcat_ld4(&Cs[row * LDC + c4 * 4]);
      if (p.k_chunk) {
        STCAT_BS_SPLITK_EPILOGUE_F4(m, n, v4)
        continue;
      }
      if (p.res) {
        const float4 rr = stcat_ld4(p.res + (long)m * p.ldr + n);
        v4.x += rr.x; v4.y += rr.y; v4.z += rr.z; v4.w += rr.w;
      }
      if (p.mask) {  // fused ReLU+BN backward of the layer below
        const float4 mk = stcat_ld4(p.mask + (long)m * p.ldc + n);
        const float mg_ = p.mask_gain != 0.f ? p.mask_gain : 1.f;
        float4 ms = make_float4(mg_, mg_, mg_, mg_);
        if (p.mscale) { ms = stcat_ld4(p.mscale + n); ms.x *= mg_; ms.y *= mg_; ms.z *= mg_; ms.w *= mg_; }
        v4.x = mk.x > 0.f ? v4.x * ms.x : 0.f; v4.y = mk.y > 0.f ? v4.y * ms.y : 0.f;
        v4.z = mk.z > 0.f ? v4.z * ms.z : 0.f; v4.w = mk.w > 0.f ? v4.w * ms.w : 0.f;
      }
      stcat_st4(p.C + (long)m * p.ldc + n, v4);
      if (p.C2) {
        const float4 s2 = stcat_ld4(p.c2scale + n);
        stcat_st4(p.C2 + (long)m * p.ldc + n, make_float4(v4.x * s2.x, v4.y * s2.y, v4.z * s2.z, v4.w * s2.w));
      }
    }
  }
}

template <int BM, int BN, int NS>
__global__ void __launch_bounds__(256, 2) igemm_bs_dgrad_kernel(IgemmParams p) {
  igemm_bs_dgrad_body<BM, BN, NS>(p);
}
template <int NS>
__global__ void __launch_bounds__(256, 2) igemm_bs_dgrad_multi_kernel(IgemmMulti mp) {
  IgemmParams p = mp.base;
  const int j = blockIdx.y;
  p.A = mp.A[j]; p.B = mp.B[j]; p.res = mp.bias[j]; p.C = mp.C[j];
  igemm_bs_dgrad_body<64, 64, NS>(p);
}

// ---------------------------------------------------------------------------------------------------
// wgrad: A[k=pixel][row=co] = dY (O), B[k=pixel][col=(tap,ci)] = gathered X (O); split-K over grid.z, atomics
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN, int NS, int NWV = 4>
static __device__ __forceinline__ void igemm_bs_wgrad_body(IgemmParams p) {
  STCAT_BS_PROLOGUE
  constexpr int JA = (BM * 2 + NTHR - 1) / NTHR, JB = (BN * 2 + NTHR - 1) / NTHR;
  const int tap = n0 / g.C, ci0 = n0 - tap * g.C;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int red0 = blockIdx.z * p.k_chunk;
  const int red1 = min(p.K, red0 + p.k_chunk);
  const int ohw = g.OH * g.OW;
  const int nk = (red1 - red0 + BK - 1) / BK;
  if (nk <= 0) return;
  STCAT_BS_ACC_INIT
  unsigned a_off[JA][4];
  STCAT_UNROLL
  for (int j = 0; j < JA; ++j) {
    const int i = t + NTHR * j, kgrp = i & 7, rowgrp = i >> 3;
    STCAT_UNROLL
    for (int e = 0; e < 4; ++e)
      a_off[j][e] = i < BM * 2 ? (unsigned)((kgrp * 4 + e) * p.ldb + m0 + rowgrp * 4) * 4 : STCAT_BUF_OOB;
  }
  float4 ra[2][JA][4], rb[2][JB][4];
  const bool do_rs = p.rowsum != nullptr && n0 == 0;  // first column tile of each row block owns the row sums
  float4 rs[JA];
  STCAT_UNROLL
  for (int j = 0; j < JA; ++j) rs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Gathered B: the pixel coordinates of a thread's 4 consecutive reduction rows come from two multiply-shift
  // divisions per K-tile and carry with selects — no branch anywhere in the load, so it shares a basic block with
  // the MFMAs and the scheduler can spread the loads between them (STCAT_BS_INTERLEAVE).
#define STCAT_BSW_LOAD(KT, SET)                                                                          \
  {                                                                                                      \
    const int mbase = red0 + (KT) * BK;                                                                  \
    const unsigned soff = (unsigned)mbase * (unsigned)p.ldb * 4u;                                        \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < JA; ++j) {                                                                       \
      const int kgrp = (t + NTHR * j) & 7;                                                                \
      STCAT_UNROLL                                                                                       \
      for (int e = 0; e < 4; ++e)                                                                        \
        ra[SET][j][e] = stcat_buf_ld4(bufA, mbase + kgrp * 4 + e < red1 ? a_off[j][e] : STCAT_BUF_OOB, soff); \
    }                                                                                                    \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < JB; ++j) {                                                                       \
      const int i = t + NTHR * j, kgrp = i & 7, rowgrp = i >> 3;                                          \
      const int mfirst = mbase + kgrp * 4;                                                               \
      int nb = stcat_fastdiv(mfirst, g.mg_ohw, g.sh_ohw);                                                \
      const int rem = mfirst - nb * ohw;                                                                 \
      int oh = stcat_fastdiv(rem, g.mg_ow, g.sh_ow);                                                     \
      int ow = rem - oh * g.OW;                                                                          \
      STCAT_UNROLL                                                                                       \
      for (int e = 0; e < 4; ++e) {                                                                      \
        const int h = oh * g.mul + g.off + kh, w = ow * g.mul + g.off + kw;                              \
        const bool ok = (mfirst + e < red1) & (i < BN * 2) & ((unsigned)h < (unsigned)g.H) &             \
                        ((unsigned)w < (unsigned)g.W);                                                   \
        const unsigned voff = (unsigned)(((nb * g.H + h) * g.W + w) * g.ld + ci0 + rowgrp * 4) * 4u;     \
        rb[SET][j][e] = stcat_buf_ld4(bufB, ok ? voff : STCAT_BUF_OOB, 0u);                              \
        ++ow;                                                                                            \
        const bool c1 = ow == g.OW;                                                                      \
        ow = c1 ? 0 : ow;                                                                                \
        oh += c1 ? 1 : 0;                                                                                \
        const bool c2 = oh == g.OH;                                                                      \
        oh = c2 ? 0 : oh;                                                                                \
        nb += c2 ? 1 : 0;                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }
#define STCAT_BSW_STORE(SET, BUF)                                                                        \
  if (do_rs) { /* bias gradient rides along: column sums of dY from the registers already loaded */     \
    STCAT_UNROLL                                                                                         \
    for (int j = 0; j < JA; ++j) {                                                                       \
      STCAT_UNROLL                                                                                       \
      for (int e = 0; e < 4; ++e) {                                                                      \
        rs[j].x += ra[SET][j][e].x; rs[j].y += ra[SET][j][e].y;                                          \
        rs[j].z += ra[SET][j][e].z; rs[j].w += ra[SET][j][e].w;                                          \
      }                                                                                                  \
    }                                                                                                    \
  }                                                                                                      \
  STCAT_BS_STORE_O(As[BUF], BM, ra[SET]) STCAT_BS_STORE_O(Bs[BUF], BN, rb[SET])
  STCAT_BS_PIPELINE(STCAT_BSW_LOAD, STCAT_BSW_STORE)
#undef STCAT_BSW_LOAD
#undef STCAT_BSW_STORE
  if (do_rs) {
    STCAT_UNROLL
    for (int j = 0; j < JA; ++j) {
      const int i = t + NTHR * j;
      if (i < BM * 2) {  // wave-uniform (BM*2 is a multiple of 64)
        float4 v = rs[j];
        STCAT_UNROLL
        for (int m = 1; m <= 4; m <<= 1) {  // the 8 k-groups of one row group are 8 consecutive lanes
          v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m);
          v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
        }
        if ((i & 7) == 0) {
          float* dst = p.rowsum + m0 + (i >> 3) * 4;
          atomicAdd(dst, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        }
      }
    }
  }
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

template <int BM, int BN, int NS, int NWV = 4>
__global__ void __launch_bounds__(NWV * 64, 2) igemm_bs_wgrad_kernel(IgemmParams p) {
  igemm_bs_wgrad_body<BM, BN, NS, NWV>(p);
}
template <int BM, int NS>
__global__ void __launch_bounds__(256, 2) igemm_bs_wgrad_multi_kernel(IgemmMulti mp) {
  IgemmParams p = mp.base;
  const int j = blockIdx.y;
  p.A = mp.A[j]; p.B = mp.B[j]; p.C = mp.C[j]; p.rowsum = mp.rowsum[j];
  igemm_bs_wgrad_body<BM, BM, NS, 4>(p);
}
