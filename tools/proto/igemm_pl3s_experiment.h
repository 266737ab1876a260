// EXPERIMENT, not built (profiles/r03_tile7_k16_experiment.log: correct but 3-9 % slower than the one-workgroup tiles).
// To try it again: move the epilogue of igemm_pl_fwd_kernel into igemm_pl_epi.inc, include this file from stcat_capi.hip,
// add a tile-table entry that launches igemm_pl3s_fwd_kernel<128, 128, 2, 2, 4> with 72 KB of dynamic LDS.
// Three-plane forward / data-gradient GEMM with SHORT K-tiles (16 reduction terms) and THREE stages of LDS — the
// form for the K <= 512 1x1 convolutions of the bottleneck blocks (torchvision Bottleneck conv1 / conv3 through
// models/vision_model/backbone.py:115-119) in mma mode bf16x6p.
//
// Why: those launches are bound by the epilogue's HBM traffic (three residual planes in, three planes + bit mask out:
// 693 MB per layer3 256 -> 1024 launch), which is as long as their K loop, and with ONE workgroup per CU (the 144 KB
// two-stage tiles of igemm_pl_fwd_kernel) nothing overlaps the two phases.  A stage of 16-wide K-tiles is
// 3 x (BM + BN) x 32 bytes: the 128 x 128 tile needs 3 x 24 KB, so TWO four-wave workgroups share a CU with the same
// 64 x 64 wave tile (24 MFMAs against 12 ds_read_b128 per k-step) as the one-workgroup kernel; their phases drift
// apart and one's epilogue runs under the other's MFMAs.  (The two-plane kernel gets the same effect from its 64 KB
// 128 x 128 tile: DESIGN.md §4; the 128 x 64 four-wave tile of round 3 paid for it with 64 x 32 wave tiles.)
//
// Pipeline (one k-step per K-tile): stage of tile t = t % 3.  Half-iteration H(t), entered behind a barrier with the
// fragments of tile t in registers, tile t+1 landed and tile t+2 in flight:
//     DMA tile t+3 -> stage t % 3 (its last readers passed the barrier) | read the fragments of tile t+1 |
//     24 MFMAs on tile t | s_waitcnt vmcnt(<DMA instructions of one tile>) = tile t+2 has landed, t+3 stays in flight |
//     barrier.
// A DMA has two half-iterations (48 MFMAs per wave, as in the 32-wide two-stage kernel) to land.  Every wave issues
// the same number of DMA instructions per tile (BM, BN multiples of 32 x NW), so the counted wait is exact.
// LDS rows are 32 B (16 bf16): piece = 1 KiB = 32 rows x 2 chunks, lane -> row (lane >> 1), physical chunk (lane & 1)
// holding source chunk (lane & 1) ^ ((row >> 3) & 1): rows r and r + 8 of a 16-lane read group land in the two
// halves of one 32-byte slot — conflict-free ds_read_b128, like the 64-byte rows of igemm_pl.h.
#pragma once
#include "igemm_pl.h"

template <int BM, int BN, int WM, int WN, int NW>
__global__ void __launch_bounds__(NW * 64, 2) igemm_pl3s_fwd_kernel(PlParams p) {
  static_assert(WM * WN == NW, "NW waves");
  constexpr int NP = 3;
  constexpr bool F32 = false;
  constexpr int BK = 16, NST = 3, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;            // bytes: rows x 32 B
  constexpr int STAGE = NP * (PLANE_A + PLANE_B);
  static_assert(NST * STAGE <= 160 * 1024, "three stages fit the CU's LDS");
  constexpr int QA = BM / 32, QB = BN / 32;                      // 1-KiB DMA pieces (32 rows) per plane
  static_assert(QA % NW == 0 && QB % NW == 0, "every wave issues the same number of DMA instructions per tile");
  constexpr int RQA = QA / NW, RQB = QB / NW;
  constexpr int LDE = TN * 32 + 4;
  constexpr int EPI_WAVE = 32 * LDE * 4;
  static_assert(NW * EPI_WAVE <= NST * STAGE, "epilogue blocks fit the operand stages");
  STCAT_DYN_SHARED(char, smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wq = STCAT_READFIRSTLANE(wave);
  const int wm = wave / WN, wn = wave % WN;
  const int num_n = p.N / BN;
  const int v = stcat_xcd_remap(blockIdx.x, gridDim.x);
  const IgemmGeom g = p.g;
  // row space and the taps of a parity class: as in igemm_pl_fwd_kernel
  const int Mc = p.par ? p.M >> 2 : p.M;
  const int tpc = (Mc + BM - 1) / BM;
  const int mt = v / num_n;
  const int cls = p.par ? mt / tpc : 0, py = cls >> 1, px = cls & 1;
  const int m0 = (mt - cls * tpc) * BM, n0 = (v % num_n) * BN;
  const int OHc = p.par ? g.OH >> 1 : g.OH, OWc = p.par ? g.OW >> 1 : g.OW;
  const int kstep = p.par ? 2 : 1;
  const int kh0 = p.par ? ((py + g.off) & 1) : 0, kw0 = p.par ? ((px + g.off) & 1) : 0;
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + kstep - 1) / kstep : 0, nkw = kw0 < g.KW ? (g.KW - kw0 + kstep - 1) / kstep : 0;
  const int nk = p.par ? nkh * nkw * (g.C / BK) : p.K / BK;

  // ---- DMA bookkeeping: piece q = wave + NW i covers rows 32 q .. 32 q + 31 of a plane
  int a_nb[RQA], a_bh[RQA], a_bw[RQA];
  unsigned a_c16[RQA], b_voff[RQB];
  STCAT_UNROLL
  for (int i = 0; i < RQA; ++i) {
    const int q = wave + NW * i, r = q * 32 + (lane >> 1), m = m0 + r;
    a_c16[i] = (unsigned)(((lane & 1) ^ ((r >> 3) & 1)) * 16);
    a_nb[i] = -1; a_bh[i] = 0; a_bw[i] = 0;
    if (m < Mc) {
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
    const int q = wave + NW * i, r = q * 32 + (lane >> 1);
    b_voff[i] = (unsigned)(((n0 + r) * p.ldb) * 2 + ((lane & 1) ^ ((r >> 3) & 1)) * 16);
  }
  const __bf16* Ap[3] = {p.Ah, p.Al, stcat_plane(p.Ah, p.Al, 2)};
  const __bf16* Bp[3] = {p.Bh, p.Bl, stcat_plane(p.Bh, p.Bl, 2)};
  int kl = 0, l_c0 = 0, l_kh = kh0, l_kw = kw0, l_tap = kh0 * g.KW + kw0;
  const int dmask = g.div - 1, dshift = g.div > 1 ? 31 - __builtin_clz(g.div) : 0;
  // tiles past the end go through zero-length descriptors (zero fill into a stage nobody reads): branch-free body
#define STCAT_PL3S_STAGE_LOAD(ST)                                                                       \
  {                                                                                                     \
    const bool live_ = kl < nk;                                                                         \
    stcat_buf_t dA_[NP], dB_[NP];                                                                       \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_) {                                                                \
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
      STCAT_UNROLL                                                                                      \
      for (int pi_ = 0; pi_ < NP; ++pi_) stcat_glds16(dA_[pi_], base_ + pi_ * PLANE_A + i * (NW * 1024), vo_, soA_); \
    }                                                                                                   \
    STCAT_UNROLL                                                                                        \
    for (int i = 0; i < RQB; ++i) {                                                                     \
      STCAT_UNROLL                                                                                      \
      for (int pi_ = 0; pi_ < NP; ++pi_)                                                                \
        stcat_glds16(dB_[pi_], base_ + NP * PLANE_A + pi_ * PLANE_B + i * (NW * 1024), b_voff[i], soB_); \
    }                                                                                                   \
    /* K order: channel chunk outer, filter tap inner (igemm_pl.h) */                                   \
    ++kl; l_kw += kstep;                                                                                \
    if (l_kw >= g.KW) {                                                                                 \
      l_kw = kw0; l_kh += kstep;                                                                        \
      if (l_kh >= g.KH) { l_kh = kh0; l_c0 += BK; }                                                     \
    }                                                                                                   \
    l_tap = l_kh * g.KW + l_kw;                                                                         \
  }

  // ---- fragment addressing: lane -> row l31 of its 32-row tile, the k-step's chunk hi ^ ((row >> 3) & 1)
  const unsigned fchunk = (unsigned)((hi ^ ((l31 >> 3) & 1)) * 16);
  const unsigned fa_off = (unsigned)((wm * TM * 32 + l31) * 32) + fchunk;
  const unsigned fb_off = (unsigned)(NP * PLANE_A + (wn * TN * 32 + l31) * 32) + fchunk;
  struct Frag { bf16x8 a[NP][TM], b[NP][TN]; };
#define STCAT_PL3S_READ_FRAG(F, SB)                                                                     \
  STCAT_UNROLL                                                                                          \
  for (int tn = 0; tn < TN; ++tn) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_)                                                                  \
      F.b[pi_][tn] = *reinterpret_cast<const bf16x8*>((SB) + fb_off + pi_ * PLANE_B + tn * 1024);       \
  }                                                                                                     \
  STCAT_UNROLL                                                                                          \
  for (int tm = 0; tm < TM; ++tm) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_)                                                                  \
      F.a[pi_][tm] = *reinterpret_cast<const bf16x8*>((SB) + fa_off + pi_ * PLANE_A + tm * 1024);       \
  }

  STCAT_PL_ACC_INIT
  constexpr int NMMA = PlProd<NP>::N * TM * TN, NRD = NP * (TM + TN), NDMA = NP * (RQA + RQB);
  Frag fa, fb;
  STCAT_PL3S_STAGE_LOAD(0)
  STCAT_PL3S_STAGE_LOAD(1)
  STCAT_PL3S_STAGE_LOAD(2)
  STCAT_WAIT_VM(NDMA);   // tiles 0 and 1 have landed, tile 2 may still be in flight
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();
  STCAT_PL3S_READ_FRAG(fa, smem)
  STCAT_WAIT_VM0_LGKM0();   // once: every wave holds tile 0 in registers before H(0) overwrites stage 0 (and tile 2 landed)
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();
  int s0 = 0, s1 = 1;       // stage of tile kt, of tile kt + 1
  for (int kt = 0; kt < nk; kt += 2) {
    // H(kt): tile kt in fa
    STCAT_PL3S_STAGE_LOAD(s0)
    STCAT_PL3S_READ_FRAG(fb, smem + s1 * STAGE)
    STCAT_PL_INTERLEAVE(NMMA, NRD, NDMA)
    STCAT_PL_MMA(fa)
    STCAT_SCHED_FENCE();
    STCAT_WAIT_VMN_LGKM0(NDMA);
    STCAT_S_BARRIER();
    STCAT_SCHED_FENCE();
    s0 = s1; s1 = s1 == NST - 1 ? 0 : s1 + 1;
    // H(kt + 1): tile kt + 1 in fb (an odd nk runs one zero tile: past-the-end stages hold zeros)
    STCAT_PL3S_STAGE_LOAD(s0)
    STCAT_PL3S_READ_FRAG(fa, smem + s1 * STAGE)
    STCAT_PL_INTERLEAVE(NMMA, NRD, NDMA)
    STCAT_PL_MMA(fb)
    STCAT_SCHED_FENCE();
    STCAT_WAIT_VMN_LGKM0(NDMA);
    STCAT_S_BARRIER();
    STCAT_SCHED_FENCE();
    s0 = s1; s1 = s1 == NST - 1 ? 0 : s1 + 1;
  }
#undef STCAT_PL3S_STAGE_LOAD
#undef STCAT_PL3S_READ_FRAG
  STCAT_WAIT_VM0_LGKM0();  // past-the-end DMA (zero fill) has landed too: the stages are reused below
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();

#include "igemm_pl_epi.inc"
}
