// "A-stationary" three-plane GEMM for the short-reduction 1x1 convolutions with wide outputs — the bottleneck blocks'
// expand convs and the data gradients of their reduce convs (torchvision resnet101 Bottleneck.conv3 / conv1, call site
// models/vision_model/backbone.py:115-119; K = 64 / 128 / 256 input channels, N = 4 K outputs) — round 5.
//
// Why a second structure.  igemm_pl_fwd_kernel stages BOTH operands of every K-tile through two LDS stages, one 8-wave
// workgroup per CU.  On these layers (profiles/r05_as_kernel_experiments.log):
//   * every K-tile of the activation operand is a cold HBM read with ONE K-tile (~1.3 us) of prefetch distance: the
//     eight K-tiles of a K = 256 tile take ~2.8 us each where the matrix pipe needs 1.3;
//   * the K loop (matrix pipe) and the epilogue (HBM: residual planes in, three planes + bit mask out) of a CU's ONE
//     workgroup alternate, so the launch takes the SUM of its matrix time and its HBM time;
//   * three planes of both operands keep the LDS ports ~85 % busy (fragment reads + DMA writes) next to the MFMAs.
// Here a wave owns 16 pixel rows (v_mfma_f32_16x16x32_bf16) and reads them ONCE, straight into registers, as ready-made
// A fragments: 16 rows x K x 3 planes = 12 (K / 32) VGPRs per lane, 96 at K = 256 — which leaves room for TWO four-wave
// workgroups per CU (<= 256 VGPRs, 58 KB of LDS each).  The two run out of phase on the same SIMDs: one's epilogue
// (VALU + HBM) and activation loads sit under the other's MFMAs.  Only the weights (L2-resident, 1.5 MB per layer) move
// through LDS: a ring of K / 32 slots (one 32-column chunk) that always holds the NEXT chunk's tiles by the time the
// current chunk's last MFMA has issued, so the K loop never waits for memory.  The residual planes / bit mask / per-column
// vectors of a chunk are requested before its K loop and consumed after it; plane stores are fire-and-forget.
//
// Synchronisation is deliberately simple (no counted vmcnt next to stores: loads and stores retire out of order with
// respect to each other): one `vmcnt(0)` + barrier per chunk, placed BEFORE the chunk's stores are issued — it confirms
// the next chunk's weight tiles, which were requested during this chunk's K loop — and one barrier per pair of k-steps
// that frees the two slots just read for the same tiles of the next chunk.
//
// Work split: a workgroup = 64 pixel rows.  The first `par` row blocks (a whole number of rounds of the chip's 2 x CUs
// workgroup slots) take ALL column chunks — their activation rows are read exactly once; the remaining row blocks are cut
// into runs of `k_chunk` chunks so that the last, partial round still fills the chip (layer3: 784 row blocks = 512 + 272 x 4).
#pragma once
#include "igemm_pl.h"

#ifdef STCAT_EMU
#define STCAT_WAIT_LGKM0() ((void)0)
#else
#define STCAT_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

template <int KS>   // k-steps of 32 reduction terms: K = 32 * KS
__global__ void __launch_bounds__(256, 2) igemm_pl_as_kernel(PlParams p) {
  constexpr int NP = 3, TN = 2, BN = TN * 16, NW = 4, BM = NW * 16;
  constexpr int PLANE_B = BN * 64, SLOT = NP * PLANE_B;    // bytes: 32 weight rows x 64 B (one k-step), three planes
  constexpr int LDE = 2 * BN + 4, EPI_WAVE = 16 * LDE * 4;   // epilogue block of a wave: 16 rows x 64 columns (a chunk PAIR)
  static_assert(KS % 2 == 0, "slots are refilled in pairs: 12 one-KiB pieces = 3 per wave");
  static_assert(2 * (KS * SLOT + NW * EPI_WAVE) <= 160 * 1024, "two workgroups per CU");
  STCAT_DYN_SHARED(char, smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, kg = lane >> 4;
  const int wq = STCAT_READFIRSTLANE(wave);
  const IgemmGeom g = p.g;
  const int chunks = p.N / BN;
  int rb, c_first, cpu;
  if ((int)blockIdx.x < p.par) {                     // whole row block
    rb = blockIdx.x; c_first = 0; cpu = chunks;
  } else {                                           // a run of k_chunk chunks; the runs of a row block share an XCD's L2
    cpu = p.k_chunk;
    const int nsplit = chunks / cpu;
    const int j = stcat_xcd_remap((int)blockIdx.x - p.par, (int)gridDim.x - p.par);
    rb = p.par + j / nsplit;
    c_first = (j % nsplit) * cpu;
  }
  const int m0 = rb * BM, nu0 = c_first * BN;
  float* ew = reinterpret_cast<float*>(smem + KS * SLOT + wave * EPI_WAVE);
#ifndef STCAT_EMU
  if (p.stagger > 0 && blockIdx.x < 2u * 256u && (blockIdx.x & 256u)) {
    // Phase offset between the two workgroups that share a CU (experiment): every workgroup of a launch costs the same, so
    // the two start together and stay in step — K loops together (one matrix pipe), epilogues together (one memory
    // path).  The second-slot workgroups of the first round wait half a chunk-pair period once.
    const long t0 = (long)wall_clock64();
    long tt = t0;
    while (tt - t0 < (long)p.stagger) { __builtin_amdgcn_s_sleep(8); tt = (long)wall_clock64(); }
  }
#endif

  // ---- the wave's 16 activation rows as MFMA A fragments: lane (l15, kg) holds row l15, terms 32 s + 8 kg .. + 7
  bf16x8 Ar[NP][KS];
  {
    const int row = m0 + wave * 16 + l15;
    const bool ok = row < p.M && !(p.debug & 512);
    const unsigned v0 = ok ? (unsigned)(row * g.ld + kg * 8) * 2u : STCAT_BUF_OOB;
    STCAT_UNROLL
    for (int pi = 0; pi < NP; ++pi) {
      const stcat_buf_t dA = stcat_make_buf(stcat_plane(p.Ah, p.Al, pi), p.a_bytes);
      STCAT_UNROLL
      for (int s = 0; s < KS; ++s)
        Ar[pi][s] = __builtin_bit_cast(bf16x8, stcat_buf_ld4(dA, ok ? v0 + (unsigned)(s * 64) : STCAT_BUF_OOB, 0u));
    }
  }

  // ---- weight ring: slot s = k-step s of a chunk, [3 planes][32 rows][64 B], 16-byte chunk XOR-ed with (-(row >> 2)) & 3
  // (on the DMA's per-lane SOURCE address and on the fragment read).  A pair of slots = 12 one-KiB pieces (plane, half of
  // the rows), three per wave.
  const __bf16* Bp[NP] = {p.Bh, p.Bl, stcat_plane(p.Bh, p.Bl, 2)};
  int pc_slot[3], pc_plane[3];
  unsigned pc_lds[3], pc_voff[3];
  STCAT_UNROLL
  for (int i = 0; i < 3; ++i) {
    const int r = wq + NW * i, q = r % 6, half = q & 1;
    pc_slot[i] = r / 6;
    pc_plane[i] = q >> 1;
    pc_lds[i] = (unsigned)(pc_slot[i] * SLOT + pc_plane[i] * PLANE_B + half * 1024);
    const int brow = half * 16 + (lane >> 2);
    pc_voff[i] = (unsigned)(brow * p.ldb) * 2u + (unsigned)(((lane & 3) ^ ((0 - (brow >> 2)) & 3)) * 16);
  }
#define STCAT_AS_ISSUE(CC, SP)                                                                          \
  {                                                                                                     \
    const bool live_ = ((CC) < cpu) & !(p.debug & 256);   /* (debug 256: zero-fill weight DMA, timing experiment) */ \
    STCAT_UNROLL                                                                                        \
    for (int i_ = 0; i_ < 3; ++i_) {                                                                    \
      const __bf16* bp_ = pc_plane[i_] == 0 ? Bp[0] : (pc_plane[i_] == 1 ? Bp[1] : Bp[2]);              \
      const unsigned so_ = (unsigned)((nu0 + (CC) * BN) * p.ldb + (2 * (SP) + pc_slot[i_]) * 32) * 2u;  \
      stcat_glds16(stcat_make_buf(bp_, live_ ? p.b_bytes : 0u), smem + (2 * (SP)) * SLOT + pc_lds[i_], pc_voff[i_], so_); \
    }                                                                                                   \
  }
  // (swizzle term (-(row >> 2)) & 3, not (row >> 2) & 3 as in the 32-row kernels: the 16-lane groups of ds_read_b128 —
  //  {0-3, 12-15, 20-27}, ... — hold rows r, r + 12 with k-group kg and rows r + 4, r + 8 with kg + 1 here; this term puts
  //  their four 16-byte chunks into four different bank quarters, the other one collides two and two)
  const unsigned fb0 = (unsigned)(l15 * 64 + ((kg ^ ((0 - (l15 >> 2)) & 3)) * 16));
  struct Frag { bf16x8 b[NP][TN]; };
#define STCAT_AS_READ(F, S)                                                                             \
  STCAT_UNROLL                                                                                          \
  for (int tn = 0; tn < TN; ++tn) {                                                                     \
    STCAT_UNROLL                                                                                        \
    for (int pi_ = 0; pi_ < NP; ++pi_)                                                                  \
      F.b[pi_][tn] = *reinterpret_cast<const bf16x8*>(smem + (S) * SLOT + fb0 + pi_ * PLANE_B + tn * 1024); \
  }
#define STCAT_AS_MMA(F, S, H)                                                                           \
  STCAT_UNROLL                                                                                          \
  for (int pr_ = 0; pr_ < PlProd<NP>::N; ++pr_) {                                                       \
    STCAT_UNROLL                                                                                        \
    for (int tn = 0; tn < TN; ++tn)                                                                     \
      acc[H][tn] = STCAT_MFMA_BF16_16x16x32(Ar[PlProd<NP>::a(pr_)][S], F.b[PlProd<NP>::b(pr_)][tn], acc[H][tn]); \
  }

  // ---- epilogue addressing: the wave's 16 x 64 block (two chunks: 128-byte row segments — 64-byte segments reach only
  // ~2.6 TB/s on the residual reads and the plane stores, measured) through its private LDS block, 8 lanes x 8 columns
  // per row, two passes of 8 rows
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  struct Pre { bf16x8 r[2][NP]; unsigned bits[2]; float4 sc[2], bi[2], ms[2]; };
  const __bf16* Rp[NP] = {p.Rh, p.Rl, stcat_plane(p.Rh, p.Rl, 2)};
  __bf16* Cp[NP] = {p.Ch, p.Cl, stcat_plane(p.Ch, p.Cl, 2)};
  const int em0 = m0 + wave * 16 + erow;
  auto prefetch = [&](Pre& q, int c) {            // c: first chunk of the pair
    const int n = nu0 + c * BN + ecol;
    // per-column epilogue vectors: whole 16-byte loads under wave-uniform branches, requested with the residual
    // (a per-element `p.scale ? p.scale[n + e] : 1` compiles to 24 branch-guarded dword loads behind the K loop)
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    q.sc[0] = q.sc[1] = q.ms[0] = q.ms[1] = one;
    q.bi[0] = q.bi[1] = zero;
    q.bits[0] = q.bits[1] = 0u;
    if (c < cpu) {
      if (p.scale) { q.sc[0] = stcat_ld4(p.scale + n); q.sc[1] = stcat_ld4(p.scale + n + 4); }
      if (p.bias) { q.bi[0] = stcat_ld4(p.bias + n); q.bi[1] = stcat_ld4(p.bias + n + 4); }
      if (p.mscale) { q.ms[0] = stcat_ld4(p.mscale + n); q.ms[1] = stcat_ld4(p.mscale + n + 4); }
      STCAT_UNROLL
      for (int ps = 0; ps < 2; ++ps) {
        const int em = em0 + ps * 8;
        if (em < p.M && !(p.debug & (2 | 32))) {
          if (p.Rh) {
            STCAT_UNROLL
            for (int pi = 0; pi < NP; ++pi) q.r[ps][pi] = STCAT_LOAD_STREAM(reinterpret_cast<const bf16x8*>(Rp[pi] + (long)em * p.ldr + n));
          }
          if (p.Mi) q.bits[ps] = p.Mi[((long)em * p.ldc + n) >> 3];
        }
      }
    }
  };

  f32x4 acc[2][TN];
  STCAT_UNROLL
  for (int h = 0; h < 2; ++h) {
    STCAT_UNROLL
    for (int tn = 0; tn < TN; ++tn) {
      STCAT_UNROLL
      for (int r = 0; r < 4; ++r) acc[h][tn][r] = 0.f;
    }
  }
  STCAT_UNROLL
  for (int sp = 0; sp < KS / 2; ++sp) STCAT_AS_ISSUE(0, sp)
  STCAT_WAIT_VM0();          // (the activation fragments too: the first MFMA needs them anyway)
  STCAT_S_BARRIER();
  STCAT_SCHED_FENCE();
  Pre pre;
  prefetch(pre, 0);
  Frag f0, f1;
  constexpr int N_START = 3 * (KS / 2 - 1), N_LOOP = KS >= 4 ? 3 * (KS / 2 - 2) : 0;
  // K loop of one chunk into accumulator set H.  CNT: the chunk's tiles were requested during the PREVIOUS chunk's K loop
  // with no store in between (the odd chunk of a pair): confirmed by counted waits — N_START / N_LOOP younger
  // DMA pieces of this wave may still be in flight (K = 256: 9 = three pairs of the previous loop, then 6 = what was issued
  // behind the pair that is needed next).  Otherwise (even
  // chunk) the pair's epilogue has confirmed them with its vmcnt(0).
#define STCAT_AS_KLOOP(H, CC, CNT)                                                                      \
  {                                                                                                     \
    if (CNT) { STCAT_WAIT_VM(N_START); STCAT_S_BARRIER(); STCAT_SCHED_FENCE(); }                        \
    STCAT_AS_READ(f0, 0)                                                                                \
    STCAT_UNROLL                                                                                        \
    for (int sp = 0; sp < KS / 2; ++sp) {                                                               \
      STCAT_AS_READ(f1, 2 * sp + 1)                                                                     \
      STCAT_PL_INTERLEAVE(PlProd<NP>::N * TN, NP * TN, 0)                                               \
      STCAT_AS_MMA(f0, 2 * sp, H)                                                                       \
      STCAT_SCHED_FENCE();                                                                              \
      /* every wave has read both slots -> they take the same k-steps of the NEXT chunk */              \
      if (CNT) { STCAT_WAIT_VM(N_LOOP); }                                                               \
      STCAT_WAIT_LGKM0();                                                                               \
      STCAT_S_BARRIER();                                                                                \
      STCAT_SCHED_FENCE();                                                                              \
      STCAT_AS_ISSUE((CC) + 1, sp)                                                                      \
      if (sp + 1 < KS / 2) {                                                                            \
        STCAT_AS_READ(f0, 2 * sp + 2)                                                                   \
        STCAT_PL_INTERLEAVE(PlProd<NP>::N * TN, NP * TN, 3)                                             \
      }                                                                                                 \
      STCAT_AS_MMA(f1, 2 * sp + 1, H)                                                                   \
      STCAT_SCHED_FENCE();                                                                              \
    }                                                                                                   \
  }
  for (int c = 0; c < cpu; c += 2) {
    STCAT_AS_KLOOP(0, c, false)
    STCAT_AS_KLOOP(1, c + 1, true)
    // ---- epilogue of the chunk pair (c, c + 1): 64 columns
    const int n = nu0 + c * BN + ecol;
    STCAT_UNROLL
    for (int h = 0; h < 2; ++h) {
      STCAT_UNROLL
      for (int tn = 0; tn < TN; ++tn) {
        STCAT_UNROLL
        for (int r = 0; r < 4; ++r) {
          ew[(4 * kg + r) * LDE + h * BN + tn * 16 + l15] = acc[h][tn][r];
          acc[h][tn][r] = 0.f;
        }
      }
    }
    // the next pair's first chunk (requested during the odd K loop) and this pair's residual planes have landed; nothing the
    // ring is waiting for is younger than a store from here on
    STCAT_WAIT_VM0_LGKM0();
    STCAT_S_BARRIER();
    STCAT_SCHED_FENCE();
    STCAT_WAVE_LDS_FENCE();
    const float sc[8] = {pre.sc[0].x, pre.sc[0].y, pre.sc[0].z, pre.sc[0].w, pre.sc[1].x, pre.sc[1].y, pre.sc[1].z, pre.sc[1].w};
    const float bi[8] = {pre.bi[0].x, pre.bi[0].y, pre.bi[0].z, pre.bi[0].w, pre.bi[1].x, pre.bi[1].y, pre.bi[1].z, pre.bi[1].w};
    const float ms[8] = {pre.ms[0].x, pre.ms[0].y, pre.ms[0].z, pre.ms[0].w, pre.ms[1].x, pre.ms[1].y, pre.ms[1].z, pre.ms[1].w};
    STCAT_UNROLL
    for (int ps = 0; ps < 2; ++ps) {
      const int em = em0 + ps * 8, row = ps * 8 + erow;
      const float4 v0 = stcat_ld4(&ew[row * LDE + ecol]), v1 = stcat_ld4(&ew[row * LDE + ecol + 4]);
      float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (em < p.M && !((p.debug & 2) && x[0] != 12345.f)) {
        STCAT_UNROLL
        for (int e = 0; e < 8; ++e) x[e] = x[e] * (sc[e] * p.acc_mul) + bi[e];
        if (p.Rh) {
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] += stcat_join1<NP, false>(pre.r[ps], e);
        }
        if (p.relu) {
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (p.drop.thresh) {
          const DropParams dp_ = stcat_drop_resolve(p.drop);
          const unsigned long long i0_ = (unsigned long long)em * (unsigned long long)p.N + (unsigned long long)n;
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] *= stcat_drop_mul(dp_, i0_ + e);
        }
        if (p.Mi) {
          const unsigned bits = pre.bits[ps];
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) x[e] = ((bits >> e) & 1u) ? x[e] * ms[e] : 0.f;
        }
        if (p.Mo) {
          unsigned bits = 0u;
          STCAT_UNROLL
          for (int e = 0; e < 8; ++e) bits |= (x[e] > 0.f ? 1u : 0u) << e;
          p.Mo[((long)em * p.ldc + n) >> 3] = (unsigned char)bits;
        }
        if (p.Ch && !((p.debug & 64) && x[0] != 12345.f)) {
          bf16x8 o8[NP];
          stcat_split8n<NP, false>(x, o8);
          STCAT_UNROLL
          for (int pi = 0; pi < NP; ++pi) STCAT_STORE_STREAM(reinterpret_cast<bf16x8*>(Cp[pi] + (long)em * p.ldc + n), o8[pi]);
        }
        if (p.Cf) {
          stcat_st4(p.Cf + (long)em * p.ldc + n, make_float4(x[0], x[1], x[2], x[3]));
          stcat_st4(p.Cf + (long)em * p.ldc + n + 4, make_float4(x[4], x[5], x[6], x[7]));
        }
      }
    }
    STCAT_WAVE_LDS_FENCE();
    prefetch(pre, c + 2);     // (behind this pair's stores; consumed after the next pair's K loops)
  }
#undef STCAT_AS_KLOOP
#undef STCAT_AS_ISSUE
#undef STCAT_AS_READ
#undef STCAT_AS_MMA
}

