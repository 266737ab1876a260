// extern "C" boundary of libstcat_hip.so (declared in include/stcat_hip.h).
// Plain pointers and sizes only; no torch types.  Built by hipcc for gfx950
// (and, with -DSTCAT_EMU, by host clang against tests/emu/hip_emu.h for index-logic tests).
#include <cstdarg>
#include <cstdio>

#include "../../include/stcat_hip.h"
#include "attention.h"
#include "attention_bs.h"
#include "optim.h"
#include "igemm.h"
#include "igemm_bs.h"
#include "igemm_pl.h"
#include "igemm_pl_as.h"
#include "stem_pl.h"
#include "pointwise.h"
#include "loss.h"

namespace {
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

#ifdef STCAT_EMU
inline int launch_status() { return 0; }
#else
inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "HIP launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
#endif

// byte extent of an operand for the 32-bit buffer loads; 0xFFFFFFFF marks 'too large' (fp32 kernels are used then)
inline unsigned bytes_of(long elems) { return elems * 4 >= 0x7FFFFFFFl ? 0xFFFFFFFFu : (unsigned)(elems * 4); }
inline bool aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int grid_for(long n, int per_block, int cap = 4096) {
  long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// tile choice: prefer the biggest tile that still fills the 256 CUs a few times over
int g_force_bm = 0, g_force_bn = 0;  // stcat_debug_force_tile
void pick_tile(int M, int N, int& BM, int& BN) {
  if (g_force_bm && N % g_force_bn == 0) {
    BM = g_force_bm;
    BN = g_force_bn;
    return;
  }
  const int cands[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  long best = -1;
  BM = 64;
  BN = 64;
  for (auto& c : cands) {
    if (N % c[1] != 0) continue;
    if (c[0] == 128 && M <= 64) continue;
    const long blocks = (long)cdiv(M, c[0]) * (N / c[1]);
    if (blocks >= 600) {
      BM = c[0];
      BN = c[1];
      return;
    }
    if (blocks > best) {
      best = blocks;
      BM = c[0];
      BN = c[1];
    }
  }
}

// 0: fp32 MFMA (exact), 2: split-bf16 x3, 3: split-bf16 x6, 4: split-bf16 x3 with the backbone's tensors pre-split into
// bf16 planes (igemm_pl.h; every other GEMM of the path runs as mode 2)   (stcat_set_mma_mode)
int g_mma_mode_raw = 0;
int g_mma_mode = 0;  // what the fp32-tensor kernels see: mode 4 -> 2, mode 5 -> 3
int g_pl_np = 2;     // planes per tensor of the plane-format entry points: 2 (modes 4, 6) or 3 (mode 5)
int g_pl_f16 = 0;    // mode 6 (f16x3p): the two planes hold IEEE fp16 (22 significand bits), three products on the f16 MFMA
int g_f16_wlog = 6, g_f16_glog = 16;   // its operand scales: weight planes hold w * 2^wlog, gradient planes dy * 2^glog
inline int pl_np_arg() { return g_pl_np | (g_pl_f16 ? 0x100 : 0); }   // what the element-wise plane kernels take as `np`

#define STCAT_TILE_SWITCH(KERNEL, GRID)                                                        \
  if (BM == 128 && BN == 128) {                                                                \
    STCAT_LAUNCH((KERNEL<128, 128>), GRID, dim3(256), 0, st, p);                               \
  } else if (BM == 128) {                                                                      \
    STCAT_LAUNCH((KERNEL<128, 64>), GRID, dim3(256), 0, st, p);                                \
  } else {                                                                                     \
    STCAT_LAUNCH((KERNEL<64, 64>), GRID, dim3(256), 0, st, p);                                 \
  }
#define STCAT_TILE_SWITCH_BS(KERNEL, GRID, NS_)                                                \
  if (BM == 128 && BN == 128) {                                                                \
    STCAT_LAUNCH((KERNEL<128, 128, NS_>), GRID, dim3(256), 0, st, p);                          \
  } else if (BM == 128) {                                                                      \
    STCAT_LAUNCH((KERNEL<128, 64, NS_>), GRID, dim3(256), 0, st, p);                           \
  } else {                                                                                     \
    STCAT_LAUNCH((KERNEL<64, 64, NS_>), GRID, dim3(256), 0, st, p);                            \
  }

// the split-bf16 kernels step K by 32 and need a K step to stay inside one filter tap
inline bool bs_ok(const IgemmParams& p) {
  return g_mma_mode != 0 && p.K % 32 == 0 && p.g.C % 32 == 0 && p.a_bytes != 0xFFFFFFFFu && p.b_bytes != 0xFFFFFFFFu;
}

// ---- stream-K forward (igemm_bs.h): one 8-wave workgroup per CU, equal shares of tiles x K-steps ----------------
int g_streamk = 0;  // 1: stream-K whenever legal, otherwise off   (stcat_debug_streamk)
#ifdef STCAT_EMU
static int sk_workspace(float** ws, int* workers) {
  static float* buf = nullptr;
  *workers = 3;  // a few "CUs": exercises tile tails, whole tiles and tile heads on small shapes
  if (!buf) buf = (float*)malloc((size_t)*workers * 2 * 256 * 128 * sizeof(float));
  *ws = buf;
  return 0;
}
#else
static int sk_workspace(float** ws, int* workers) {
  static float* buf[64] = {};
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail("stream-K: no current device");
  if (!buf[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return fail("stream-K: cannot query the CU count");
    // two partial 256x128 fp32 tiles per worker (64 MB on 256 CUs), allocated once per device
    if (hipMalloc((void**)&buf[dev], (size_t)n * 2 * 256 * 128 * sizeof(float)) != hipSuccess)
      return fail("stream-K: workspace allocation failed");
    cus[dev] = n;
  }
  *ws = buf[dev];
  *workers = cus[dev];
  return 0;
}
#endif

// Use it when tile-per-workgroup scheduling would leave a badly filled last round and the reduction is long enough
// to amortise the partial-tile round trip.
static bool sk_wanted(const IgemmParams& p, int workers) {
  if (g_streamk < 0 || !(bs_ok(p) && g_mma_mode == 2) || p.N % 128 != 0 || g_force_bm) return false;
  const long nt = (long)cdiv(p.M, 256) * (p.N / 128);
  const int nk = p.K / 32;
  if (nt < workers || nk < 2) return false;  // a tile may be split between at most two workers
  // Opt-in only (stcat_debug_streamk(1)).  Measured on the C3 shapes (tools/bench_gemm.py): the partial-tile round
  // trip (64 MB written + read) and the second launch cost 30-40 us; only the layer3 3x3 conv gains in isolation
  // (392 tiles, 72 K-steps: 250 -> 264 TF), shapes with 32 K-steps or >= 3 rounds lose 10-25 %, and in the full
  // step the layer3 gain disappears (84.0 vs 83.5 ms).  An in-kernel fix-up (owner adds the neighbour's partial
  // after a flag) would halve the overhead; it needs cross-XCD release/acquire and is left for a later round.
  return g_streamk > 0;
}

// Skinny launches with a long reduction (FFN of the decoders / temporal encoder on [T(+1),256] states, K = 2048): 4-8 workgroups
// walking 64 K-tiles each is ~35 us of pure latency; split the reduction over grid.z into >= 8-tile slices that
// add into the zeroed output.  Only when the epilogue is bias / residual (no scale, ReLU, mask, second output).
static bool g_acc_output = false;  // set by the *_acc entries around the launch: C already holds the value to add onto
static int skinny_splits(const IgemmParams& p) {
  const int min_k = g_acc_output ? 128 : 1024;   // without the memset launch a 4-K-tile reduction is worth splitting too
  if (!(bs_ok(p) && p.M <= 128 && p.K >= min_k && !p.scale && !p.relu && !p.mask && !p.C2 && p.c_group >= p.M)) return 0;
  const int nk = p.K / 32;
  int splits = g_acc_output ? nk / 2 : nk / 8;
  if (splits > 8) splits = 8;
  return splits >= 2 ? splits : 0;
}
static int skinny_zero(const IgemmParams& p, hipStream_t st) {
  if (g_acc_output) return 0;
  for (int m = 0; m < (p.ldc == p.N ? 1 : p.M); ++m) {  // dense output: one memset, else row by row
    const size_t bytes = (p.ldc == p.N ? (size_t)p.M * p.N : (size_t)p.N) * sizeof(float);
    if (hipMemsetAsync(p.C + (size_t)m * p.ldc, 0, bytes, st) != hipSuccess) return fail("split-K: memset failed");
  }
  return 0;
}

int launch_pl_fwd_f32(const IgemmParams& p, hipStream_t st);
bool pl_f32_ok(const IgemmParams& p);

int launch_fwd(const IgemmParams& p, hipStream_t st) {
  if (g_mma_mode == 0 && pl_f32_ok(p)) return launch_pl_fwd_f32(p, st);  // exact-fp32 mode on the LDS-DMA structure
  if (const int splits = skinny_splits(p)) {
    IgemmParams q = p;
    q.k_chunk = cdiv(p.K / 32, splits);
    if (int rc = skinny_zero(p, st)) return rc;
    const dim3 grid(cdiv(p.M, 64) * (p.N / 64), 1, cdiv(p.K / 32, q.k_chunk));
    if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_fwd_kernel<64, 64, 3>), grid, dim3(256), 0, st, q); }
    else { STCAT_LAUNCH((igemm_bs_fwd_kernel<64, 64, 2>), grid, dim3(256), 0, st, q); }
    return launch_status();
  }
  {
    float* ws = nullptr;
    int workers = 0;
#ifdef STCAT_EMU
    const bool probe = g_streamk > 0;
#else
    const bool probe = g_streamk > 0 && bs_ok(p) && g_mma_mode == 2 && !g_force_bm && p.N % 128 == 0 &&
                       (long)cdiv(p.M, 256) * (p.N / 128) >= 128;  // (no workspace allocation for small problems)
#endif
    if (probe) {
      if (int rc = sk_workspace(&ws, &workers)) return rc;
      if (sk_wanted(p, workers)) {
        IgemmParams q = p;
        q.sk_ws = ws;
        STCAT_LAUNCH((igemm_bs_fwd_sk_kernel<256, 128, 2, 8>), dim3(workers), dim3(512), 0, st, q);
        if (int rc = launch_status()) return rc;
        STCAT_LAUNCH((igemm_bs_fwd_sk_fixup_kernel<256, 128>), dim3(workers), dim3(256), 0, st, q, workers);
        return launch_status();
      }
    }
  }
  int BM, BN;
  pick_tile(p.M, p.N, BM, BN);
  // Less than one round of 128x128 tiles on the 512 resident slots and a long reduction (layer4's convs): one
  // 8-wave 256x128 workgroup per CU is 20-25 % faster there (tools/bench_gemm.py --tile 256x128); elsewhere the
  // two-workgroup 128x128 form wins.
  if (!g_force_bm && p.N % 128 == 0 && p.K >= 1024 && (long)cdiv(p.M, 128) * (p.N / 128) < 512 &&
      (long)cdiv(p.M, 256) * (p.N / 128) >= 160) {  // ... and still a workgroup for most of the 256 CUs
    BM = 256;
    BN = 128;
  }
  if (BM == 256 && !(bs_ok(p) && g_mma_mode == 2)) pick_tile(p.M, p.N, BM, BN);  // 8-wave tile: bf16x3 only (LDS)
  if (BM == 256 && !(bs_ok(p) && g_mma_mode == 2)) BM = 128;                       // (forced by the debug hook)
  if (BM == 256) {  // 8-wave 256x128 tile, one workgroup per CU
    STCAT_LAUNCH((igemm_bs_fwd_kernel<256, 128, 2, 8>), dim3(cdiv(p.M, 256) * (p.N / 128)), dim3(512), 0, st, p);
    return launch_status();
  }
  const dim3 grid(cdiv(p.M, BM) * (p.N / BN));
  if (bs_ok(p)) {
    if (g_mma_mode == 3) { STCAT_TILE_SWITCH_BS(igemm_bs_fwd_kernel, grid, 3) } else { STCAT_TILE_SWITCH_BS(igemm_bs_fwd_kernel, grid, 2) }
  } else {
    STCAT_TILE_SWITCH(igemm_fwd_kernel, grid)
  }
  return launch_status();
}

int launch_dgrad(const IgemmParams& p, hipStream_t st) {
  if (const int splits = skinny_splits(p)) {
    IgemmParams q = p;
    q.k_chunk = cdiv(p.K / 32, splits);
    if (int rc = skinny_zero(p, st)) return rc;
    const dim3 grid(cdiv(p.M, 64) * (p.N / 64), 1, cdiv(p.K / 32, q.k_chunk));
    if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_dgrad_kernel<64, 64, 3>), grid, dim3(256), 0, st, q); }
    else { STCAT_LAUNCH((igemm_bs_dgrad_kernel<64, 64, 2>), grid, dim3(256), 0, st, q); }
    return launch_status();
  }
  int BM, BN;
  pick_tile(p.M, p.N, BM, BN);
  if (BM == 256) BM = 128;
  const dim3 grid(cdiv(p.M, BM) * (p.N / BN));
  if (bs_ok(p)) {
    if (g_mma_mode == 3) { STCAT_TILE_SWITCH_BS(igemm_bs_dgrad_kernel, grid, 3) } else { STCAT_TILE_SWITCH_BS(igemm_bs_dgrad_kernel, grid, 2) }
  } else {
    STCAT_TILE_SWITCH(igemm_dgrad_kernel, grid)
  }
  return launch_status();
}

// rows = output rows (Cout / N_lin), cols = output columns (KH*KW*Cin / K_lin), red = reduction length (pixels)
int launch_wgrad(IgemmParams p, int rows, int cols, int red, hipStream_t st) {
  const bool bs = g_mma_mode != 0 && p.a_bytes != 0xFFFFFFFFu && p.b_bytes != 0xFFFFFFFFu;
  // 8-wave 256x128 tile, ONE workgroup per CU and at most one round of them: the most efficient configuration
  // measured for long reductions (tools/bench_quant.py); needs Cout % 256 == 0 (layer3/4, the FFN)
  const bool big8 = bs && g_mma_mode == 2 && (rows % 256 == 0) && (p.g.C % 128 == 0) &&
                    (g_force_bm == 256 || (!g_force_bm && red >= 4096));
  const bool big = (rows % 128 == 0) && (p.g.C % 128 == 0) && g_force_bm != 64;
  const int BM = big8 ? 256 : (big ? 128 : 64), BN = big8 ? 128 : (big ? 128 : 64);
  const int tiles = (rows / BM) * (cols / BN);
  // split the reduction so that the grid is at most two full rounds of the resident slots (512 four-wave or
  // 256 eight-wave workgroups): rounding UP here put 1044 blocks on the layer3 3x3 conv — a third, almost empty round
  const int slots = big8 ? 256 : 1024;
  int nsplit = slots / tiles;
  const int max_split = cdiv(red, 256);  // at least 8 K-tiles per workgroup: shorter loops are all prologue + atomics
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  int chunk = cdiv(red, nsplit);
  chunk = ((chunk + 31) / 32) * 32;
  nsplit = cdiv(red, chunk);
  p.M = rows;
  p.N = cols;
  p.K = red;
  p.k_chunk = chunk;
  const dim3 grid(tiles, 1, nsplit);
  if (big8) {
    STCAT_LAUNCH((igemm_bs_wgrad_kernel<256, 128, 2, 8>), grid, dim3(512), 0, st, p);
  } else if (bs) {
    if (big) {
      if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_wgrad_kernel<128, 128, 3>), grid, dim3(256), 0, st, p); }
      else { STCAT_LAUNCH((igemm_bs_wgrad_kernel<128, 128, 2>), grid, dim3(256), 0, st, p); }
    } else {
      if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_wgrad_kernel<64, 64, 3>), grid, dim3(256), 0, st, p); }
      else { STCAT_LAUNCH((igemm_bs_wgrad_kernel<64, 64, 2>), grid, dim3(256), 0, st, p); }
    }
  } else if (big) {
    STCAT_LAUNCH((igemm_wgrad_kernel<128, 128>), grid, dim3(256), 0, st, p);
  } else {
    STCAT_LAUNCH((igemm_wgrad_kernel<64, 64>), grid, dim3(256), 0, st, p);
  }
  return launch_status();
}

// ---- plane-format GEMMs (igemm_pl.h) ---------------------------------------------------------------------------
template <class K>
int pl_prepare(K kernel, int lds_bytes) {
#ifndef STCAT_EMU
  static bool done = false;  // per instantiation: opt in to > 64 KB of dynamic LDS once
  if (!done) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return fail("plane GEMM: cannot reserve %d bytes of LDS", lds_bytes);
    done = true;
  }
#endif
  return 0;
}
#define STCAT_PL_LAUNCH(KERNEL, BM_, BN_, WM_, WN_, GRID)                                              \
  {                                                                                                    \
    constexpr int lds_ = 4 * (BM_ + BN_) * 64;                                                         \
    if (int rc_ = pl_prepare(KERNEL<BM_, BN_, WM_, WN_>, lds_)) return rc_;                            \
    STCAT_LAUNCH((KERNEL<BM_, BN_, WM_, WN_>), GRID, dim3(512), lds_, st, p);                          \
  }
// fp16-plane (mode 6) instantiations of the two-plane tiles
#define STCAT_PLH_FWD(BM_, BN_, WM_, WN_, GRID)                                                                 \
  {                                                                                                            \
    constexpr int lds_ = 4 * (BM_ + BN_) * 64;                                                                 \
    if (int rc_ = pl_prepare(igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, false, 2, 8, true>, lds_)) return rc_;    \
    STCAT_LAUNCH((igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, false, 2, 8, true>), GRID, dim3(512), lds_, st, p);  \
  }
#define STCAT_PLH_WGRAD(BM_, BN_, WM_, WN_, GRID)                                                      \
  {                                                                                                    \
    constexpr int lds_ = 4 * (BM_ + BN_) * 64;                                                         \
    if (int rc_ = pl_prepare(igemm_pl_wgrad_kernel<BM_, BN_, WM_, WN_, 2, true>, lds_)) return rc_;    \
    STCAT_LAUNCH((igemm_pl_wgrad_kernel<BM_, BN_, WM_, WN_, 2, true>), GRID, dim3(512), lds_, st, p);  \
  }
// three-plane (mode 5) instantiations: 6 x (BM + BN) x 64 bytes of LDS for the two stages
#define STCAT_PL3_FWD_NW(BM_, BN_, WM_, WN_, NW_, GRID)                                                         \
  {                                                                                                            \
    constexpr int lds_ = 6 * (BM_ + BN_) * 64;                                                                 \
    if (int rc_ = pl_prepare(igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, false, 3, NW_>, lds_)) return rc_;        \
    STCAT_LAUNCH((igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, false, 3, NW_>), GRID, dim3(64 * NW_), lds_, st, p); \
  }
#define STCAT_PL3_FWD(BM_, BN_, WM_, WN_, GRID) STCAT_PL3_FWD_NW(BM_, BN_, WM_, WN_, 8, GRID)
#define STCAT_PL3_WGRAD(BM_, BN_, WM_, WN_, GRID)                                                      \
  {                                                                                                    \
    constexpr int lds_ = 6 * (BM_ + BN_) * 64;                                                         \
    if (int rc_ = pl_prepare(igemm_pl_wgrad_kernel<BM_, BN_, WM_, WN_, 3>, lds_)) return rc_;          \
    STCAT_LAUNCH((igemm_pl_wgrad_kernel<BM_, BN_, WM_, WN_, 3>), GRID, dim3(512), lds_, st, p);        \
  }

int g_pl_debug = 0;   // stcat_debug_pl_flags (timing experiments)
int g_pl_as = 1;      // stcat_debug_pl_flags bit 7 (128) switches the A-stationary kernel off (A/B against the two-stage kernel)
int g_pl_force = -1;  // stcat_debug_force_pl_tile: index into the tile table below, -1 = heuristic
struct PlTile { int bm, bn; float eff; };
// relative cost per MAC of each tile shape (bigger wave tiles amortise fragment reads and DMA issue better)
// (224 x 256: 7 x 32 rows, 8 waves side by side — 50176 = 224 * 224 rows of layer3 fill 224 of 256 CUs in ONE round)
// (128 x 64, index 6: three planes only — 4 waves, 74 KB of LDS, two workgroups per CU; chosen by rule, not by cost)
const PlTile kPlTiles[7] = {{256, 256, 1.00f}, {256, 128, 1.12f}, {128, 256, 1.12f}, {128, 128, 1.30f}, {256, 64, 1.35f},
                            {224, 256, 1.04f}, {128, 64, 1.60f}};

// mode 5 (three planes): two stages of 3 x (BM + BN) x 64 bytes must fit 160 KB -> BM + BN <= 384; the six-term
// contraction issues 24 MFMAs per k-step against 12 fragment reads on the 64 x 64 wave tile of the 256 x 128 shapes
// (128 x 256 reads each A row block once instead of twice when N = 256 and measured 3-5 % ahead of 256 x 128 on every
// layer3 shape — step-like operands: 3x3 forward 0.380 -> 0.363 ms, 1024 -> 256 forward 0.195 -> 0.185 ms)
const float kPl3Eff[7] = {0.f, 1.00f, 0.97f, 1.15f, 1.20f, 0.f, 1.60f};   // indexed like kPlTiles; 0 = not available
int g_pl3_small = 0;   // stcat_debug_pl_flags bit 2 sets it: the two-workgroup 128 x 64 tile for short reductions
int pick_pl3_tile(int M, int N, int K) {
  if (g_pl_force >= 0 && g_pl_force < 7) {
    int f = g_pl_force;
    if (kPl3Eff[f] == 0.f) f = 1;                         // 256x256 / 224x256 do not exist with three planes
    if (N % kPlTiles[f].bn == 0) return f;
  }
  // Short reductions (K <= 512: the 1x1 expand / reduce convs and their data gradients): the epilogue's HBM traffic
  // (residual planes in, three planes + bit mask out) takes as long as the K loop, and with ONE workgroup per CU nothing
  // overlaps them.  The 4-wave 128 x 64 tile lets two workgroups share the CU — MEASURED NEUTRAL in isolation (256 -> 1024
  // forward 0.264 vs 0.267 ms, data gradient 0.198 vs 0.195, step-like operands) and 3.4 ms SLOWER per C3 step (its
  // 32 x 64 wave tiles read LDS twice as often per MFMA and the launches that run beside the weight-gradient stream lose
  // more than the overlap gains; profiles/r03_tile6_experiment.log): opt-in through stcat_debug_pl_flags(4) only.
  if (g_pl3_small && K <= 512 && N % 64 == 0 && (long)M * N >= (1l << 22)) return 6;
  int best = -1;
  float best_cost = 0.f;
  for (int i = 1; i <= 4; ++i) {
    const PlTile& tl = kPlTiles[i];
    if (N % tl.bn != 0) continue;
    const long tiles = (long)cdiv(M, tl.bm) * (N / tl.bn);
    const long rounds = (tiles + 255) / 256;
    const float cost = (float)rounds * tl.bm * tl.bn * kPl3Eff[i];
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  return best;
}

int pick_pl_tile(int M, int N, int K) {
  if (g_pl_force >= 0 && g_pl_force < 7 && N % kPlTiles[g_pl_force].bn == 0) return g_pl_force;
  // Short reductions (K <= 512: the 1x1 expand / reduce convs and their data gradients) and N = 128 are bound by the
  // epilogue's HBM traffic (residual / mask in, planes out), not by the matrix pipe.  The 128 x 128 tile needs 64 KB of
  // LDS, so TWO workgroups share a CU and one's epilogue overlaps the other's loads: measured with step-like operands
  // (tools/bench_gemm.py --step-like, profiles/r02_plane_gemm_steplike.log) 0.220 -> 0.186 ms (256 -> 1024 forward),
  // 0.251 -> 0.204 ms (1024 -> 256 data gradient), 0.583 -> 0.472 ms (layer1 64 -> 256).
  if (N % 128 == 0 && (K <= 512 || N == 128) && (long)M * N >= (1l << 22)) return 3;
  int best = -1;
  float best_cost = 0.f;
  for (int i = 0; i < 6; ++i) {
    const PlTile& tl = kPlTiles[i];
    if (N % tl.bn != 0) continue;
    const long tiles = (long)cdiv(M, tl.bm) * (N / tl.bn);
    const long rounds = (tiles + 255) / 256;  // one 8-wave workgroup per CU
    const float cost = (float)rounds * tl.bm * tl.bn * tl.eff;
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  return best;
}

// ---- A-stationary form (igemm_pl_as.h): 1x1, stride 1, K = 64 / 128 / 256, three bf16 planes ---------------------
int g_pl_as_skew = 0;    // stcat_debug_pl_flags bit 0x2000: flags >> 16 = the A-stationary kernel's phase offset in 10 ns ticks
bool pl_as_ok(const PlParams& p) {
  if (!g_pl_as || g_pl_np != 3 || g_pl_f16) return false;
  if (g_pl_force >= 0 && g_pl_force != 7) return false;            // a test / experiment asked for a tile of the table
  const IgemmGeom& g = p.g;
  if (g.KH != 1 || g.KW != 1 || g.mul != 1 || g.div != 1 || g.off != 0 || p.par) return false;
  if (!(p.K == 64 || p.K == 128 || p.K == 256) || p.K != g.C || p.N % 64 != 0 || p.N < 4 * 64) return false;
  if (p.C2h || p.Rf || (p.Yh && !p.Mi) || p.radd_div > 1 || p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return false;
  if ((long)p.M * g.ld * 2 >= 0x7FFFFFFFl) return false;          // 32-bit byte offsets of the fragment loads
  return g_pl_force == 7 || (long)p.M * p.N >= (1l << 22);         // (small problems: the tile kernel's finer grid)
}
static int wg_slots_as() {   // workgroup slots of the chip for the A-stationary kernel: two per CU
#ifdef STCAT_EMU
  return 4;
#else
  static int slots = 0;
  if (!slots) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      slots = 2 * n;
    else
      slots = 512;
  }
  return slots;
#endif
}
template <int KS>
int launch_pl_as_ks(PlParams& p, hipStream_t st) {
  constexpr int lds_min = KS * 3 * 32 * 64 + 4 * 16 * 68 * 4;
  // (experiment 0x1000: ONE workgroup per CU — the LDS request is raised past half the CU's 160 KB)
  const int lds = ((g_pl_debug & 0x1000) && !(g_pl_debug & 8)) ? 96 * 1024 : lds_min;
  if (int rc = pl_prepare(igemm_pl_as_kernel<KS>, 96 * 1024)) return rc;
  // resident workgroups per CU: 2 at K = 256 (196 VGPRs), 3 at K = 128 (148), 4 at K = 64 (124)
  const int rbs = cdiv(p.M, 64), chunks = p.N / 32, slots = wg_slots_as() / 2 * (KS == 8 ? 2 : (KS == 4 ? 3 : 4));
  // whole rounds of the chip as whole row blocks (activation rows read once); the partial last round in quarter runs
  int cpu = (chunks % 8 == 0) ? chunks / 4 : ((chunks % 4 == 0) ? chunks / 2 : chunks);   // (even: the epilogue works on chunk pairs)
  // Rounds are counted on HALF the chip's slots: in the step these launches run beside another stream's (the second
  // forward chain, the weight-gradient stream).  Measured at C3, same box, 10 steps each: whole-chip rounds 82.5 / 82.4 ms,
  // "less than a round = whole row blocks" 82.7 / 82.2, half-chip rounds 82.2 / 82.0 (the tile kernel: 84.1).
  int tier1 = (rbs / (slots / 2)) * (slots / 2);
  if ((g_pl_debug & 0x400) && !(g_pl_debug & 8)) tier1 = 0;                          // (experiment: every row block in runs)
  if ((g_pl_debug & 0x800) && !(g_pl_debug & 8)) tier1 = (rbs / slots) * slots;      // (experiment: whole-chip rounds)
  if (rbs - tier1 == 0) cpu = chunks;
  p.par = tier1;
  p.k_chunk = cpu;
  p.stagger = g_pl_as_skew;      // 10 ns ticks; second-slot workgroups of the first round (igemm_pl_as.h)
  const int grid = tier1 + (rbs - tier1) * (chunks / cpu);
  STCAT_LAUNCH((igemm_pl_as_kernel<KS>), dim3(grid), dim3(256), lds, st, p);
  return launch_status();
}
int launch_pl_as(PlParams& p, hipStream_t st) {
  if (p.K == 256) return launch_pl_as_ks<8>(p, st);
  if (p.K == 128) return launch_pl_as_ks<4>(p, st);
  return launch_pl_as_ks<2>(p, st);
}

int launch_pl_fwd(const PlParams& p_, hipStream_t st) {
  PlParams p = p_;
  p.debug = (g_pl_debug & 8) ? (g_pl_debug & 0xff) : (g_pl_debug & 0x3ff);   // (bits 8.. are the stagger ticks when bit 3 is set)
  p.stagger = (g_pl_debug & 8) ? (g_pl_debug >> 8) : 0;
  // mode f16x3p: the B operand (weight planes, plain or transposed) carries 2^wlog; gradient planes keep their 2^glog
  p.acc_mul = g_pl_f16 ? ldexpf(1.f, -g_f16_wlog) : 1.f;
  if (p.K % 32 != 0 || p.g.C % 32 != 0) return fail("plane GEMM: K and the channel count must be multiples of 32");
  if (p.g.div != 1 && p.g.div != 2 && p.g.div != 4) return fail("plane GEMM: stride must be 1, 2 or 4");
  if ((p.Ch == nullptr) != (p.Cl == nullptr)) return fail("plane GEMM: output planes go together");
  if (pl_as_ok(p)) return launch_pl_as(p, st);
  // stride-2 data gradient on even dims: parity-class row order (igemm_pl.h, PlParams::par): 4 x tiles(M / 4)
  p.par = (p.g.div == 2 && p.g.mul == 1 && p.g.sgn == -1 && p.g.OH % 2 == 0 && p.g.OW % 2 == 0 && !(g_pl_debug & 4)) ? 1 : 0;
  const int Mrows = p.par ? p.M / 4 : p.M;
  const int ti = g_pl_np == 3 ? pick_pl3_tile(p.par ? p.M : Mrows, p.N, p.par ? p.K / 4 : p.K) : pick_pl_tile(p.par ? p.M : Mrows, p.N, p.par ? p.K / 4 : p.K);
  if (ti < 0) return fail("plane GEMM: N = %d is not a multiple of 64", p.N);
  const int BM = kPlTiles[ti].bm, BN = kPlTiles[ti].bn;
  const dim3 grid((p.par ? 4 : 1) * cdiv(Mrows, BM) * (p.N / BN));
  if (g_pl_np == 3) {
    switch (ti) {
      case 1: STCAT_PL3_FWD(256, 128, 4, 2, grid) break;
      case 2: STCAT_PL3_FWD(128, 256, 2, 4, grid) break;
      case 3: STCAT_PL3_FWD(128, 128, 2, 4, grid) break;
      case 6: STCAT_PL3_FWD_NW(128, 64, 2, 2, 4, grid) break;
      default: STCAT_PL3_FWD(256, 64, 8, 1, grid) break;
    }
    return launch_status();
  }
  if (g_pl_f16) {
    switch (ti) {
      case 0: STCAT_PLH_FWD(256, 256, 2, 4, grid) break;
      case 1: STCAT_PLH_FWD(256, 128, 4, 2, grid) break;
      case 2: STCAT_PLH_FWD(128, 256, 2, 4, grid) break;
      case 3: STCAT_PLH_FWD(128, 128, 2, 4, grid) break;
      case 5: STCAT_PLH_FWD(224, 256, 1, 8, grid) break;
      default: STCAT_PLH_FWD(256, 64, 8, 1, grid) break;
    }
    return launch_status();
  }
  switch (ti) {
    case 0: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 256, 256, 2, 4, grid) break;
    case 1: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 256, 128, 4, 2, grid) break;
    case 2: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 128, 256, 2, 4, grid) break;
    case 3: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 128, 128, 2, 4, grid) break;
    case 5: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 224, 256, 1, 8, grid) break;
    default: STCAT_PL_LAUNCH(igemm_pl_fwd_kernel, 256, 64, 8, 1, grid) break;
  }
  return launch_status();
}

// ---- exact-fp32 mode through the same kernel (template parameter F32): the fp32 tensors are addressed in units of
// half a float, so gather, DMA, swizzle and fragment reads are the plane kernel's; the K-step of 32 units is 16 fp32
// reduction terms on v_mfma_f32_32x32x2_f32.  Conv forward, conv / Linear data gradient on pre-transposed weights and
// the large Linear layers of the encoder come through here in mma mode "f32".
bool pl_f32_ok(const IgemmParams& p) {
  return !g_force_bm && p.M >= 256 && p.g.C % 16 == 0 && p.N % 64 == 0 && p.K % 16 == 0 && p.c_group >= p.M &&
         p.a_bytes != 0xFFFFFFFFu && p.b_bytes != 0xFFFFFFFFu && p.ldc % 4 == 0 && p.ldb % 4 == 0 && p.g.ld % 4 == 0 &&
         (!p.res || p.ldr % 4 == 0) && aligned16(p.A) && aligned16(p.B) && aligned16(p.C) && (!p.res || aligned16(p.res)) &&
         (!p.mask || aligned16(p.mask)) && (!p.C2 || aligned16(p.C2)) && !p.sk_ws;
}

int launch_pl_fwd_f32(const IgemmParams& s, hipStream_t st) {
  PlParams p = {};
  p.Ah = reinterpret_cast<const __bf16*>(s.A); p.Bh = reinterpret_cast<const __bf16*>(s.B);
  p.Cf = s.C; p.scale = s.scale; p.bias = s.bias; p.Rf = s.res; p.Yf = s.mask; p.mscale = s.mscale;
  p.C2f = s.C2; p.c2scale = s.c2scale; p.acc_mul = 1.f;
  p.M = s.M; p.N = s.N; p.K = 2 * s.K; p.ldb = 2 * s.ldb; p.ldc = s.ldc; p.ldr = s.ldr; p.relu = s.relu;
  p.a_bytes = s.a_bytes; p.b_bytes = s.b_bytes;
  p.b_tap_stride = s.b_tap_stride / 2;          // bytes -> half-float units
  p.g = s.g;
  p.g.C = 2 * s.g.C; p.g.ld = 2 * s.g.ld;
  stcat_fastdiv_magic(p.g.OW, &p.g.mg_ow, &p.g.sh_ow);
  stcat_fastdiv_magic(p.g.OH * p.g.OW, &p.g.mg_ohw, &p.g.sh_ohw);
  p.debug = g_pl_debug;
  // tile: the rounds x area cost model (the fp32 pipe is 5x slower per product than bf16x3: every shape is matrix-bound)
  int ti = -1;
  float best = 0.f;
  for (int i = 0; i < 6; ++i) {
    const PlTile& tl = kPlTiles[i];
    if (p.N % tl.bn != 0) continue;
    const long tiles = (long)cdiv(p.M, tl.bm) * (p.N / tl.bn);
    // (224 x 256: its 1 x 8 wave layout re-reads the A fragments 8x — irrelevant here, 4 fp32 MFMAs = 256 cycles per
    // fragment pair; what counts is that 50176 = 224 * 224 rows of layer3 fill 224 of the 256 CUs in one round)
    const float cost = (float)((tiles + 255) / 256) * tl.bm * tl.bn * (i == 5 ? 1.0f : tl.eff);
    if (ti < 0 || cost < best) { ti = i; best = cost; }
  }
  if (g_pl_force >= 0 && g_pl_force < 6 && p.N % kPlTiles[g_pl_force].bn == 0) ti = g_pl_force;
  if (ti < 0) return fail("fp32 plane GEMM: N = %d is not a multiple of 64", p.N);
  const int BM = kPlTiles[ti].bm, BN = kPlTiles[ti].bn;
  p.par = (p.g.div == 2 && p.g.mul == 1 && p.g.sgn == -1 && p.g.OH % 2 == 0 && p.g.OW % 2 == 0 && !(g_pl_debug & 4)) ? 1 : 0;
  const dim3 grid((p.par ? 4 : 1) * cdiv(p.par ? p.M / 4 : p.M, BM) * (p.N / BN));
#define STCAT_PLF_LAUNCH(BM_, BN_, WM_, WN_)                                                          \
  {                                                                                                    \
    constexpr int lds_ = 4 * (BM_ + BN_) * 64;                                                         \
    if (int rc_ = pl_prepare(igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, true>, lds_)) return rc_;         \
    STCAT_LAUNCH((igemm_pl_fwd_kernel<BM_, BN_, WM_, WN_, true>), grid, dim3(512), lds_, st, p);       \
  }
  switch (ti) {
    case 0: STCAT_PLF_LAUNCH(256, 256, 2, 4) break;
    case 1: STCAT_PLF_LAUNCH(256, 128, 4, 2) break;
    case 2: STCAT_PLF_LAUNCH(128, 256, 2, 4) break;
    case 3: STCAT_PLF_LAUNCH(128, 128, 2, 4) break;
    case 5: STCAT_PLF_LAUNCH(224, 256, 1, 8) break;
    default: STCAT_PLF_LAUNCH(256, 64, 8, 1) break;
  }
#undef STCAT_PLF_LAUNCH
  return launch_status();
}

// rows = Cout, cols = taps * Cin, red = pixels
int launch_pl_wgrad(PlParams p, int rows, int cols, int red, hipStream_t st, float* ws = nullptr, long ws_floats = 0) {
  if (rows % 128 != 0 || p.g.C % 128 != 0) return fail("plane wgrad: need Cout, Cin %% 128 == 0 (%d, %d)", rows, p.g.C);
  const int BM = (rows % 256 == 0 && g_pl_force != 3) ? 256 : 128;
  // (three planes: BM + BN <= 384, the 256 x 256 tile does not fit)
  const int BN = (p.g.C % 256 == 0 && g_pl_force != 3 && !(g_pl_np == 3 && BM == 256)) ? 256 : 128;
  const int tiles = (rows / BM) * (cols / BN);
  int nsplit = 256 / tiles;                 // one round of one workgroup per CU
  // The tap tiles of a slice share dY and the (shifted) X pixels through their XCD's 4 MB L2 — if the slice fits.  At
  // layer2 (200704 pixels, 128 channels) a 1/28 slice is 7.3 MB of operands and nothing is shared (PMC: 1.65 GB read
  // per launch, the sum of all workgroups' private reads); two rounds of half-size slices: 0.256 -> 0.173 ms.
  // (Single-tap layers have no such sharing and only pay the extra atomics: 0.131 -> 0.152 ms, so they keep one round.)
  // (Only when a slice's tiles sit on ONE XCD, <= 32 of them: layer4's 36 tiles per slice span XCDs and lose, 0.173 -> 0.212.)
  if (p.g.KH * p.g.KW > 1 && tiles <= 32 && nsplit >= 1 && (double)cdiv(red, nsplit) * (rows + p.g.C) * 4.0 > 4.5e6) nsplit *= 2;
  const int max_split = cdiv(red, 512);     // at least 16 K-tiles per workgroup
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  int chunk = cdiv(red, nsplit);
  chunk = ((chunk + 31) / 32) * 32;
  nsplit = cdiv(red, chunk);
  p.M = rows; p.N = cols; p.K = red; p.k_chunk = chunk; p.debug = g_pl_debug;
  p.acc_mul = g_pl_f16 ? ldexpf(1.f, -g_f16_glog) : 1.f;     // mode f16x3p: dY planes hold dy * 2^glog, X planes are unscaled
  const dim3 grid(tiles, 1, nsplit);
  // atomics-free form: the slices store partial tiles, one more launch sums them in order (bit-reproducible)
  const long out_floats = (long)rows * p.ldc;
  const bool use_ws = ws && nsplit > 1 && (long)nsplit * out_floats <= ws_floats && out_floats % 4 == 0 && aligned16(ws) &&
                      aligned16(p.Wf) && !(g_pl_debug & 0x4000);
  p.Ws = use_ws ? ws : nullptr;
  auto finish = [&]() -> int {
    if (int rc = launch_status()) return rc;
    if (!use_ws) return 0;
    const long n4 = out_floats / 4;
    STCAT_LAUNCH(pl_wgrad_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, (const float*)ws, p.Wf, n4,
                 nsplit, out_floats);
    return launch_status();
  };
  if (g_pl_f16) {
    if (BM == 256 && BN == 256) STCAT_PLH_WGRAD(256, 256, 2, 4, grid)
    else if (BM == 256) STCAT_PLH_WGRAD(256, 128, 4, 2, grid)
    else if (BN == 256) STCAT_PLH_WGRAD(128, 256, 2, 4, grid)
    else STCAT_PLH_WGRAD(128, 128, 2, 4, grid)
    return finish();
  }
  if (g_pl_np == 3) {
    if (BM == 256) STCAT_PL3_WGRAD(256, 128, 4, 2, grid)
    else if (BN == 256) STCAT_PL3_WGRAD(128, 256, 2, 4, grid)
    else STCAT_PL3_WGRAD(128, 128, 2, 4, grid)
    return finish();
  }
  if (BM == 256 && BN == 256) STCAT_PL_LAUNCH(igemm_pl_wgrad_kernel, 256, 256, 2, 4, grid)
  else if (BM == 256) STCAT_PL_LAUNCH(igemm_pl_wgrad_kernel, 256, 128, 4, 2, grid)
  else if (BN == 256) STCAT_PL_LAUNCH(igemm_pl_wgrad_kernel, 128, 256, 2, 4, grid)
  else STCAT_PL_LAUNCH(igemm_pl_wgrad_kernel, 128, 128, 2, 4, grid)
  return finish();
}

inline unsigned plane_bytes(long elems) { return elems * 2 >= 0x7FFFFFFFl ? 0xFFFFFFFFu : (unsigned)(elems * 2); }

IgemmGeom conv_geom_fwd(int H, int W, int C, int ld, int OH, int OW, int KH, int KW, int stride, int pad) {
  IgemmGeom g;
  g.H = H; g.W = W; g.C = C; g.ld = ld; g.OH = OH; g.OW = OW; g.KH = KH; g.KW = KW;
  g.mul = stride; g.off = -pad; g.sgn = 1; g.div = 1;
  stcat_fastdiv_magic(OW, &g.mg_ow, &g.sh_ow);
  stcat_fastdiv_magic(OH * OW, &g.mg_ohw, &g.sh_ohw);
  return g;
}
}  // namespace

int g_stem_pl = 1;    // stcat_debug_pl_flags bit 0x8000 clear: the six-product modes run the LDS-staged bf16 stem (stem_pl.h)
// the stem of the six-product modes (3: bf16x6, 5: bf16x6p): persistent workgroups, one per CU, over 16 x 16 output tiles
template <bool U8>
static int launch_stem_pl(const void* frames, const float* w, const float* in_scale, const float* in_shift, const float* scale,
                          const float* bias, float* y, int n, int H, int W, int OH, int OW, hipStream_t st) {
  constexpr int lds = 3 * 37 * 48 * 4 + 3 * 64 * 184 * 2;
  if (int rc = pl_prepare(stem_pl_kernel<U8>, lds)) return rc;
  StemPlParams q = {};
  q.A = frames; q.w = w; q.scale = scale; q.bias = bias; q.in_scale = in_scale; q.in_shift = in_shift; q.y = y;
  q.n = n; q.H = H; q.W = W; q.OH = OH; q.OW = OW;
  q.tiles_x = cdiv(OW, 16); q.tiles_y = cdiv(OH, 16); q.total = n * q.tiles_x * q.tiles_y;
  const int slots = wg_slots_as() / 2;      // one 8-wave workgroup per CU
  STCAT_LAUNCH(stem_pl_kernel<U8>, dim3(q.total < slots ? q.total : slots), dim3(512), lds, st, q);
  return launch_status();
}
static bool stem_pl_mode() { return g_stem_pl && (g_mma_mode_raw == 3 || g_mma_mode_raw == 5); }


extern "C" {

int stcat_version(void) { return 100; }
int stcat_set_mma_mode(int mode) {
  if (mode != 0 && mode != 2 && mode != 3 && mode != 4 && mode != 5 && mode != 6)
    return fail("set_mma_mode: mode must be 0 (f32), 2 (bf16x3), 3 (bf16x6), 4 (bf16x3 on two bf16 planes), 5 (bf16x6 on three) "
                "or 6 (f16x3 on two fp16 planes)");
  g_mma_mode_raw = mode;
  g_mma_mode = mode == 4 ? 2 : ((mode == 5 || mode == 6) ? 3 : mode);   // (modes 5 / 6: every other GEMM as bf16x6)
  g_pl_np = mode == 5 ? 3 : 2;
  g_pl_f16 = mode == 6 ? 1 : 0;
  return 0;
}
int stcat_get_mma_mode(void) { return g_mma_mode_raw; }
int stcat_set_f16_scales(int weight_log2, int grad_log2) {
  if (weight_log2 < 0 || weight_log2 > 14 || grad_log2 < 0 || grad_log2 > 30) return fail("set_f16_scales: exponents out of range");
  g_f16_wlog = weight_log2;
  g_f16_glog = grad_log2;
  return 0;
}
int stcat_get_f16_scale(int which) { return which == 0 ? g_f16_wlog : g_f16_glog; }

int stcat_debug_force_tile(int bm, int bn) {
  const bool ok = (bm == 0 && bn == 0) || (bm == 128 && bn == 128) || (bm == 128 && bn == 64) || (bm == 64 && bn == 64) ||
                  (bm == 256 && bn == 128);
  if (!ok) return fail("debug_force_tile: unsupported tile %dx%d", bm, bn);
  g_force_bm = bm;
  g_force_bn = bn;
  return 0;
}
int stcat_debug_streamk(int mode) {
  if (mode < -1 || mode > 1) return fail("debug_streamk: mode must be -1, 0 or 1");
  g_streamk = mode;
  return 0;
}
const char* stcat_last_error(void) { return g_err; }

__global__ void __launch_bounds__(64) spin_kernel(long ticks, int* sink) {
#ifndef STCAT_EMU
  const long t0 = (long)wall_clock64();     // constant-rate counter, 100 MHz on gfx950
  long t = t0;
  while (t - t0 < ticks) t = (long)wall_clock64();
  if (sink && t == 0x7fffffffffffffffl) *sink = 1;   // (keeps the loop alive under optimisation)
#endif
}
int stcat_spin(int microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return fail("spin: 0 .. 100000 us");
  STCAT_LAUNCH(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)microseconds * 100, (int*)nullptr);
  return launch_status();
}

// A HIP stream for a background lane of the step (round 6: the next clip's frozen prefix under the grounding section).
// priority: -1 = the device's greatest, 0 = default, 1 = the device's least (the command processor serves the queues of a
// higher class first, so a least-priority lane gives way to the dependent chains it runs under); cus > 0: restrict the
// stream to `cus` compute units (hipExtStreamCreateWithCUMask: the first `cus` bits of the mask) — the chains always find
// free CUs, and the lane draws proportionally less HBM bandwidth.  A CU mask and a priority cannot be combined (HIP has no
// such constructor): cus wins.  Returns the hipStream_t in *out; the caller owns it (stcat_stream_destroy).
int stcat_stream_create(int priority, int cus, void** out) {
  if (!out) return fail("stream_create: out is null");
  *out = nullptr;
#ifdef STCAT_EMU
  (void)priority; (void)cus;
  return 0;                      // (the emulator has no streams: every launch is synchronous)
#else
  hipStream_t st = nullptr;
  if (cus > 0) {
    if (cus > 1024) return fail("stream_create: cus = %d", cus);
    uint32_t mask[32] = {0};
    for (int i = 0; i < cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    if (hipExtStreamCreateWithCUMask(&st, 32, mask) != hipSuccess) return fail("stream_create: hipExtStreamCreateWithCUMask failed");
  } else {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return fail("stream_create: no priority range");
    const int pr = priority < 0 ? greatest : (priority > 0 ? least : 0);
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr) != hipSuccess) return fail("stream_create: hipStreamCreateWithPriority failed");
  }
  *out = (void*)st;
  return 0;
#endif
}

int stcat_stream_destroy(void* stream) {
#ifndef STCAT_EMU
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return fail("stream_destroy failed");
#endif
  return 0;
}

int stcat_frozen_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                         float* bias, int C, float eps, void* stream) {
  if (C <= 0) return fail("frozen_bn_fold: C=%d", C);
  STCAT_LAUNCH(frozen_bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, w, b, rm, rv, scale,
               bias, C, eps);
  return launch_status();
}

int stcat_stem_fwd(const float* frames, const float* w, const float* scale, const float* bias, float* y, int n,
                   int H, int W, void* stream) {
  if (n <= 0 || H < 7 || W < 7) return fail("stem_fwd: bad shape n=%d H=%d W=%d", n, H, W);
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  if (stem_pl_mode())
    return launch_stem_pl<false>(frames, w, nullptr, nullptr, scale, bias, y, n, H, W, OH, OW, (hipStream_t)stream);
  IgemmParams p = {};
  p.A = frames; p.B = w; p.C = y; p.scale = scale; p.bias = bias; p.res = nullptr;
  p.M = n * OH * OW; p.N = 64; p.K = 147; p.ldb = 147; p.ldc = 64; p.ldr = 0;
  p.c_group = p.M; p.c_group_stride = 0; p.relu = 1;
  p.g = conv_geom_fwd(H, W, 3, 0, OH, OW, 7, 7, 2, 3);
  STCAT_LAUNCH(igemm_stem_kernel<false>, dim3(cdiv(p.M, 128)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

int stcat_stem_u8_fwd(const unsigned char* frames_hwc, const float* w, const float* in_scale, const float* in_shift,
                      const float* scale, const float* bias, float* y, int n, int H, int W, void* stream) {
  if (n <= 0 || H < 7 || W < 7) return fail("stem_u8_fwd: bad shape n=%d H=%d W=%d", n, H, W);
  if (!in_scale || !in_shift) return fail("stem_u8_fwd: the per-channel input scale / shift are required");
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  if (stem_pl_mode())
    return launch_stem_pl<true>(frames_hwc, w, in_scale, in_shift, scale, bias, y, n, H, W, OH, OW, (hipStream_t)stream);
  IgemmParams p = {};
  p.A = reinterpret_cast<const float*>(frames_hwc); p.B = w; p.C = y; p.scale = scale; p.bias = bias; p.res = nullptr;
  p.mscale = in_scale; p.c2scale = in_shift;
  p.M = n * OH * OW; p.N = 64; p.K = 147; p.ldb = 147; p.ldc = 64; p.ldr = 0;
  p.c_group = p.M; p.c_group_stride = 0; p.relu = 1;
  p.g = conv_geom_fwd(H, W, 3, 0, OH, OW, 7, 7, 2, 3);
  STCAT_LAUNCH(igemm_stem_kernel<true>, dim3(cdiv(p.M, 128)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

int stcat_maxpool3x3s2(const float* x, float* y, int n, int H, int W, int C, void* stream) {
  if (C % 4 != 0 || !aligned16(x) || !aligned16(y)) return fail("maxpool: C %% 4 != 0 or unaligned");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)n * OH * OW * (C / 4);
  STCAT_LAUNCH(maxpool3x3s2_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, n, H,
               W, C, OH, OW);
  return launch_status();
}

int stcat_conv_fwd(const float* x, const float* w, const float* scale, const float* bias, const float* res,
                   float* y, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   int relu, void* stream) {
  if (Cin % 16 != 0 || Cout % 64 != 0) return fail("conv_fwd: need Cin %% 16 == 0 and Cout %% 64 == 0 (%d, %d)", Cin, Cout);
  if (!aligned16(x) || !aligned16(w)) return fail("conv_fwd: operands must be 16-byte aligned");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  IgemmParams p = {};
  p.A = x; p.B = w; p.C = y; p.scale = scale; p.bias = bias; p.res = res;
  p.a_bytes = bytes_of((long)n * H * W * Cin); p.b_bytes = bytes_of((long)Cout * KH * KW * Cin);
  p.b_tap_stride = (unsigned)Cin * 4;
  p.M = n * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout;
  p.c_group = p.M; p.c_group_stride = 0; p.relu = relu;
  p.g = conv_geom_fwd(H, W, Cin, Cin, OH, OW, KH, KW, stride, pad);
  return launch_fwd(p, (hipStream_t)stream);
}

int stcat_conv_dgrad(const float* g, const float* w, const float* add, const float* mask_y, const float* mask_scale,
                     float* dx, float* dx2, const float* dx2_scale, const float* wt, int n, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, void* stream) {
  if ((dx2 != nullptr) != (dx2_scale != nullptr)) return fail("conv_dgrad: dx2 and dx2_scale go together");
  if (Cout % 16 != 0 || Cin % 64 != 0) return fail("conv_dgrad: need Cout %% 16 == 0 and Cin %% 64 == 0 (%d, %d)", Cout, Cin);
  if (!aligned16(g) || !aligned16(w)) return fail("conv_dgrad: operands must be 16-byte aligned");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  IgemmParams p = {};
  p.A = g; p.B = w; p.C = dx; p.res = add; p.mask = mask_y; p.mscale = mask_scale; p.C2 = dx2; p.c2scale = dx2_scale;
  p.a_bytes = bytes_of((long)n * OH * OW * Cout); p.b_bytes = bytes_of((long)Cout * KH * KW * Cin);
  p.M = n * H * W; p.N = Cin; p.K = KH * KW * Cout; p.ldb = KH * KW * Cin; p.ldc = Cin; p.ldr = Cin;
  p.c_group = p.M; p.relu = 0;
  // gathered tensor = g [n,OH,OW,Cout]; rows enumerate input pixels (hi, wi): ho = (hi + pad - kh) / stride
  IgemmGeom q;
  q.H = OH; q.W = OW; q.C = Cout; q.ld = Cout; q.OH = H; q.OW = W; q.KH = KH; q.KW = KW;
  q.mul = 1; q.off = pad; q.sgn = -1; q.div = stride;
  p.g = q;
  if (wt && ((g_mma_mode != 0 && bs_ok(p)) || g_mma_mode == 0)) {  // transposed weights [tap][Cin][Cout]: the forward
    IgemmParams t = p;                                              // kernel's staging path
    t.B = wt; t.ldb = Cout; t.b_tap_stride = (unsigned)((long)Cin * Cout * 4);
    t.c_group = t.M;
    if (g_mma_mode != 0 || pl_f32_ok(t)) return launch_fwd(t, (hipStream_t)stream);
  }
  return launch_dgrad(p, (hipStream_t)stream);
}

int stcat_conv_wgrad(const float* g, const float* x, float* dw, int n, int H, int W, int Cin, int Cout, int KH,
                     int KW, int stride, int pad, void* stream) {
  if (Cout % 64 != 0 || Cin % 64 != 0) return fail("conv_wgrad: need Cout, Cin %% 64 == 0 (%d, %d)", Cout, Cin);
  if (!aligned16(g) || !aligned16(x)) return fail("conv_wgrad: operands must be 16-byte aligned");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  IgemmParams p = {};
  p.A = g; p.B = x; p.C = dw; p.ldb = Cout; p.ldc = KH * KW * Cin;
  p.a_bytes = bytes_of((long)n * OH * OW * Cout); p.b_bytes = bytes_of((long)n * H * W * Cin);
  p.g = conv_geom_fwd(H, W, Cin, Cin, OH, OW, KH, KW, stride, pad);
  return launch_wgrad(p, Cout, KH * KW * Cin, n * OH * OW, (hipStream_t)stream);
}

int stcat_weight_transpose(const float* w, float* wt, int Cout, int taps, int Cin, void* stream) {
  if (Cout <= 0 || taps <= 0 || Cin <= 0) return fail("weight_transpose: bad shape");
  STCAT_LAUNCH(weight_transpose_kernel, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(256), 0, (hipStream_t)stream, w,
               wt, Cout, taps, Cin);
  return launch_status();
}

int stcat_weight_transpose_entry_bytes(void) { return (int)sizeof(WtEntry); }

int stcat_weight_transpose_multi(const void* table, int n_entries, int total_blocks, void* stream) {
  if (n_entries <= 0 || total_blocks <= 0) return fail("weight_transpose_multi: empty table");
  STCAT_LAUNCH(weight_transpose_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
               (const WtEntry*)table, n_entries);
  return launch_status();
}

int stcat_act_bwd(const float* dy, const float* y, const float* scale, float* G, float* dres, long n, int C,
                  int relu, void* stream) {
  if (n % 4 != 0 || C % 4 != 0) return fail("act_bwd: n and C must be multiples of 4");
  STCAT_LAUNCH(act_bwd_kernel, dim3(grid_for(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dy, y, scale, G,
               dres, n / 4, C, relu);
  return launch_status();
}

int stcat_pos_sine_2d(const unsigned char* mask, const float* dimt, float* pos, int n, int h, int w, void* stream) {
  const long total = (long)n * h * w * 256;
  STCAT_LAUNCH(pos_sine_2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, mask, dimt, pos, n,
               h, w);
  return launch_status();
}

int stcat_sine_embed_fwd(const float* anchor, const float* dimt, float* out, int M, void* stream) {
  STCAT_LAUNCH(sine_embed_fwd_kernel, dim3(grid_for((long)M * 512, 256)), dim3(256), 0, (hipStream_t)stream, anchor,
               dimt, out, M);
  return launch_status();
}

int stcat_sine_embed_bwd(const float* anchor, const float* dimt, const float* dout, float* danchor, int M,
                         void* stream) {
  STCAT_LAUNCH(sine_embed_bwd_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, anchor, dimt, dout,
               danchor, M);
  return launch_status();
}

int stcat_linear_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                     int K, int ldx, int ldy, int ldr, int relu, int c_group, long c_group_stride, void* stream) {
  if (N % 64 != 0 || K % 16 != 0) return fail("linear_fwd: need N %% 64 == 0, K %% 16 == 0 (N=%d K=%d)", N, K);
  if (ldx % 4 != 0 || !aligned16(x) || !aligned16(w)) return fail("linear_fwd: x/w must be 16-byte aligned rows");
  if (M <= 0) return fail("linear_fwd: M=%d", M);
  IgemmParams p = {};
  p.A = x; p.B = w; p.C = y; p.scale = nullptr; p.bias = bias; p.res = res;
  p.a_bytes = bytes_of((long)(M - 1) * ldx + K); p.b_bytes = bytes_of((long)N * K);
  p.b_tap_stride = (unsigned)K * 4;
  p.M = M; p.N = N; p.K = K; p.ldb = K; p.ldc = ldy; p.ldr = ldr;
  p.c_group = c_group > 0 ? c_group : M;
  p.c_group_stride = (int)c_group_stride;
  p.relu = relu;
  p.g = conv_geom_fwd(1, 1, K, ldx, 1, 1, 1, 1, 1, 0);
  return launch_fwd(p, (hipStream_t)stream);
}

int stcat_linear_dgrad(const float* g, const float* w, const float* add, const float* wt, float* dx, int M, int N,
                       int K, int ldg, int lddx, void* stream) {
  if (K % 64 != 0 || N % 16 != 0) return fail("linear_dgrad: need K %% 64 == 0, N %% 16 == 0 (N=%d K=%d)", N, K);
  if (ldg % 4 != 0 || !aligned16(g) || !aligned16(w)) return fail("linear_dgrad: g/w must be 16-byte aligned rows");
  IgemmParams p = {};
  p.A = g; p.B = w; p.C = dx; p.res = add;
  p.a_bytes = bytes_of((long)(M - 1) * ldg + N); p.b_bytes = bytes_of((long)N * K);
  p.M = M; p.N = K; p.K = N; p.ldb = K; p.ldc = lddx; p.ldr = lddx;
  p.c_group = M; p.relu = 0;
  IgemmGeom q;
  q.H = 1; q.W = 1; q.C = N; q.ld = ldg; q.OH = 1; q.OW = 1; q.KH = 1; q.KW = 1;
  q.mul = 1; q.off = 0; q.sgn = -1; q.div = 1;
  p.g = q;
  if (wt && ((g_mma_mode != 0 && bs_ok(p)) || g_mma_mode == 0)) {  // wt = w^T [K][N]
    IgemmParams t = p;
    t.B = wt; t.ldb = N; t.b_tap_stride = 0;
    if (g_mma_mode != 0 || pl_f32_ok(t)) return launch_fwd(t, (hipStream_t)stream);
  }
  return launch_dgrad(p, (hipStream_t)stream);
}

// y += x w^T + bias (+ res), dx += g w (+ add): the output already holds the value to accumulate onto (zeros from the
// caller's zeroed arena in the decoders).  For the skinny launches of the decoders (M <= 128) the reduction can then be
// split over grid.z without a memset launch in front; other shapes are refused (no beta = 1 epilogue in the big tiles).
int stcat_linear_fwd_acc(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                         int K, int ldx, int ldy, int ldr, void* stream) {
  if (M > 128 || K < 128 || K % 64 != 0 || g_mma_mode == 0)
    return fail("linear_fwd_acc: the accumulate form serves M <= 128, K >= 128, K %% 64 == 0 in the split-bf16 modes (M=%d K=%d)", M, K);
  g_acc_output = true;
  const int rc = stcat_linear_fwd(x, w, bias, res, y, M, N, K, ldx, ldy, ldr, 0, 0, 0, stream);
  g_acc_output = false;
  return rc;
}

int stcat_linear_dgrad_acc(const float* g, const float* w, const float* add, float* dx, int M, int N, int K, int ldg,
                           int lddx, void* stream) {
  if (M > 128 || N < 128 || N % 64 != 0 || g_mma_mode == 0)
    return fail("linear_dgrad_acc: the accumulate form serves M <= 128, N >= 128, N %% 64 == 0 in the split-bf16 modes (M=%d N=%d)", M, N);
  g_acc_output = true;
  const int rc = stcat_linear_dgrad(g, w, add, nullptr, dx, M, N, K, ldg, lddx, stream);
  g_acc_output = false;
  return rc;
}

// y = dropout_p(relu?(x w^T + bias (+ res))): the FFN's first Linear with its ReLU and dropout in ONE epilogue
// (modal_encoder.py:239-240, query_decoder.py:435-436, 657-658: `dropout(activation(linear1(x)))`).  Element (m, n) of
// the site draws counter drop_offset + m * N + n, exactly what stcat_dropout draws on the dense [M, N] tensor.
int stcat_linear_fwd_drop(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                          int K, int ldx, int ldy, int ldr, int relu, float drop_p, long drop_seed, long drop_offset,
                          const long* drop_base, void* stream) {
  if (N % 64 != 0 || K % 32 != 0) return fail("linear_fwd_drop: need N %% 64 == 0, K %% 32 == 0 (N=%d K=%d)", N, K);
  if (ldx % 4 != 0 || !aligned16(x) || !aligned16(w)) return fail("linear_fwd_drop: x/w must be 16-byte aligned rows");
  if (M <= 0 || ldy != N) return fail("linear_fwd_drop: dense output rows expected (M=%d ldy=%d N=%d)", M, ldy, N);
  if (g_mma_mode == 0) return fail("linear_fwd_drop: the fused form runs on the split-bf16 kernels (not in mma mode f32)");
  IgemmParams p = {};
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.res = res;
  p.a_bytes = bytes_of((long)(M - 1) * ldx + K); p.b_bytes = bytes_of((long)N * K);
  p.b_tap_stride = (unsigned)K * 4;
  p.M = M; p.N = N; p.K = K; p.ldb = K; p.ldc = ldy; p.ldr = ldr;
  p.c_group = M; p.relu = relu;
  p.scale = nullptr;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  p.g = conv_geom_fwd(1, 1, K, ldx, 1, 1, 1, 1, 1, 0);
  if (!bs_ok(p)) return fail("linear_fwd_drop: operands too large for the bounds-checked loads");
  // (launched directly: the split-K form of launch_fwd has no place for a non-linear epilogue)
  int BM, BN;
  pick_tile(p.M, p.N, BM, BN);
  if (BM == 256) BM = 128;
  const dim3 grid(cdiv(p.M, BM) * (p.N / BN));
  hipStream_t st = (hipStream_t)stream;
  if (g_mma_mode == 3) { STCAT_TILE_SWITCH_BS(igemm_bs_fwd_kernel, grid, 3) } else { STCAT_TILE_SWITCH_BS(igemm_bs_fwd_kernel, grid, 2) }
  return launch_status();
}

// dx = [mask_y > 0] * mask_gain * (g w (+ add)): the data gradient of a Linear whose INPUT was `mask_y = dropout(relu(.))`
// — ReLU backward and dropout backward in the epilogue (mask_y > 0 <=> the ReLU passed AND the element was kept;
// mask_gain = 1 / (1 - p)).  wt = w^T [K, N] (optional: the transposed-operand kernel, as in stcat_linear_dgrad).
int stcat_linear_dgrad_mask(const float* g, const float* w, const float* add, const float* wt, const float* mask_y,
                            float mask_gain, float* dx, int M, int N, int K, int ldg, int lddx, void* stream) {
  if (K % 64 != 0 || N % 32 != 0) return fail("linear_dgrad_mask: need K %% 64 == 0, N %% 32 == 0 (N=%d K=%d)", N, K);
  if (ldg % 4 != 0 || !aligned16(g) || !aligned16(w)) return fail("linear_dgrad_mask: g/w must be 16-byte aligned rows");
  if (!mask_y || lddx != K) return fail("linear_dgrad_mask: mask_y and a dense dx are required");
  if (g_mma_mode == 0) return fail("linear_dgrad_mask: the fused form runs on the split-bf16 kernels (not in mma mode f32)");
  IgemmParams p = {};
  p.A = g; p.B = w; p.C = dx; p.res = add; p.mask = mask_y; p.mask_gain = mask_gain;
  p.a_bytes = bytes_of((long)(M - 1) * ldg + N); p.b_bytes = bytes_of((long)N * K);
  p.M = M; p.N = K; p.K = N; p.ldb = K; p.ldc = lddx; p.ldr = lddx;
  p.c_group = M; p.relu = 0;
  IgemmGeom q;
  q.H = 1; q.W = 1; q.C = N; q.ld = ldg; q.OH = 1; q.OW = 1; q.KH = 1; q.KW = 1;
  q.mul = 1; q.off = 0; q.sgn = -1; q.div = 1;
  p.g = q;
  if (!bs_ok(p)) return fail("linear_dgrad_mask: operands too large for the bounds-checked loads");
  if (wt) {  // wt = w^T [K][N]: the forward kernel's staging path
    IgemmParams t = p;
    t.B = wt; t.ldb = N; t.b_tap_stride = 0;
    return launch_fwd(t, (hipStream_t)stream);
  }
  return launch_dgrad(p, (hipStream_t)stream);
}

// ---- several independent skinny Linear problems of ONE shape in one launch (igemm_bs.h: IgemmMulti) ------------------
// Accumulating forms (the outputs hold the value to add onto: zeros from the caller's arena), M <= 128, split-bf16 modes.
// Problems may share an output (y = sum_j x_j w_j^T + b_j): the slices add atomically.
static int multi_check(const char* what, int n, int M, int N, int K) {
  if (n < 1 || n > 8) return fail("%s: 1 .. 8 problems (n=%d)", what, n);
  if (M <= 0 || M > 128 || N % 64 != 0 || K % 64 != 0) return fail("%s: need M <= 128, N %% 64 == 0, K %% 64 == 0 (M=%d N=%d K=%d)", what, M, N, K);
  if (g_mma_mode == 0) return fail("%s: split-bf16 modes only", what);
  return 0;
}
static int multi_splits(int nk, int tiles, int n) {   // reduction slices: >= 2 K-tiles each, at most ~512 workgroups in all
  int splits = nk / 2;
  if (splits > 8) splits = 8;
  while (splits > 1 && (long)tiles * n * splits > 512) --splits;
  return splits < 1 ? 1 : splits;
}
int stcat_linear_fwd_multi(int n, const float* x0, const float* x1, const float* x2, const float* x3, const float* x4, const float* x5, const float* x6, const float* x7, const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5, const float* w6, const float* w7, const float* b0, const float* b1, const float* b2, const float* b3, const float* b4, const float* b5, const float* b6, const float* b7,
                           float* y0, float* y1, float* y2, float* y3, float* y4, float* y5, float* y6, float* y7, int M, int N, int K, void* stream) {
  if (int rc = multi_check("linear_fwd_multi", n, M, N, K)) return rc;
  IgemmMulti mp = {};
  const float* xs[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
  const float* ws[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
  const float* bs[8] = {b0, b1, b2, b3, b4, b5, b6, b7};
  float* ys[8] = {y0, y1, y2, y3, y4, y5, y6, y7};
  for (int j = 0; j < n; ++j) {
    if (!xs[j] || !ws[j] || !ys[j] || !aligned16(xs[j]) || !aligned16(ws[j])) return fail("linear_fwd_multi: problem %d: null / unaligned operand", j);
    mp.A[j] = xs[j]; mp.B[j] = ws[j]; mp.bias[j] = bs[j]; mp.C[j] = ys[j];
  }
  IgemmParams& p = mp.base;
  p.a_bytes = bytes_of((long)M * K); p.b_bytes = bytes_of((long)N * K);
  p.b_tap_stride = (unsigned)K * 4;
  p.M = M; p.N = N; p.K = K; p.ldb = K; p.ldc = N; p.ldr = 0; p.c_group = M; p.relu = 0;
  p.g = conv_geom_fwd(1, 1, K, K, 1, 1, 1, 1, 1, 0);
  const int tiles = cdiv(M, 64) * (N / 64), splits = multi_splits(K / 32, tiles, n);
  p.k_chunk = cdiv(K / 32, splits);
  const dim3 grid(tiles, n, cdiv(K / 32, p.k_chunk));
  hipStream_t st = (hipStream_t)stream;
  if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_fwd_multi_kernel<3>), grid, dim3(256), 0, st, mp); }
  else { STCAT_LAUNCH((igemm_bs_fwd_multi_kernel<2>), grid, dim3(256), 0, st, mp); }
  return launch_status();
}

// dx_j += g_j . w_j (+ add_j): the data gradients of such a group (outputs may coincide: d_x = sum_j g_j w_j)
int stcat_linear_dgrad_multi(int n, const float* g0, const float* g1, const float* g2, const float* g3, const float* g4, const float* g5, const float* g6, const float* g7, const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5, const float* w6, const float* w7, const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5, const float* a6, const float* a7,
                             float* d0, float* d1, float* d2, float* d3, float* d4, float* d5, float* d6, float* d7, int M, int N, int K, void* stream) {
  if (int rc = multi_check("linear_dgrad_multi", n, M, N, K)) return rc;
  IgemmMulti mp = {};
  const float* gs[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
  const float* ws[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
  const float* as[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
  float* ds[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
  for (int j = 0; j < n; ++j) {
    if (!gs[j] || !ws[j] || !ds[j] || !aligned16(gs[j]) || !aligned16(ws[j])) return fail("linear_dgrad_multi: problem %d: null / unaligned operand", j);
    mp.A[j] = gs[j]; mp.B[j] = ws[j]; mp.bias[j] = as[j]; mp.C[j] = ds[j];
  }
  IgemmParams& p = mp.base;          // as stcat_linear_dgrad: rows = M, columns = K (the Linear's inputs), reduction = N
  p.a_bytes = bytes_of((long)M * N); p.b_bytes = bytes_of((long)N * K);
  p.M = M; p.N = K; p.K = N; p.ldb = K; p.ldc = K; p.ldr = K; p.c_group = M; p.relu = 0;
  IgemmGeom q;
  q.H = 1; q.W = 1; q.C = N; q.ld = N; q.OH = 1; q.OW = 1; q.KH = 1; q.KW = 1;
  q.mul = 1; q.off = 0; q.sgn = -1; q.div = 1;
  p.g = q;
  const int tiles = cdiv(M, 64) * (K / 64), splits = multi_splits(N / 32, tiles, n);
  p.k_chunk = cdiv(N / 32, splits);
  const dim3 grid(tiles, n, cdiv(N / 32, p.k_chunk));
  hipStream_t st = (hipStream_t)stream;
  if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_dgrad_multi_kernel<3>), grid, dim3(256), 0, st, mp); }
  else { STCAT_LAUNCH((igemm_bs_dgrad_multi_kernel<2>), grid, dim3(256), 0, st, mp); }
  return launch_status();
}

// dw_j += g_j^T x_j, db_j += column sums of g_j: the weight gradients of such a group (N, K % 128 == 0)
int stcat_linear_wgrad_multi(int n, const float* g0, const float* g1, const float* g2, const float* g3, const float* g4, const float* g5, const float* g6, const float* g7, const float* x0, const float* x1, const float* x2, const float* x3, const float* x4, const float* x5, const float* x6, const float* x7, float* dw0, float* dw1, float* dw2, float* dw3, float* dw4, float* dw5, float* dw6, float* dw7,
                             float* db0, float* db1, float* db2, float* db3, float* db4, float* db5, float* db6, float* db7, int M, int N, int K, void* stream) {
  if (n < 1 || n > 8) return fail("linear_wgrad_multi: 1 .. 8 problems (n=%d)", n);
  if (M <= 0 || N % 128 != 0 || K % 128 != 0 || g_mma_mode == 0)
    return fail("linear_wgrad_multi: need N, K %% 128 == 0 in a split-bf16 mode (M=%d N=%d K=%d)", M, N, K);
  IgemmMulti mp = {};
  const float* gs[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
  const float* xs[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
  float* dws[8] = {dw0, dw1, dw2, dw3, dw4, dw5, dw6, dw7};
  float* dbs[8] = {db0, db1, db2, db3, db4, db5, db6, db7};
  for (int j = 0; j < n; ++j) {
    if (!gs[j] || !xs[j] || !dws[j] || !aligned16(gs[j]) || !aligned16(xs[j])) return fail("linear_wgrad_multi: problem %d: null / unaligned operand", j);
    mp.A[j] = gs[j]; mp.B[j] = xs[j]; mp.C[j] = dws[j]; mp.rowsum[j] = dbs[j];
  }
  IgemmParams& p = mp.base;          // as stcat_linear_wgrad / launch_wgrad: rows = N, columns = K, reduction = M
  p.ldb = N; p.ldc = K;
  p.a_bytes = bytes_of((long)M * N); p.b_bytes = bytes_of((long)M * K);
  p.g = conv_geom_fwd(1, 1, K, K, 1, 1, 1, 1, 1, 0);
  p.M = N; p.N = K; p.K = M;
  const int tiles = (N / 128) * (K / 128);
  int nsplit = cdiv(M, 256);
  if (nsplit < 1) nsplit = 1;
  int chunk = ((cdiv(M, nsplit) + 31) / 32) * 32;
  p.k_chunk = chunk;
  const dim3 grid(tiles, n, cdiv(M, chunk));
  hipStream_t st = (hipStream_t)stream;
  if (g_mma_mode == 3) { STCAT_LAUNCH((igemm_bs_wgrad_multi_kernel<128, 3>), grid, dim3(256), 0, st, mp); }
  else { STCAT_LAUNCH((igemm_bs_wgrad_multi_kernel<128, 2>), grid, dim3(256), 0, st, mp); }
  return launch_status();
}

int stcat_colsum(const float* a, const float* b, float* out, int M, int N, void* stream);

int stcat_linear_wgrad(const float* g, const float* x, float* dw, float* db, int M, int N, int K, int ldg, int ldx,
                       void* stream) {
  if (N % 64 != 0 || K % 64 != 0) return fail("linear_wgrad: need N, K %% 64 == 0 (N=%d K=%d)", N, K);
  if (ldg % 4 != 0 || ldx % 4 != 0 || !aligned16(g) || !aligned16(x)) return fail("linear_wgrad: unaligned");
  IgemmParams p = {};
  p.A = g; p.B = x; p.C = dw; p.ldb = ldg; p.ldc = K;
  p.a_bytes = bytes_of((long)(M - 1) * ldg + N); p.b_bytes = bytes_of((long)(M - 1) * ldx + K);
  p.g = conv_geom_fwd(1, 1, K, ldx, 1, 1, 1, 1, 1, 0);
  if (db) {
    // split-bf16 kernels sum dY's columns from the operand registers they stage anyway; the fp32 kernels
    // (reference mode) run the separate column-sum launch
    if (g_mma_mode != 0 && p.a_bytes != 0xFFFFFFFFu && p.b_bytes != 0xFFFFFFFFu && ldg == N) p.rowsum = db;
    else if (ldg != N) return fail("linear_wgrad: db needs a dense dY (ldg == N)");
    else if (int rc = stcat_colsum(g, nullptr, db, M, N, stream)) return rc;
  }
  return launch_wgrad(p, N, K, M, (hipStream_t)stream);
}

int stcat_small_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                           void* stream) {
  if (N > 16 || K % 4 != 0) return fail("small_linear_fwd: need N <= 16, K %% 4 == 0 (N=%d K=%d)", N, K);
  STCAT_LAUNCH(small_linear_fwd_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N,
               K);
  return launch_status();
}

int stcat_small_linear_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                           int N, int K, void* stream) {
  if (N > 16) return fail("small_linear_bwd: N=%d > 16", N);
  if (dx) {
    STCAT_LAUNCH(small_linear_dx_kernel, dim3(grid_for((long)M * K, 256)), dim3(256), 0, (hipStream_t)stream, g, w, dx,
                 M, N, K);
  }
  if (dw) {
    STCAT_LAUNCH(small_linear_dw_kernel, dim3(cdiv(K, 32), N), dim3(256), 0, (hipStream_t)stream, g, x, dw, db, M, N,
                 K);
  }
  return launch_status();
}

int stcat_colsum(const float* a, const float* b, float* out, int M, int N, void* stream) {
  if (M <= 0 || N <= 0) return fail("colsum: M=%d N=%d", M, N);
  int rows = cdiv(M, 512);
  if (rows < 8) rows = 8;
  STCAT_LAUNCH(colsum_kernel, dim3(cdiv(M, rows), cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, M, N,
               rows);
  return launch_status();
}

int stcat_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, int M, int D, float eps, float drop_p, long drop_seed,
                        long drop_offset, const long* drop_base, void* stream) {
  if (D != 256) return fail("layernorm: D must be 256 (got %d)", D);
  STCAT_LAUNCH(layernorm_fwd_kernel, dim3(grid_for(M, 4, 2048)), dim3(256), 0, (hipStream_t)stream, x, res, gamma,
               beta, y, mean, rstd, M, eps, stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base));
  return launch_status();
}

int stcat_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                        const float* rstd, float* dz, float* dx, float* dgamma, float* dbeta, int M, int D,
                        float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (D != 256) return fail("layernorm: D must be 256 (got %d)", D);
  if (drop_p > 0.f && !dx) return fail("layernorm_bwd: dropout needs the dx output");
  // small M (the decoders' [T,256] states): one row per wave, so the rows of a launch are normalised in parallel
  STCAT_LAUNCH(layernorm_bwd_kernel, dim3(grid_for(M, M <= 1024 ? 4 : 16, 512)), dim3(256), 0, (hipStream_t)stream, dy, x, res, gamma,
               mean, rstd, dz, dx, dgamma, dbeta, M, stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base));
  return launch_status();
}

int stcat_ew(int op, const float* a, const float* b, const float* c, float* out, long n, long bmod, float alpha,
             float beta, void* stream) {
  if (n <= 0) return fail("ew: n=%ld", n);
  if (bmod <= 0) bmod = n;
  STCAT_LAUNCH(ew_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, op, a, b, c, out, n, bmod,
               alpha, beta);
  return launch_status();
}

static int stg_loss_fill(StgLossParams& q, const float* boxes, const long* rows, const float* tgt, const float* sted,
                         const float* dist, const unsigned char* time_mask, const float* w, const unsigned char* pos_or_pad,
                         const float* nb_neg, const float* act, const float* act_tgt, const float* act_w,
                         const float* num_boxes_dev, float num_boxes, int nl, int rows_total, int nbox, int b, int T,
                         const float* wmat) {
  if (nl <= 0 || b <= 0 || T <= 0 || nbox < 0 || rows_total <= 0) return fail("stg_loss: nl=%d b=%d T=%d nbox=%d rows=%d", nl, b, T, nbox, rows_total);
  if (!boxes || !tgt || !sted || !dist || !time_mask || !w || !pos_or_pad || !nb_neg) return fail("stg_loss: NULL operand");
  if (nbox > 0 && !rows) return fail("stg_loss: rows is NULL");
  if (act && (!act_tgt || !act_w)) return fail("stg_loss: actioness logits without targets / weights");
  if (!num_boxes_dev && !(num_boxes > 0.f)) return fail("stg_loss: num_boxes=%f", (double)num_boxes);
  if (!aligned16(boxes) || !aligned16(tgt)) return fail("stg_loss: boxes / targets must be 16-byte aligned");
  q = StgLossParams{};
  q.boxes = boxes; q.rows = rows; q.tgt = tgt; q.sted = sted; q.dist = dist; q.time_mask = time_mask; q.w = w;
  q.pos_or_pad = pos_or_pad; q.nb_neg = nb_neg; q.act = act; q.act_tgt = act_tgt; q.act_w = act_w;
  q.num_boxes_dev = num_boxes_dev; q.num_boxes = num_boxes; q.nl = nl; q.rows_total = rows_total; q.nbox = nbox; q.b = b;
  q.T = T; q.wmat = wmat;
  return 0;
}

int stcat_stg_loss_fwd(const float* boxes, const long* rows, const float* tgt, const float* sted, const float* dist,
                       const unsigned char* time_mask, const float* w, const unsigned char* pos_or_pad,
                       const float* nb_neg, const float* act, const float* act_tgt, const float* act_w,
                       const float* num_boxes_dev, float num_boxes, int nl, int rows_total, int nbox, int b, int T,
                       const float* wmat, float* vec, float* total, void* stream) {
  StgLossParams q;
  if (int e = stg_loss_fill(q, boxes, rows, tgt, sted, dist, time_mask, w, pos_or_pad, nb_neg, act, act_tgt, act_w,
                            num_boxes_dev, num_boxes, nl, rows_total, nbox, b, T, wmat)) return e;
  if (!vec) return fail("stg_loss_fwd: vec is NULL");
  if (total && !wmat) return fail("stg_loss_fwd: total needs the weight matrix");
  q.vec = vec; q.total = total;
  STCAT_LAUNCH(stg_loss_fwd_kernel, dim3(nl), dim3(256), 0, (hipStream_t)stream, q);
  return launch_status();
}

int stcat_stg_loss_bwd(const float* boxes, const long* rows, const float* tgt, const float* sted, const float* dist,
                       const unsigned char* time_mask, const float* w, const unsigned char* pos_or_pad,
                       const float* nb_neg, const float* act, const float* act_tgt, const float* act_w,
                       const float* num_boxes_dev, float num_boxes, int nl, int rows_total, int nbox, int b, int T,
                       const float* wmat, const float* gvec, const float* gtotal, float* d_boxes, float* d_sted,
                       float* d_w, float* d_act, void* stream) {
  StgLossParams q;
  if (int e = stg_loss_fill(q, boxes, rows, tgt, sted, dist, time_mask, w, pos_or_pad, nb_neg, act, act_tgt, act_w,
                            num_boxes_dev, num_boxes, nl, rows_total, nbox, b, T, wmat)) return e;
  if (!gvec && !(gtotal && wmat)) return fail("stg_loss_bwd: no incoming gradient (gvec, or gtotal with the weight matrix)");
  if (!d_boxes || !d_sted || !d_w || (act && !d_act)) return fail("stg_loss_bwd: NULL gradient output");
  q.gvec = gvec; q.gtotal = gtotal; q.d_boxes = d_boxes; q.d_sted = d_sted; q.d_w = d_w; q.d_act = d_act;
  STCAT_LAUNCH(stg_loss_bwd_kernel, dim3(nl), dim3(256), 0, (hipStream_t)stream, q);
  return launch_status();
}

int stcat_ew2d(int op, const float* a, long lda, const float* b, long ldb, float* out, long ldo, long rows, int cols,
               float alpha, float beta, void* stream) {
  if (rows <= 0 || cols <= 0) return fail("ew2d: rows=%ld cols=%d", rows, cols);
  if (op != EW_ADD && op != EW_MUL && op != EW_AXPBY && op != EW_COPY) return fail("ew2d: op %d is not a two-operand op / copy", op);
  if (op != EW_COPY && !b) return fail("ew2d: op %d needs b", op);
  // lda / ldb == 0: one row broadcast to every output row
  if ((lda && lda < cols) || ldo < cols || (b && ldb && ldb < cols)) return fail("ew2d: leading dimension below cols");
  STCAT_LAUNCH(ew2d_kernel, dim3(grid_for(rows * cols, 256, 4096)), dim3(256), 0, (hipStream_t)stream, op, a, lda, b, ldb,
               out, ldo, rows, cols, alpha, beta);
  return launch_status();
}

int stcat_dropout(const float* x, const float* res, float* y, long n, float p, long seed, long offset,
                  const long* base, void* stream) {
  if (n <= 0) return fail("dropout: n=%ld", n);
  if (!(p >= 0.f && p < 1.f)) return fail("dropout: p=%f outside [0,1)", (double)p);
  if (!aligned16(x) || !aligned16(y) || (res && !aligned16(res))) return fail("dropout: pointers must be 16-byte aligned");
  STCAT_LAUNCH(dropout_kernel, dim3(grid_for((n + 3) / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, res, y, n,
               stcat_make_drop(p, seed, offset, base));
  return launch_status();
}

#define STCAT_NT_SWITCH(NT_, CALL)                                   \
  switch (NT_) {                                                     \
    case 1: { constexpr int NT = 1; CALL; } break;                   \
    case 2: { constexpr int NT = 2; CALL; } break;                   \
    case 3: { constexpr int NT = 3; CALL; } break;                   \
    case 4: { constexpr int NT = 4; CALL; } break;                   \
    case 5: { constexpr int NT = 5; CALL; } break;                   \
    case 6: { constexpr int NT = 6; CALL; } break;                   \
    case 7: { constexpr int NT = 7; CALL; } break;                   \
    case 8: { constexpr int NT = 8; CALL; } break;                   \
    default: return fail("mha_self: S=%d exceeds 256 tokens", S);    \
  }

int stcat_mha_self_fwd(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o,
                       float* pt, int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                       float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0) return fail("mha_self_fwd: bad shape");
  if ((ldq | ldk | ldv) % 4 != 0 || !aligned16(q) || !aligned16(k) || !aligned16(v))
    return fail("mha_self_fwd: q/k/v must be 16-byte aligned with ld %% 4 == 0");
  AttnParams p = {q, k, v, o, pt, kpm, B, H, S, ldq, ldk, ldv, ldo, scale,
                  stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base)};
  const int nt = cdiv(S, 32);
  if (nt > 8) {   // 256 < S <= 512 (non-square clips): K / V in dynamic LDS, 8 query tiles per workgroup
    if (nt > 16) return fail("mha_self_fwd: S=%d exceeds 512 tokens per frame", S);
    const int lds = nt * 32 * (33 + 32 + 1) * 4;
    if (int rc = pl_prepare(mha_self_fwd_long_kernel, 160 * 1024)) return rc;
    STCAT_LAUNCH(mha_self_fwd_long_kernel, dim3(B * H, cdiv(nt, 8)), dim3(512), lds, (hipStream_t)stream, p);
    return launch_status();
  }
  STCAT_NT_SWITCH(nt, STCAT_LAUNCH((mha_self_fwd_kernel<NT>), dim3(B * H), dim3(64 * NT), 0, (hipStream_t)stream, p))
  return launch_status();
}

// The recomputing pair (round 5): the forward keeps (row maximum, 1 / row sum) per query in lse [B][H][Sp][2] instead of the
// S x S probabilities; the backward rebuilds its probability tiles (csrc/attention.h).  S <= 256, no head-mean weights.
int stcat_mha_self_fwd_lse(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o, float* lse,
                           int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale, float drop_p,
                           long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0 || S > 256) return fail("mha_self_fwd_lse: bad shape (S = %d, at most 256)", S);
  if ((ldq | ldk | ldv) % 4 != 0 || !aligned16(q) || !aligned16(k) || !aligned16(v) || !lse)
    return fail("mha_self_fwd_lse: q/k/v must be 16-byte aligned with ld %% 4 == 0, lse set");
  AttnParams p = {q, k, v, o, nullptr, kpm, B, H, S, ldq, ldk, ldv, ldo, scale,
                  stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base), lse};
  const int nt = cdiv(S, 32);
  STCAT_NT_SWITCH(nt, STCAT_LAUNCH((mha_self_fwd_kernel<NT>), dim3(B * H), dim3(64 * NT), 0, (hipStream_t)stream, p))
  return launch_status();
}

int stcat_mha_self_bwd_lse(const float* q, const float* k, const float* v, const unsigned char* kpm, const float* out,
                           const float* dout, const float* lse, float* dq, float* dk, float* dv, int B, int H, int S,
                           int ldq, int ldk, int ldv, int ldo, int ldg, int ldgv, float scale, float drop_p,
                           long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0 || S > 256) return fail("mha_self_bwd_lse: bad shape (S = %d, at most 256)", S);
  if ((ldq | ldk | ldv | ldo) % 4 != 0 || !lse) return fail("mha_self_bwd_lse: ld %% 4 != 0 or lse missing");
  AttnBwdParams p = {};
  p.Q = q; p.K = k; p.V = v; p.dO = dout; p.O = out; p.Lse = lse; p.kpm = kpm;
  p.dQ = dq; p.dK = dk; p.dV = dv; p.B = B; p.H = H; p.S = S;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ldg = ldg; p.ldgv = ldgv; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  const int nt = cdiv(S, 32), sp = nt * 32;
  const int lds_q = (2 * sp * 33 + sp) * 4, lds_kv = (2 * sp * 33 + 3 * sp) * 4;
  STCAT_NT_SWITCH(nt, {
    if (int rc = pl_prepare(mha_self_bwd_dq_rc_kernel<NT>, 80 * 1024)) return rc;
    if (int rc = pl_prepare(mha_self_bwd_dkv_rc_kernel<NT>, 80 * 1024)) return rc;
    STCAT_LAUNCH((mha_self_bwd_dq_rc_kernel<NT>), dim3(B * H), dim3(64 * NT), lds_q, (hipStream_t)stream, p);
    STCAT_LAUNCH((mha_self_bwd_dkv_rc_kernel<NT>), dim3(B * H), dim3(64 * NT), lds_kv, (hipStream_t)stream, p);
  })
  return launch_status();
}

int stcat_mha_self_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                       const float* pt, const float* dw, float* corr, float* dst, float* dq, float* dk, float* dv,
                       int B, int H, int S, int ldq, int ldk, int ldv, int ldo, int ldg, int ldgv, float scale,
                       float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0) return fail("mha_self_bwd: bad shape");
  if ((ldq | ldk | ldv | ldo) % 4 != 0) return fail("mha_self_bwd: ld %% 4 != 0");
  if (dw && !corr) return fail("mha_self_bwd: dw given without corr scratch");
  AttnBwdParams p = {};
  p.Q = q; p.K = k; p.V = v; p.dO = dout; p.Pt = pt; p.dW = dw; p.O = out; p.corr = corr; p.dSt = dst;
  p.dQ = dq; p.dK = dk; p.dV = dv; p.B = B; p.H = H; p.S = S;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ldg = ldg; p.ldgv = ldgv; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  const int nt = cdiv(S, 32);
  if (dw) {
    STCAT_LAUNCH(attn_dw_corr_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, pt, dw, corr, B, H, S, nt * 32,
                 p.drop);
  }
  if (nt > 8) {
    if (nt > 16) return fail("mha_self_bwd: S=%d exceeds 512 tokens per frame", S);
    const int lds_q = nt * 32 * (33 + 32) * 4, lds_kv = nt * 32 * 64 * 4;
    if (int rc = pl_prepare(mha_self_bwd_dq_long_kernel, 160 * 1024)) return rc;
    if (int rc = pl_prepare(mha_self_bwd_dkv_long_kernel, 160 * 1024)) return rc;
    STCAT_LAUNCH(mha_self_bwd_dq_long_kernel, dim3(B * H, cdiv(nt, 8)), dim3(512), lds_q, (hipStream_t)stream, p);
    if (int rc = launch_status()) return rc;
    STCAT_LAUNCH(mha_self_bwd_dkv_long_kernel, dim3(B * H, cdiv(nt, 8)), dim3(512), lds_kv, (hipStream_t)stream, p);
    return launch_status();
  }
  STCAT_NT_SWITCH(nt, STCAT_LAUNCH((mha_self_bwd_dq_kernel<NT>), dim3(B * H), dim3(64 * NT), 0, (hipStream_t)stream, p))
  int rc = launch_status();
  if (rc) return rc;
  STCAT_NT_SWITCH(nt, STCAT_LAUNCH((mha_self_bwd_dkv_kernel<NT>), dim3(B * H), dim3(64 * NT), 0, (hipStream_t)stream, p))
  return launch_status();
}

int stcat_attn_weights_mean(const float* pt, float* w, int B, int H, int S, float drop_p, long drop_seed,
                            long drop_offset, const long* drop_base, void* stream) {
  const int SP = cdiv(S, 32) * 32;
  STCAT_LAUNCH(attn_weights_mean_kernel, dim3(grid_for((long)B * S * S, 256)), dim3(256), 0, (hipStream_t)stream, pt, w,
               B, H, S, SP, stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base));
  return launch_status();
}

int stcat_attn_q1_fwd(const float* q1, const float* q2, const float* k1, const float* k2, const float* v,
                      const unsigned char* kpm, float* out, float* P, int B, int H, int S, int ldq, int ldk,
                      int ldv, float scale, float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || S > 256 * STCAT_Q1_MAXC) return fail("attn_q1: S=%d out of range (1..%d)", S, 256 * STCAT_Q1_MAXC);
  if ((ldq | ldk | ldv) % 4 != 0) return fail("attn_q1: ld %% 4 != 0");
  AttnQ1Params p = {};
  p.q1 = q1; p.q2 = q2; p.k1 = k1; p.k2 = k2; p.v = v; p.kpm = kpm; p.out = out; p.P = P;
  p.B = B; p.H = H; p.S = S; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  switch ((S + 255) / 256) {   // 256-key chunks per (frame, head)
    case 1: STCAT_LAUNCH(attn_q1_fwd_kernel<1>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
    case 2: STCAT_LAUNCH(attn_q1_fwd_kernel<2>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
    default: STCAT_LAUNCH(attn_q1_fwd_kernel<STCAT_Q1_MAXC>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
  }
  return launch_status();
}

int stcat_attn_q1_bwd(const float* q1, const float* q2, const float* k1, const float* k2, const float* v,
                      const float* P, const float* dout, float* dq1, float* dq2, float* dk1, float* dk2, float* dv,
                      int B, int H, int S, int ldq, int ldk, int ldv, float scale, float drop_p, long drop_seed,
                      long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || S > 256 * STCAT_Q1_MAXC) return fail("attn_q1: S=%d out of range (1..%d)", S, 256 * STCAT_Q1_MAXC);
  AttnQ1Params p = {};
  p.q1 = q1; p.q2 = q2; p.k1 = k1; p.k2 = k2; p.v = v; p.P = const_cast<float*>(P); p.dout = dout;
  p.dq1 = dq1; p.dq2 = dq2; p.dk1 = dk1; p.dk2 = dk2; p.dv = dv;
  p.B = B; p.H = H; p.S = S; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  switch ((S + 255) / 256) {
    case 1: STCAT_LAUNCH(attn_q1_bwd_kernel<1>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
    case 2: STCAT_LAUNCH(attn_q1_bwd_kernel<2>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
    default: STCAT_LAUNCH(attn_q1_bwd_kernel<STCAT_Q1_MAXC>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, p); break;
  }
  return launch_status();
}

// ---- optimizer tail --------------------------------------------------------------------------------------
int stcat_optim_table_entry_bytes(void) { return (int)sizeof(OptTensor); }

int stcat_grad_sqnorm(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                      float* out_sq, void* stream) {
  if (n_chunks <= 0 || chunk <= 0 || chunk % 4 != 0) return fail("grad_sqnorm: bad chunking (%d x %d)", n_chunks, chunk);
  hipError_t e = hipMemsetAsync(out_sq, 0, sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return fail("grad_sqnorm: memset: %s", hipGetErrorString(e));
  STCAT_LAUNCH(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table,
               chunk_tensor, chunk_off, chunk, out_sq);
  return launch_status();
}

int stcat_adamw_ema_step(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                         const float* sqnorm, const float* lr, const float* wd, int n_groups, float beta1,
                         float beta2, float eps, int step, float max_norm, float ema_decay, void* stream) {
  if (n_chunks <= 0 || chunk <= 0 || chunk % 4 != 0) return fail("adamw: bad chunking (%d x %d)", n_chunks, chunk);
  if (n_groups <= 0 || n_groups > STCAT_OPT_MAX_GROUPS) return fail("adamw: %d parameter groups (max %d)", n_groups,
                                                                     STCAT_OPT_MAX_GROUPS);
  if (step < 1) return fail("adamw: step counts from 1");
  if (max_norm > 0.f && !sqnorm) return fail("adamw: clipping needs the squared gradient norm");
  OptHyper h = {};
  for (int i = 0; i < n_groups; ++i) { h.lr[i] = lr[i]; h.wd[i] = wd[i]; }  // lr / wd are HOST arrays
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
  h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  h.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  h.max_norm = max_norm; h.ema_decay = ema_decay;
  h.skip_nonfinite = g_pl_f16 ? 1 : 0;
  STCAT_LAUNCH(adamw_ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table,
               chunk_tensor, chunk_off, chunk, sqnorm, h);
  return launch_status();
}

int stcat_grad_clip_scale(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                          const float* sqnorm, float max_norm, void* stream) {
  if (n_chunks <= 0 || chunk <= 0) return fail("grad_clip_scale: bad chunking");
  if (!sqnorm || !(max_norm > 0.f)) return fail("grad_clip_scale: needs the squared norm and max_norm > 0");
  STCAT_LAUNCH(grad_clip_scale_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table,
               chunk_tensor, chunk_off, chunk, sqnorm, max_norm);
  return launch_status();
}

int stcat_ema_update(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                     float decay, void* stream) {
  if (n_chunks <= 0 || chunk <= 0) return fail("ema_update: bad chunking");
  STCAT_LAUNCH(ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table, chunk_tensor,
               chunk_off, chunk, decay);
  return launch_status();
}

int stcat_temporal_map_argmax(const float* sted, const int* durations, int* out, int b, int T, void* stream) {
  if (T <= 0 || T > 1024) return fail("temporal_map_argmax: T=%d out of range (1..1024)", T);
  STCAT_LAUNCH(temporal_map_argmax_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, sted, durations, out, T);
  return launch_status();
}

// ---- plane-format backbone (mma mode 4): every tensor is a pair of bf16 planes (hi, lo) ------------------------
int stcat_debug_force_pl_tile(int index) {
  if (index < -1 || index > 7) return fail("debug_force_pl_tile: index must be -1 .. 7 (7 = the A-stationary kernel where it applies)");
  g_pl_force = index;
  return 0;
}

int stcat_debug_pl_flags(int flags) {
  g_stem_pl = (flags & 0x8000) ? 0 : 1;   // bit 0x8000: the six-product modes back on the exact-fp32 stem (A/B of stem_pl.h)
  g_pl_as_skew = (flags & 0x2000) ? ((flags >> 16) & 0xffff) : 0;
  if (flags & 0x2000) flags &= 0x1fff;
  g_pl_debug = flags & ~(4 | 128);     // bits 0,1,3.. : timing experiments (PlParams::debug, stagger)
  g_pl3_small = (flags & 4) ? 1 : 0;   // bit 2: three-plane short reductions on the two-workgroup 128 x 64 tile
  g_pl_as = (flags & 128) ? 0 : 1;     // bit 7: the K <= 256 1x1 layers back on the two-stage tile kernel (A/B of igemm_pl_as.h)
  return 0;
}

int stcat_pl_conv_fwd(const void* xh, const void* xl, const void* wh, const void* wl, const float* scale,
                      const float* bias, const void* rh, const void* rl, void* yh, void* yl, float* yf,
                      unsigned char* ymask, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                      int relu, void* stream) {
  if (Cin % 32 != 0 || Cout % 64 != 0) return fail("pl_conv_fwd: need Cin %% 32 == 0 and Cout %% 64 == 0 (%d, %d)", Cin, Cout);
  if (!aligned16(xh) || !aligned16(xl) || !aligned16(wh) || !aligned16(wl)) return fail("pl_conv_fwd: planes must be 16-byte aligned");
  if (!yh && !yf) return fail("pl_conv_fwd: no output");
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  PlParams p = {};
  p.Ah = (const __bf16*)xh; p.Al = (const __bf16*)xl; p.Bh = (const __bf16*)wh; p.Bl = (const __bf16*)wl;
  p.Ch = (__bf16*)yh; p.Cl = (__bf16*)yl; p.Cf = yf; p.Mo = ymask; p.scale = scale; p.bias = bias;
  p.Rh = (const __bf16*)rh; p.Rl = (const __bf16*)rl;
  p.a_bytes = plane_bytes((long)n * H * W * Cin); p.b_bytes = plane_bytes((long)Cout * KH * KW * Cin);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_conv_fwd: a plane exceeds 2 GB");
  p.b_tap_stride = (unsigned)Cin;
  p.M = n * OH * OW; p.N = Cout; p.K = KH * KW * Cin; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout; p.relu = relu;
  p.g = conv_geom_fwd(H, W, Cin, Cin, OH, OW, KH, KW, stride, pad);
  return launch_pl_fwd(p, (hipStream_t)stream);
}

int stcat_pl_conv_dgrad(const void* gh, const void* gl, const void* th, const void* tl, const void* addh,
                        const void* addl, const void* yh, const void* yl, const unsigned char* ybits,
                        const float* mask_scale, void* dxh, void* dxl,
                        void* dx2h, void* dx2l, const float* dx2_scale, int n, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, void* stream) {
  if ((dx2h != nullptr) != (dx2_scale != nullptr)) return fail("pl_conv_dgrad: dx2 and dx2_scale go together");
  if (Cout % 32 != 0 || Cin % 64 != 0) return fail("pl_conv_dgrad: need Cout %% 32 == 0 and Cin %% 64 == 0 (%d, %d)", Cout, Cin);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  PlParams p = {};
  p.Ah = (const __bf16*)gh; p.Al = (const __bf16*)gl; p.Bh = (const __bf16*)th; p.Bl = (const __bf16*)tl;
  p.Ch = (__bf16*)dxh; p.Cl = (__bf16*)dxl; p.Rh = (const __bf16*)addh; p.Rl = (const __bf16*)addl;
  p.Yh = (const __bf16*)yh; p.Yl = (const __bf16*)yl; p.Mi = ybits; p.mscale = mask_scale;
  p.C2h = (__bf16*)dx2h; p.C2l = (__bf16*)dx2l; p.c2scale = dx2_scale;
  p.a_bytes = plane_bytes((long)n * OH * OW * Cout); p.b_bytes = plane_bytes((long)Cout * KH * KW * Cin);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_conv_dgrad: a plane exceeds 2 GB");
  // transposed weight planes [tap][Cin][Cout]: row n = ci (stride Cout), tap stride Cin * Cout
  p.ldb = Cout; p.b_tap_stride = (unsigned)((long)Cin * Cout);
  p.M = n * H * W; p.N = Cin; p.K = KH * KW * Cout; p.ldc = Cin; p.ldr = Cin; p.relu = 0;
  IgemmGeom q;
  q.H = OH; q.W = OW; q.C = Cout; q.ld = Cout; q.OH = H; q.OW = W; q.KH = KH; q.KW = KW;
  q.mul = 1; q.off = pad; q.sgn = -1; q.div = stride;
  stcat_fastdiv_magic(W, &q.mg_ow, &q.sh_ow);
  stcat_fastdiv_magic(H * W, &q.mg_ohw, &q.sh_ohw);
  p.g = q;
  return launch_pl_fwd(p, (hipStream_t)stream);
}

// The block-boundary data gradient of a bottleneck WITH a strided downsample branch (torchvision Bottleneck: out =
// conv3(..) + downsample(x), models/vision_model/backbone.py:115-119): dx = [ybits] (g . W1^T + scatter(addc)) where addc
// [n, ceil(H / add_stride), ceil(W / add_stride), Cin] is the downsample conv's data gradient on ITS OWN (coarse) grid —
// a plain 1x1 GEMM — and scatter() places row (i, j) at pixel (add_stride i, add_stride j).  Round 4 materialised the
// scattered tensor (3/4 zeros: 0.92 ms per launch at layer3.0 / layer4.0) and read it back as `add`.
int stcat_pl_conv_dgrad_cadd(const void* gh, const void* gl, const void* th, const void* tl, const void* addch,
                             const void* addcl, int add_stride, const unsigned char* ybits, const float* mask_scale,
                             void* dxh, void* dxl, int n, int H, int W, int Cin, int Cout, void* stream) {
  if (Cout % 32 != 0 || Cin % 64 != 0) return fail("pl_conv_dgrad_cadd: need Cout %% 32 == 0 and Cin %% 64 == 0 (%d, %d)", Cout, Cin);
  if (add_stride != 2 && add_stride != 4) return fail("pl_conv_dgrad_cadd: add_stride must be 2 or 4");
  if (!addch || !addcl) return fail("pl_conv_dgrad_cadd: the coarse-grid operand is required");
  if (g_mma_mode_raw < 4) return fail("pl_conv_dgrad_cadd: plane modes only");
  PlParams p = {};
  p.Ah = (const __bf16*)gh; p.Al = (const __bf16*)gl; p.Bh = (const __bf16*)th; p.Bl = (const __bf16*)tl;
  p.Ch = (__bf16*)dxh; p.Cl = (__bf16*)dxl; p.Rh = (const __bf16*)addch; p.Rl = (const __bf16*)addcl;
  p.Mi = ybits; p.mscale = mask_scale;
  p.radd_div = add_stride; p.radd_h = (H - 1) / add_stride + 1; p.radd_w = (W - 1) / add_stride + 1;
  p.a_bytes = plane_bytes((long)n * H * W * Cout); p.b_bytes = plane_bytes((long)Cout * Cin);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_conv_dgrad_cadd: a plane exceeds 2 GB");
  p.ldb = Cout; p.b_tap_stride = (unsigned)((long)Cin * Cout);
  p.M = n * H * W; p.N = Cin; p.K = Cout; p.ldc = Cin; p.ldr = Cin; p.relu = 0;
  IgemmGeom q;
  q.H = H; q.W = W; q.C = Cout; q.ld = Cout; q.OH = H; q.OW = W; q.KH = 1; q.KW = 1;
  q.mul = 1; q.off = 0; q.sgn = -1; q.div = 1;
  stcat_fastdiv_magic(W, &q.mg_ow, &q.sh_ow);
  stcat_fastdiv_magic(H * W, &q.mg_ohw, &q.sh_ohw);
  p.g = q;
  return launch_pl_fwd(p, (hipStream_t)stream);
}

// ---- Linear layers on the plane kernels (round 5): the encoder FFN's wide side (modal_encoder.py:239-240, K = 256 ->
// N = 2048 at 13 248 rows) is exactly the A-stationary kernel's shape; operands arrive as planes (stcat_pl_split of the
// LayerNorm output / the upstream gradient: 10 us), results leave as fp32 for the consumers that stay on fp32 tensors.
// y = dropout_p(relu?(x w^T + bias)) [M, N] fp32 (+ the bit mask y > 0 for the backward pass)
int stcat_pl_linear_fwd(const void* xh, const void* xl, const void* wh, const void* wl, const float* bias, const float* addf,
                        float* yf, void* yh, void* yl, unsigned char* ymask, int M, int N, int K, int relu, float drop_p,
                        long drop_seed, long drop_offset, const long* drop_base, void* stream) {
  if (K % 32 != 0 || N % 64 != 0 || M <= 0) return fail("pl_linear_fwd: need K %% 32 == 0, N %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
  if (!aligned16(xh) || !aligned16(xl) || !aligned16(wh) || !aligned16(wl)) return fail("pl_linear_fwd: planes must be 16-byte aligned");
  if (!yf && !yh) return fail("pl_linear_fwd: no output (yf and yh both null)");
  if ((yh != nullptr) != (yl != nullptr) || !aligned16(yh) || !aligned16(yl) || !aligned16(yf) || !aligned16(addf))
    return fail("pl_linear_fwd: yh / yl go together; outputs and addf 16-byte aligned");
  if (g_mma_mode_raw < 4) return fail("pl_linear_fwd: plane modes only");
  PlParams p = {};
  p.Ah = (const __bf16*)xh; p.Al = (const __bf16*)xl; p.Bh = (const __bf16*)wh; p.Bl = (const __bf16*)wl;
  p.Cf = yf; p.Ch = (__bf16*)yh; p.Cl = (__bf16*)yl; p.Mo = ymask; p.bias = bias; p.relu = relu; p.Rf = addf;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  p.a_bytes = plane_bytes((long)M * K); p.b_bytes = plane_bytes((long)N * K);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_linear_fwd: a plane exceeds 2 GB");
  p.b_tap_stride = (unsigned)K;
  p.M = M; p.N = N; p.K = K; p.ldb = K; p.ldc = N; p.ldr = N;
  p.g = conv_geom_fwd(1, M, K, K, 1, M, 1, 1, 1, 0);
  return launch_pl_fwd(p, (hipStream_t)stream);
}
// dx [M, K] fp32 = [ybits] mask_scale[k] * (g [M, N] . w [N, K]); th / tl = the TRANSPOSED weight planes [K][N]
int stcat_pl_linear_dgrad_mask(const void* gh, const void* gl, const void* th, const void* tl, const unsigned char* ybits,
                               const float* mask_scale, float* dxf, void* dxh, void* dxl, int M, int N, int K, void* stream) {
  if (N % 32 != 0 || K % 64 != 0 || M <= 0) return fail("pl_linear_dgrad_mask: need N %% 32 == 0, K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
  if (!aligned16(gh) || !aligned16(gl) || !aligned16(th) || !aligned16(tl)) return fail("pl_linear_dgrad_mask: planes must be 16-byte aligned");
  if ((!dxf && !dxh) || (dxh != nullptr) != (dxl != nullptr) || !aligned16(dxh) || !aligned16(dxl) || !aligned16(dxf))
    return fail("pl_linear_dgrad_mask: need dxf and / or the plane pair dxh + dxl, 16-byte aligned");
  if (g_mma_mode_raw < 4) return fail("pl_linear_dgrad_mask: plane modes only");
  PlParams p = {};
  p.Ah = (const __bf16*)gh; p.Al = (const __bf16*)gl; p.Bh = (const __bf16*)th; p.Bl = (const __bf16*)tl;
  p.Cf = dxf; p.Ch = (__bf16*)dxh; p.Cl = (__bf16*)dxl; p.Mi = ybits; p.mscale = mask_scale;
  p.a_bytes = plane_bytes((long)M * N); p.b_bytes = plane_bytes((long)N * K);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_linear_dgrad_mask: a plane exceeds 2 GB");
  p.ldb = N; p.b_tap_stride = (unsigned)((long)K * N);
  p.M = M; p.N = K; p.K = N; p.ldc = K; p.ldr = K; p.relu = 0;
  IgemmGeom q;
  q.H = 1; q.W = M; q.C = N; q.ld = N; q.OH = 1; q.OW = M; q.KH = 1; q.KW = 1;
  q.mul = 1; q.off = 0; q.sgn = -1; q.div = 1;
  stcat_fastdiv_magic(M, &q.mg_ow, &q.sh_ow);
  stcat_fastdiv_magic(M, &q.mg_ohw, &q.sh_ohw);
  p.g = q;
  return launch_pl_fwd(p, (hipStream_t)stream);
}

int stcat_pl_conv_wgrad(const void* gh, const void* gl, const void* xh, const void* xl, float* dw, const float* row_scale,
                        int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  return stcat_pl_conv_wgrad_ws(gh, gl, xh, xl, dw, row_scale, n, H, W, Cin, Cout, KH, KW, stride, pad, nullptr, 0, stream);
}

// ... with a workspace of ws_floats fp32 values: when the launch's slices x Cout x KH KW Cin partial tiles fit, no atomics are
// used and dw is bit-identical run to run; else (or ws null) the atomic form above
int stcat_pl_conv_wgrad_ws(const void* gh, const void* gl, const void* xh, const void* xl, float* dw, const float* row_scale,
                           int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws,
                           long ws_floats, void* stream) {
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  PlParams p = {};
  p.Ah = (const __bf16*)gh; p.Al = (const __bf16*)gl; p.Bh = (const __bf16*)xh; p.Bl = (const __bf16*)xl;
  p.Wf = dw; p.wscale = row_scale; p.ldb = Cout; p.ldc = KH * KW * Cin;
  p.a_bytes = plane_bytes((long)n * OH * OW * Cout); p.b_bytes = plane_bytes((long)n * H * W * Cin);
  if (p.a_bytes == 0xFFFFFFFFu || p.b_bytes == 0xFFFFFFFFu) return fail("pl_conv_wgrad: a plane exceeds 2 GB");
  p.g = conv_geom_fwd(H, W, Cin, Cin, OH, OW, KH, KW, stride, pad);
  return launch_pl_wgrad(p, Cout, KH * KW * Cin, n * OH * OW, (hipStream_t)stream, ws, ws_floats);
}

// out[n] (caller-zeroed) += sum over rows of the plane set [M][N]
int stcat_pl_colsum(const void* h, const void* l, float* out, int M, int N, void* stream) {
  if (M <= 0 || N <= 0 || N % 8 != 0 || !aligned16(h) || !aligned16(l)) return fail("pl_colsum: M=%d N=%d (N %% 8 == 0, planes 16-byte aligned)", M, N);
  if (g_mma_mode_raw < 4) return fail("pl_colsum: plane modes only");
  int rows = cdiv(M, 256);
  if (rows < 32) rows = 32;
  STCAT_LAUNCH(pl_colsum_kernel, dim3(cdiv(M, rows), cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)h,
               (const __bf16*)l, out, M, N, pl_np_arg(), rows);
  return launch_status();
}

int stcat_pl_maxpool3x3s2(const float* x, void* yh, void* yl, int n, int H, int W, int C, void* stream) {
  if (C % 8 != 0 || !aligned16(x) || !aligned16(yh) || !aligned16(yl)) return fail("pl_maxpool: C %% 8 != 0 or unaligned");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)n * OH * OW * (C / 8);
  STCAT_LAUNCH(maxpool3x3s2_pl_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x,
               (__bf16*)yh, (__bf16*)yl, n, H, W, C, OH, OW, pl_np_arg());
  return launch_status();
}

static int pl_ew(PlEwParams p, long n, void* stream) {
  if (n <= 0 || n % 8 != 0) return fail("plane element-wise: n = %ld must be a positive multiple of 8", n);
  p.n8 = n / 8;
  p.np = pl_np_arg();
  p.gmul = (g_pl_f16 && p.mode == 1) ? ldexpf(1.f, g_f16_glog) : 1.f;   // gradients enter the fp16 plane domain scaled
  STCAT_LAUNCH(planes_ew_kernel, dim3(grid_for(p.n8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

int stcat_pl_split(const float* x, void* h, void* l, long n, void* stream) {
  PlEwParams p = {};
  p.mode = 0; p.xf = x; p.Rh = (__bf16*)h; p.Rl = (__bf16*)l; p.C = 8;
  return pl_ew(p, n, stream);
}

// planes of x + y, and (optional) the fp32 sum itself: the encoder's q = k = src + pos feeding the in-projection on the
// plane kernels without a separate add pass (round 5)
int stcat_pl_split_sum(const float* x, const float* y, float* sum, void* h, void* l, long n, void* stream) {
  if (!x || !y || !h || !l) return fail("pl_split_sum: x, y and the plane pair are required");
  PlEwParams p = {};
  p.mode = 4; p.xf = x; p.yf = y; p.of = sum; p.Rh = (__bf16*)h; p.Rl = (__bf16*)l; p.C = 8;
  return pl_ew(p, n, stream);
}

int stcat_pl_join(const void* h, const void* l, float* out, long n, void* stream) {
  PlEwParams p = {};
  p.mode = 3; p.Xh = (const __bf16*)h; p.Xl = (const __bf16*)l; p.of = out; p.C = 8;
  return pl_ew(p, n, stream);
}

int stcat_pl_act_bwd(const float* dy, const float* y, const float* scale, void* gh, void* gl, void* rh, void* rl, long n,
                     int C, int relu, void* stream) {
  if (C % 8 != 0) return fail("pl_act_bwd: C %% 8 != 0");
  PlEwParams p = {};
  p.mode = 1; p.xf = dy; p.yf = y; p.scale = scale; p.Gh = (__bf16*)gh; p.Gl = (__bf16*)gl; p.Rh = (__bf16*)rh;
  p.Rl = (__bf16*)rl; p.C = C; p.relu = relu;
  return pl_ew(p, n, stream);
}

int stcat_pl_scale(const void* xh, const void* xl, const float* scale, void* gh, void* gl, long n, int C, void* stream) {
  if (C % 8 != 0) return fail("pl_scale: C %% 8 != 0");
  PlEwParams p = {};
  p.mode = 2; p.Xh = (const __bf16*)xh; p.Xl = (const __bf16*)xl; p.scale = scale; p.Gh = (__bf16*)gh; p.Gl = (__bf16*)gl;
  p.C = C;
  return pl_ew(p, n, stream);
}

int stcat_weight_planes_entry_bytes(void) { return (int)sizeof(WplEntry); }

int stcat_weight_planes_multi(const void* table, int n_entries, int total_blocks, void* stream) {
  if (n_entries <= 0 || total_blocks <= 0) return fail("weight_planes_multi: empty table");
  STCAT_LAUNCH(weight_planes_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const WplEntry*)table,
               n_entries, pl_np_arg(), g_pl_f16 ? ldexpf(1.f, g_f16_wlog) : 1.f);
  return launch_status();
}

// ---- self-attention on the bf16 pipe (attention_bs.h) ---------------------------------------------------------
#define STCAT_NW_SWITCH(NW_, CALL)                                   \
  switch (NW_) {                                                     \
    case 1: { constexpr int NW = 1; CALL; } break;                   \
    case 2: { constexpr int NW = 2; CALL; } break;                   \
    case 3: { constexpr int NW = 3; CALL; } break;                   \
    case 4: { constexpr int NW = 4; CALL; } break;                   \
    case 5: { constexpr int NW = 5; CALL; } break;                   \
    case 6: { constexpr int NW = 6; CALL; } break;                   \
    case 7: { constexpr int NW = 7; CALL; } break;                   \
    default: { constexpr int NW = 8; CALL; } break;                  \
  }

// planes per operand of the bf16-pipe self-attention: 3 (six products, fp32-class) in mode bf16x6p, else 2 (three products)
static int mha_bs_planes() { return g_mma_mode_raw == 5 ? 3 : 2; }

int stcat_mha_bs_fwd(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o, float* lse,
                     int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale, float drop_p, long drop_seed,
                     long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0) return fail("mha_bs_fwd: bad shape");
  if ((ldq | ldk | ldv | ldo) % 4 != 0 || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o))
    return fail("mha_bs_fwd: q/k/v/o must be 16-byte aligned with ld %% 4 == 0");
  AttnBsParams p = {};
  p.Q = q; p.K = k; p.V = v; p.O = o; p.lse = lse; p.kpm = kpm; p.B = B; p.H = H; p.S = S;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  const int nq = cdiv(S, 32), nw = nq < 8 ? nq : 8;
  hipStream_t st = (hipStream_t)stream;
  if (mha_bs_planes() == 3) {      // mode bf16x6p: three planes per operand, six products (round 6)
    const int lds = 6 * 256 * 64 + 256 * 4;
    STCAT_NW_SWITCH(nw, {
      if (int rc = pl_prepare(mha_bs_fwd_kernel<NW, 3>, lds)) return rc;
      STCAT_LAUNCH((mha_bs_fwd_kernel<NW, 3>), dim3(B * H, cdiv(nq, nw)), dim3(64 * NW), lds, st, p);
    })
    return launch_status();
  }
  const int lds = 4 * 256 * 64 + 256 * 4;
  STCAT_NW_SWITCH(nw, {
    if (int rc = pl_prepare(mha_bs_fwd_kernel<NW, 2>, lds)) return rc;
    STCAT_LAUNCH((mha_bs_fwd_kernel<NW, 2>), dim3(B * H, cdiv(nq, nw)), dim3(64 * NW), lds, st, p);
  })
  return launch_status();
}

int stcat_mha_bs_bwd(const float* q, const float* k, const float* v, const unsigned char* kpm, const float* out,
                     const float* dout, const float* lse, float* dq, float* dk, float* dv, int B, int H, int S, int ldq,
                     int ldk, int ldv, int ldo, int ldg, int ldgv, float scale, float drop_p, long drop_seed,
                     long drop_offset, const long* drop_base, void* stream) {
  if (S <= 0 || B <= 0 || H <= 0) return fail("mha_bs_bwd: bad shape");
  if (S > 256) return fail("mha_bs_bwd: S=%d exceeds 256 tokens (training on longer token rows is not built yet)", S);
  if ((ldq | ldk | ldv | ldo | ldg | ldgv) % 4 != 0) return fail("mha_bs_bwd: ld %% 4 != 0");
  AttnBsParams p = {};
  p.Q = q; p.K = k; p.V = v; p.lse = const_cast<float*>(lse); p.kpm = kpm; p.dO = dout; p.dQ = dq; p.dK = dk; p.dV = dv;
  p.B = B; p.H = H; p.S = S; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ldg = ldg; p.ldgv = ldgv; p.scale = scale;
  p.drop = stcat_make_drop(drop_p, drop_seed, drop_offset, drop_base);
  const int nw = cdiv(S, 32);
  hipStream_t st = (hipStream_t)stream;
  if (mha_bs_planes() == 3) {
    // three planes of Q, K, V, dO do not fit one workgroup's LDS (172 KB at S = 224): dQ and dK / dV as two launches, each
    // with two operands as planes in LDS (86 KB) and its own tile's rows in registers
    const int lds = 6 * nw * 32 * 64 + 2 * nw * 32 * 4;
    STCAT_NW_SWITCH(nw, {
      if (int rc = pl_prepare(mha_bs_bwd_dq_kernel<NW, 3>, lds)) return rc;
      if (int rc = pl_prepare(mha_bs_bwd_dkv_kernel<NW, 3>, lds)) return rc;
      STCAT_LAUNCH((mha_bs_bwd_dq_kernel<NW, 3>), dim3(B * H), dim3(64 * NW), lds, st, p, out);
      STCAT_LAUNCH((mha_bs_bwd_dkv_kernel<NW, 3>), dim3(B * H), dim3(64 * NW), lds, st, p, out);
    })
    return launch_status();
  }
  const int lds = 8 * nw * 32 * 64 + 3 * nw * 32 * 4;
  STCAT_NW_SWITCH(nw, {
    if (int rc = pl_prepare(mha_bs_bwd_kernel<NW>, lds)) return rc;
    STCAT_LAUNCH((mha_bs_bwd_kernel<NW>), dim3(B * H), dim3(64 * NW), lds, st, p, out);
  })
  return launch_status();
}

// ---- 2D temporal map head (models/map2d_head.py), optional op -------------------------------------
int stcat_map2d_pool(const float* x, float* pooled, int b, int T, int N, int D, void* stream) {
  if (b <= 0 || T <= 0 || N <= 0 || D <= 0) return fail("map2d_pool: bad shape");
  STCAT_LAUNCH(map2d_pool_kernel, dim3(grid_for((long)b * N * D, 256)), dim3(256), 0, (hipStream_t)stream, x, pooled, b, T,
               N, D);
  return launch_status();
}

int stcat_map2d_cells(const float* pooled, const int* cell_i, const int* cell_j, int ncells, float* map, int b, int N, int D,
                      void* stream) {
  if (D % 4 != 0 || ncells <= 0) return fail("map2d_cells: D %% 4 != 0 or no cells");
  STCAT_LAUNCH(map2d_cells_kernel, dim3(grid_for((long)b * ncells * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream,
               pooled, cell_i, cell_j, ncells, map, b, N, D);
  return launch_status();
}

int stcat_map2d_cells_bwd(const float* pooled, const int* cell_i, const int* cell_j, int ncells, const float* dmap,
                          float* dpooled, int b, int N, int D, void* stream) {
  if (ncells <= 0 || b <= 0 || N <= 0 || D <= 0) return fail("map2d_cells_bwd: bad shape");
  STCAT_LAUNCH(map2d_cells_bwd_kernel, dim3(grid_for((long)b * ncells * D, 256)), dim3(256), 0, (hipStream_t)stream, pooled,
               cell_i, cell_j, ncells, dmap, dpooled, b, N, D);
  return launch_status();
}

int stcat_map2d_pool_bwd(const float* x, const float* dpooled, float* dx, int b, int T, int N, int D, void* stream) {
  if (b <= 0 || T <= 0 || N <= 0 || D <= 0) return fail("map2d_pool_bwd: bad shape");
  STCAT_LAUNCH(map2d_pool_bwd_kernel, dim3(grid_for((long)b * N * D, 256)), dim3(256), 0, (hipStream_t)stream, x, dpooled, dx,
               b, T, N, D);
  return launch_status();
}

int stcat_rowscale(float* y, const float* w, long rows, int C, int period, void* stream) {
  if (C % 4 != 0 || rows <= 0 || period <= 0) return fail("rowscale: bad shape");
  STCAT_LAUNCH(rowscale_kernel, dim3(grid_for(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, y, w, rows, C, period);
  return launch_status();
}

}  // extern "C"

// ---- launch plans (launch_plan.h): record once, replay with one host call -----------------------------------------
#include "launch_plan.h"

namespace {
#define STCAT_PLAN_FN(name) \
  { #name, &stcat_plan::Thunk<decltype(&name), &name>::call, stcat_plan::Thunk<decltype(&name), &name>::nargs }
const stcat_plan::FnEntry g_plan_fns[] = {
    STCAT_PLAN_FN(stcat_frozen_bn_fold),
    STCAT_PLAN_FN(stcat_stem_fwd),
    STCAT_PLAN_FN(stcat_stem_u8_fwd),
    STCAT_PLAN_FN(stcat_maxpool3x3s2),
    STCAT_PLAN_FN(stcat_conv_fwd),
    STCAT_PLAN_FN(stcat_conv_dgrad),
    STCAT_PLAN_FN(stcat_weight_transpose),
    STCAT_PLAN_FN(stcat_weight_transpose_multi),
    STCAT_PLAN_FN(stcat_conv_wgrad),
    STCAT_PLAN_FN(stcat_act_bwd),
    STCAT_PLAN_FN(stcat_pos_sine_2d),
    STCAT_PLAN_FN(stcat_sine_embed_fwd),
    STCAT_PLAN_FN(stcat_sine_embed_bwd),
    STCAT_PLAN_FN(stcat_linear_fwd),
    STCAT_PLAN_FN(stcat_linear_dgrad),
    STCAT_PLAN_FN(stcat_linear_fwd_acc),
    STCAT_PLAN_FN(stcat_linear_dgrad_acc),
    STCAT_PLAN_FN(stcat_linear_fwd_drop),
    STCAT_PLAN_FN(stcat_linear_fwd_multi),
    STCAT_PLAN_FN(stcat_linear_dgrad_multi),
    STCAT_PLAN_FN(stcat_linear_wgrad_multi),
    STCAT_PLAN_FN(stcat_linear_dgrad_mask),
    STCAT_PLAN_FN(stcat_linear_wgrad),
    STCAT_PLAN_FN(stcat_small_linear_fwd),
    STCAT_PLAN_FN(stcat_small_linear_bwd),
    STCAT_PLAN_FN(stcat_colsum),
    STCAT_PLAN_FN(stcat_layernorm_fwd),
    STCAT_PLAN_FN(stcat_layernorm_bwd),
    STCAT_PLAN_FN(stcat_ew),
    STCAT_PLAN_FN(stcat_spin),
    STCAT_PLAN_FN(stcat_ew2d),
    STCAT_PLAN_FN(stcat_stg_loss_fwd),
    STCAT_PLAN_FN(stcat_stg_loss_bwd),
    STCAT_PLAN_FN(stcat_dropout),
    STCAT_PLAN_FN(stcat_mha_self_fwd),
    STCAT_PLAN_FN(stcat_mha_self_bwd),
    STCAT_PLAN_FN(stcat_mha_self_fwd_lse),
    STCAT_PLAN_FN(stcat_mha_self_bwd_lse),
    STCAT_PLAN_FN(stcat_mha_bs_fwd),
    STCAT_PLAN_FN(stcat_mha_bs_bwd),
    STCAT_PLAN_FN(stcat_attn_weights_mean),
    STCAT_PLAN_FN(stcat_attn_q1_fwd),
    STCAT_PLAN_FN(stcat_attn_q1_bwd),
    STCAT_PLAN_FN(stcat_map2d_pool),
    STCAT_PLAN_FN(stcat_map2d_cells),
    STCAT_PLAN_FN(stcat_map2d_cells_bwd),
    STCAT_PLAN_FN(stcat_map2d_pool_bwd),
    STCAT_PLAN_FN(stcat_rowscale),
    STCAT_PLAN_FN(stcat_grad_sqnorm),
    STCAT_PLAN_FN(stcat_grad_clip_scale),
    STCAT_PLAN_FN(stcat_ema_update),
    STCAT_PLAN_FN(stcat_temporal_map_argmax),
    STCAT_PLAN_FN(stcat_pl_conv_fwd),
    STCAT_PLAN_FN(stcat_pl_conv_dgrad),
    STCAT_PLAN_FN(stcat_pl_conv_dgrad_cadd),
    STCAT_PLAN_FN(stcat_pl_linear_fwd),
    STCAT_PLAN_FN(stcat_pl_linear_dgrad_mask),
    STCAT_PLAN_FN(stcat_pl_colsum),
    STCAT_PLAN_FN(stcat_pl_split_sum),
    STCAT_PLAN_FN(stcat_pl_conv_wgrad),
    STCAT_PLAN_FN(stcat_pl_conv_wgrad_ws),
    STCAT_PLAN_FN(stcat_pl_maxpool3x3s2),
    STCAT_PLAN_FN(stcat_pl_split),
    STCAT_PLAN_FN(stcat_pl_join),
    STCAT_PLAN_FN(stcat_pl_act_bwd),
    STCAT_PLAN_FN(stcat_pl_scale),
    STCAT_PLAN_FN(stcat_weight_planes_multi),
};
constexpr int kPlanFns = (int)(sizeof(g_plan_fns) / sizeof(g_plan_fns[0]));
inline stcat_plan::Plan* plan_of(void* h) { return static_cast<stcat_plan::Plan*>(h); }
}  // namespace

extern "C" {

int stcat_plan_fn_index(const char* name) {
  for (int i = 0; i < kPlanFns; ++i)
    if (strcmp(g_plan_fns[i].name, name) == 0) return i;
  return -1;
}

int stcat_plan_fn_nargs(int fn) { return (fn >= 0 && fn < kPlanFns) ? g_plan_fns[fn].nargs : -1; }

void* stcat_plan_create(void) { return new stcat_plan::Plan(); }

int stcat_plan_destroy(void* h) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl) return 0;
#ifndef STCAT_EMU
  for (void* e : pl->events)
    if (e) (void)hipEventDestroy((hipEvent_t)e);
#endif
  delete pl;
  return 0;
}

int stcat_plan_add_call(void* h, int fn, const unsigned long long* words, int nargs, int slot, int stream_arg) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || fn < 0 || fn >= kPlanFns) return fail("plan_add_call: unknown entry point %d", fn);
  if (nargs != g_plan_fns[fn].nargs) return fail("plan_add_call: %s takes %d arguments, got %d", g_plan_fns[fn].name, g_plan_fns[fn].nargs, nargs);
  if (slot < 0 || slot > 250 || stream_arg >= nargs) return fail("plan_add_call: bad stream slot / position");
  stcat_plan::Op op = {};
  op.kind = stcat_plan::OP_CALL; op.slot = (uint8_t)slot; op.fn = fn; op.arg0 = (uint32_t)pl->words.size(); op.nargs = nargs;
  op.stream_arg = stream_arg;
  for (int i = 0; i < nargs; ++i) pl->words.push_back((uint64_t)words[i]);
  pl->ops.push_back(op);
  if (slot + 1 > pl->n_slots) pl->n_slots = slot + 1;
  return (int)op.arg0;   // index of the call's first argument word (relocations refer to it)
}

int stcat_plan_add_wait(void* h, int waiter_slot, int signal_slot) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || waiter_slot < 0 || signal_slot < 0 || waiter_slot > 250 || signal_slot > 250) return fail("plan_add_wait: bad slot");
  stcat_plan::Op op = {};
  op.kind = stcat_plan::OP_WAIT; op.slot = (uint8_t)waiter_slot; op.slot2 = (uint8_t)signal_slot;
  op.arg0 = (uint32_t)pl->events.size();
  pl->events.push_back(nullptr);
  pl->ops.push_back(op);
  const int m = (waiter_slot > signal_slot ? waiter_slot : signal_slot) + 1;
  if (m > pl->n_slots) pl->n_slots = m;
  return 0;
}

int stcat_plan_add_memset(void* h, void* ptr, unsigned long long bytes, int slot, int at_front) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || slot < 0 || slot > 250) return fail("plan_add_memset: bad slot");
  stcat_plan::Op op = {};
  op.kind = stcat_plan::OP_MEMSET; op.slot = (uint8_t)slot; op.arg0 = (uint32_t)pl->words.size();
  pl->words.push_back((uint64_t)(uintptr_t)ptr);
  pl->words.push_back((uint64_t)bytes);
  if (at_front) pl->ops.insert(pl->ops.begin(), op); else pl->ops.push_back(op);
  return (int)op.arg0;
}

int stcat_plan_set_word(void* h, int word, unsigned long long value) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || word < 0 || (size_t)word >= pl->words.size()) return fail("plan_set_word: bad word index");
  pl->words[word] = (uint64_t)value;
  return 0;
}

int stcat_plan_add_yield(void* h, int tag) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl) return fail("plan_add_yield: no plan");
  stcat_plan::Op op = {};
  op.kind = stcat_plan::OP_YIELD; op.fn = tag;
  pl->ops.push_back(op);
  return 0;
}

int stcat_plan_add_reloc(void* h, int word, int ext, unsigned long long offset) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || word < 0 || (size_t)word >= pl->words.size() || ext < 0) return fail("plan_add_reloc: bad word / external index");
  pl->relocs.push_back(stcat_plan::Reloc{(uint32_t)word, (uint32_t)ext, (uint64_t)offset});
  if (ext + 1 > pl->n_ext) pl->n_ext = ext + 1;
  return 0;
}

int stcat_plan_size(void* h, int* n_ops, int* n_words, int* n_relocs) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl) return fail("plan_size: no plan");
  if (n_ops) *n_ops = (int)pl->ops.size();
  if (n_words) *n_words = (int)pl->words.size();
  if (n_relocs) *n_relocs = (int)pl->relocs.size();
  return 0;
}

/* Replay ops [start, ...) of the plan: returns 0 with *next = -1 at the end of the plan, or 0 with *next = the op after
 * a YIELD and *tag = its tag (the host does its part and calls again with start = *next).  ext[i]: this step's base
 * address of external i (relocations are applied when start == 0); streams[s]: the hipStream_t of slot s. */
int stcat_plan_run(void* h, const unsigned long long* ext, int n_ext, void* const* streams, int n_streams, int start,
                   int* next, int* tag) {
  stcat_plan::Plan* pl = plan_of(h);
  if (!pl || !next) return fail("plan_run: no plan");
  if (n_ext < pl->n_ext) return fail("plan_run: %d externals given, the plan relocates %d", n_ext, pl->n_ext);
  if (n_streams < pl->n_slots) return fail("plan_run: %d streams given, the plan uses %d slots", n_streams, pl->n_slots);
  uint64_t* w = pl->words.data();
  if (start == 0) {
    for (const stcat_plan::Reloc& r : pl->relocs) w[r.word] = (uint64_t)ext[r.ext] + r.off;
    ++pl->replays;
  }
  const int n = (int)pl->ops.size();
  for (int i = start; i < n; ++i) {
    const stcat_plan::Op& op = pl->ops[i];
    switch (op.kind) {
      case stcat_plan::OP_CALL: {
        if (op.stream_arg >= 0) w[op.arg0 + op.stream_arg] = (uint64_t)(uintptr_t)streams[op.slot];
        const int rc = g_plan_fns[op.fn].call(w + op.arg0);
        if (rc != 0) {
          char inner[400];
          snprintf(inner, sizeof(inner), "%s", g_err);
          snprintf(g_err, sizeof(g_err), "plan op %d (%s): %s", i, g_plan_fns[op.fn].name, inner);
          return rc;
        }
        break;
      }
      case stcat_plan::OP_WAIT: {
#ifndef STCAT_EMU
        if (streams[op.slot] == streams[op.slot2]) break;
        hipEvent_t ev = (hipEvent_t)pl->events[op.arg0];
        if (!ev) {
          if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail("plan_run: event creation failed");
          pl->events[op.arg0] = (void*)ev;
        }
        hipError_t e = hipEventRecord(ev, (hipStream_t)streams[op.slot2]);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)streams[op.slot], ev, 0);
        if (e != hipSuccess) return fail("plan_run: stream wait failed at op %d: %s", i, hipGetErrorString(e));
#endif
        break;
      }
      case stcat_plan::OP_MEMSET: {
        const hipError_t e = hipMemsetAsync((void*)(uintptr_t)w[op.arg0], 0, (size_t)w[op.arg0 + 1], (hipStream_t)streams[op.slot]);
        if (e != hipSuccess) return fail("plan_run: memset failed at op %d", i);
        break;
      }
      case stcat_plan::OP_YIELD:
        *next = i + 1;
        if (tag) *tag = op.fn;
        return 0;
    }
  }
  *next = -1;
  return 0;
}

}  // extern "C"
