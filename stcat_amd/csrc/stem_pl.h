// ResNet stem on the bf16 matrix pipe (round 5): conv 7x7 / 2, 3 -> 64 + FrozenBN + ReLU (models/vision_model/backbone.py:
// torchvision resnet101.conv1 / bn1 / relu behind IntermediateLayerGetter) in the split-product arithmetic of the plane
// kernels — three bf16 pieces per operand, six cross terms, fp32 accumulate — for the modes whose other GEMMs run that way.
// (The exact-fp32 stem, igemm_stem_kernel, stays what mode f32 and the two-piece modes launch.)
//
// Why a second stem: igemm_stem_kernel gathers every A element with a scalar global load (8 per thread and K-tile, each with
// its own bounds test) and multiplies on the fp32 pipe (v_mfma_f32_32x32x2_f32, 64 cycles for 4 K flops): 0.88 ms at C3,
// 50 % MFMA-busy, against 0.43 ms at the fp32 pipe's peak.  Here
//   * a workgroup owns a 16 x 16 tile of output pixels of one frame and stages the 37 x 37 x 3 input patch it needs in LDS
//     ONCE (coalesced row reads, zero fill = the conv's padding; uint8 frames are normalised on the way in: ToTensor +
//     Normalize, datasets/transforms.py:155-168) — every input pixel is read from HBM / L2 once per tile instead of ~12 times;
//   * the reduction is laid out as 21 (+1 zero) rows (ci, kh) of 8 columns (kw = 0..6 + one zero column): a lane's MFMA A
//     fragment (8 consecutive k of its pixel) is 8 CONSECUTIVE floats of the patch — four ds_read_b64 — split into three bf16
//     pieces in registers; the weights sit in LDS as three [64][176] bf16 planes, split once per workgroup (persistent
//     workgroups: one per CU walks ~49 tiles);
//   * 11 k-steps x 6 cross terms x 2 column tiles = 132 v_mfma_f32_32x32x16_bf16 per wave and tile.
// LDS: patch 3 x 37 rows x 48 floats (row stride 48: the two pixel rows of a wave sit 96 dwords = 32 banks apart, so the 32
// lanes of a ds_read_b64 group touch 64 different banks) + weights 3 x 64 x 184 bf16 (row stride 184: conflict-free for the
// 16-lane groups of ds_read_b128) = 91.9 KB, one 8-wave workgroup per CU.
#pragma once
#include "igemm_pl.h"

struct StemPlParams {
  const void* A;          // fp32 [n][3][H][W]  or  uint8 [n][H][W][3]
  const float* w;         // fp32 [64][3][7][7]
  const float* scale;     // FrozenBN scale / bias [64]
  const float* bias;
  const float* in_scale;  // uint8 form: 1 / (255 std[c]), -mean[c] / std[c]
  const float* in_shift;
  float* y;               // fp32 [n][OH][OW][64]
  int n, H, W, OH, OW;
  int tiles_x, tiles_y, total;
};

template <bool U8>
__global__ void __launch_bounds__(512) stem_pl_kernel(StemPlParams p) {
  constexpr int TH = 16, TW = 16, PH = 2 * TH + 5, PW = 2 * TW + 5, PWS = 48;      // patch 37 x 37, row stride 48 floats
  constexpr int KR = 22, KP = KR * 8, KPS = KP + 8;                               // 21 (ci, kh) rows + one zero row; x 8 columns
  constexpr int PATCH = 3 * PH * PWS, WPLANE = 64 * KPS, NEL = 3 * PH * PW, PER = (NEL + 511) / 512;
  STCAT_DYN_SHARED(char, smem);
  float* patch = reinterpret_cast<float*>(smem);
  __bf16* wpl = reinterpret_cast<__bf16*>(smem + PATCH * 4);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;

  // ---- once per workgroup: zero the patch (its pad columns are read against zero weights: they must stay finite), split the
  // weights into three planes [pi][n][(ci, kh) row][kw + zero column]
  for (int i = t; i < PATCH; i += 512) patch[i] = 0.f;
  for (int i = t; i < 64 * KP; i += 512) {
    const int n = i / KP, kk = i - n * KP, r = kk >> 3, kw = kk & 7;
    float v = (r < 21 && kw < 7) ? p.w[n * 147 + r * 7 + kw] : 0.f;       // (ci * 49 + kh * 7 + kw = r * 7 + kw, r = ci * 7 + kh)
    STCAT_UNROLL
    for (int pi = 0; pi < 3; ++pi) {
      const __bf16 q = (__bf16)v;
      wpl[pi * WPLANE + n * KPS + kk] = q;
      v -= (float)q;
    }
  }
  float isc[3] = {1.f, 1.f, 1.f}, ish[3] = {0.f, 0.f, 0.f};
  if (U8) {
    STCAT_UNROLL
    for (int c = 0; c < 3; ++c) { isc[c] = p.in_scale[c]; ish[c] = p.in_shift[c]; }
  }
  const float* Af = reinterpret_cast<const float*>(p.A);
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.A);
  const int tiles_per_frame = p.tiles_x * p.tiles_y;

  // the patch of tile `tile` -> registers (element i of the flattened [3][37][37] patch goes to thread i % 512)
  float nx[PER];
  auto fetch = [&](int tile) {
    const int nb = tile / tiles_per_frame, rem = tile - nb * tiles_per_frame, ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int gy0 = 2 * (ty * TH) - 3, gx0 = 2 * (tx * TW) - 3;
    STCAT_UNROLL
    for (int j = 0; j < PER; ++j) {
      const int i = t + j * 512;
      float v = 0.f;
      if (i < NEL && tile < p.total) {
        const int c = i / (PH * PW), r2 = i - c * (PH * PW), py = r2 / PW, px = r2 - py * PW;
        const int gy = gy0 + py, gx = gx0 + px;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
          if (U8) v = (float)A8[(((long)nb * p.H + gy) * p.W + gx) * 3 + c] * isc[c] + ish[c];
          else v = Af[(((long)nb * 3 + c) * p.H + gy) * p.W + gx];
        }
      }
      nx[j] = v;
    }
  };
  auto stash = [&]() {
    STCAT_UNROLL
    for (int j = 0; j < PER; ++j) {
      const int i = t + j * 512;
      if (i < NEL) {
        const int c = i / (PH * PW), r2 = i - c * (PH * PW), py = r2 / PW, px = r2 - py * PW;
        patch[(c * PH + py) * PWS + px] = nx[j];
      }
    }
  };

  // this lane's pixel inside the tile: wave w owns rows 2w, 2w + 1
  const int oy_l = 2 * wave + (l31 >> 4), ox_l = l31 & 15;
  const float sc0 = p.scale ? p.scale[l31] : 1.f, sc1 = p.scale ? p.scale[32 + l31] : 1.f;
  const float bi0 = p.bias ? p.bias[l31] : 0.f, bi1 = p.bias ? p.bias[32 + l31] : 0.f;

  int tile = blockIdx.x;
  fetch(tile);
  __syncthreads();          // (zero fill + weight planes complete before the first stash / read)
  for (; tile < p.total; tile += gridDim.x) {
    stash();
    __syncthreads();
    fetch(tile + gridDim.x);                 // the next tile's patch travels while this one is multiplied
    f32x16 acc[2];
    STCAT_UNROLL
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    STCAT_UNROLL
    for (int j = 0; j < KR / 2; ++j) {
      const int r = 2 * j + hi;                                   // (ci, kh) row of this lane half
      const int ci = r / 7, kh = r - ci * 7;                      // (r = 21: the zero row of the weights; any finite data)
      const int cc = r < 21 ? ci : 0, kk = r < 21 ? kh : 0;
      const float* src = &patch[(cc * PH + 2 * oy_l + kk) * PWS + 2 * ox_l];
      float av[8];
      STCAT_UNROLL
      for (int e = 0; e < 4; ++e) {
        const float2 v2 = *reinterpret_cast<const float2*>(src + 2 * e);
        av[2 * e] = v2.x; av[2 * e + 1] = v2.y;
      }
      bf16x8 a3[3];
      stcat_split8n<3>(av, a3);
      bf16x8 b3[3][2];
      STCAT_UNROLL
      for (int pi = 0; pi < 3; ++pi) {
        STCAT_UNROLL
        for (int tn = 0; tn < 2; ++tn)
          b3[pi][tn] = *reinterpret_cast<const bf16x8*>(&wpl[pi * WPLANE + (tn * 32 + l31) * KPS + r * 8]);
      }
      STCAT_UNROLL
      for (int pr = 0; pr < PlProd<3>::N; ++pr) {
        STCAT_UNROLL
        for (int tn = 0; tn < 2; ++tn)
          acc[tn] = stcat_pl_mfma<false>(a3[PlProd<3>::a(pr)], b3[PlProd<3>::b(pr)][tn], acc[tn]);
      }
    }
    // epilogue: accumulator register rr of a lane = pixel (rr & 3) + 8 (rr >> 2) + 4 hi of the wave's 32, column l31 (+ 32)
    const int nb = tile / tiles_per_frame, rem = tile - nb * tiles_per_frame, ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    STCAT_UNROLL
    for (int rr = 0; rr < 16; ++rr) {
      const int p32 = (rr & 3) + 8 * (rr >> 2) + 4 * hi;
      const int oy = ty * TH + 2 * wave + (p32 >> 4), ox = tx * TW + (p32 & 15);
      if (oy < p.OH && ox < p.OW) {
        float* dst = p.y + (((long)nb * p.OH + oy) * p.OW + ox) * 64 + l31;
        dst[0] = fmaxf(acc[0][rr] * sc0 + bi0, 0.f);
        dst[32] = fmaxf(acc[1][rr] * sc1 + bi1, 0.f);
      }
    }
    __syncthreads();        // every wave is past its patch reads: the next stash may overwrite
  }
}
